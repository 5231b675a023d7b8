import sys, torch, numpy as np
sys.path.insert(0, ".")
import pita_amd
from pita_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = (torch.randn(B, 39, device="cuda") * 0.5 + torch.linspace(-2, 2, 39, device="cuda")).contiguous()
logp = torch.empty(B, device="cuda"); force = torch.empty_like(x)
L = _lib.lib(); sp = _lib.stream_ptr()
def run(n):
    for _ in range(n):
        L.pita_lj_logp_force(x.data_ptr(), logp.data_ptr(), force.data_ptr(), B, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
run(20); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(300); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 300
print(f"B={B} {us:.2f} us/launch (incl. launch gaps) -> {B*316/us/1e6:.2f} TB/s")
