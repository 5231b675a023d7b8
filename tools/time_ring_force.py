"""Only the LJ55 logp+force launches (for rocprofv3 --pmc passes): 32 768 walkers, 50 launches."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = np.load(os.path.join(ROOT, "tests/golden/lj55_logp_force.npz"))
base = torch.as_tensor(g["x"][: int(g["n_cold"])], dtype=torch.float32)
x = base.repeat((B + base.shape[0] - 1) // base.shape[0], 1)[:B].contiguous().cuda()
lp, f = torch.empty(B, device="cuda"), torch.empty_like(x)
L, sp = pa._lib.lib(), pa._lib.stream_ptr()
for _ in range(50):
    L.pita_lj_logp_force(x.data_ptr(), lp.data_ptr(), f.data_ptr(), B, 55, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    L.pita_lj_logp_force(x.data_ptr(), lp.data_ptr(), f.data_ptr(), B, 55, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
e1.record(); torch.cuda.synchronize()
print(f"LJ55 B={B}: {e0.elapsed_time(e1) * 20:.1f} us per launch")
