"""LJ13 logp+force: the lane-per-walker kernels against the lane-per-particle ring kernel (PITA_LJ13_RING=1), same inputs,
back-to-back launches after 40 ms of the same work (development aid).  python tools/time_lj_variants.py [B ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pita_amd
from pita_amd import _lib
L = _lib.lib(); sp = _lib.stream_ptr()
Bs = [int(a) for a in sys.argv[1:]] or [65536, 262144, 1 << 21]
gen = torch.Generator(device="cuda").manual_seed(1)
for B in Bs:
    x = (torch.randn(B, 39, device="cuda", generator=gen) * 0.5 + torch.linspace(-2, 2, 39, device="cuda")).contiguous()
    res = {}
    for name, env in (("lane=walker", None), ("ring", "1")):
        if env is None: os.environ.pop("PITA_LJ13_RING", None)
        else: os.environ["PITA_LJ13_RING"] = env
        logp = torch.empty(B, device="cuda"); force = torch.empty_like(x)
        def run(n):
            for _ in range(n):
                L.pita_lj_logp_force(x.data_ptr(), logp.data_ptr(), force.data_ptr(), B, 13, 3, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
        run(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(20); e1.record(); torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 20
        run(min(4000, int(40 / per) + 1))
        k = max(200, min(4000, int(20 / per)))
        e0.record(); run(k); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / k
        res[name] = (us, logp.clone(), force.clone())
        print(f"B={B:8d} {name:12s} {us:8.2f} us/launch -> {B*316/us/1e6:.2f} TB/s = {B*316/us/1e6/8:.3f} of 8 TB/s", flush=True)
    a, b = res["lane=walker"], res["ring"]
    print(f"            ring vs lane=walker: logp max rel {float(((a[1]-b[1]).abs()/a[1].abs().clamp_min(1e-30)).max()):.2e}, "
          f"force rel-L2 {float((a[2]-b[2]).norm()/a[2].norm()):.2e}")
os.environ.pop("PITA_LJ13_RING", None)
