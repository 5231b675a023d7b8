"""Fused descent (generic pair kernel) for LJ55 at the config-C5 per-GPU batch and for DW4 at the C2 batch."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import pita_amd as pa
for name, e, B, x0 in (
    ("LJ55", pa.LennardJonesEnergy(165, 55, 3), 32768, None),
    ("DW4", pa.MultiDoubleWellEnergy(8, 4, 2), 65536, None)):
    n, d = e.n_particles, e.n_spatial_dim
    if name == "LJ55":
        g = np.load("tests/golden/lj55_logp_force.npz")
        base = torch.as_tensor(g["x"][: int(g["n_cold"])], dtype=torch.float32)
        x = base.repeat((B + base.shape[0] - 1) // base.shape[0], 1)[:B].contiguous().cuda()
    else:
        x = (torch.randn(B, 8) * 1.5).cuda()
    S = 200
    integ = pa.WeightedSDEIntegrator(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                                     num_negative_time_steps=S, dt_negative_time=1e-5)
    for fused in (True, False):
        integ.negative_time_descent(x, e, fused=fused); torch.cuda.synchronize()
        t0 = time.perf_counter()
        integ.negative_time_descent(x, e, fused=fused); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / S
        print(f"{name} B={B} fused={fused}: {dt*1e6:.1f} us/step -> {B/dt:.3e} walker-evals/s", flush=True)
    lp, f = e(x, return_force=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): e(x, return_force=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{name} B={B} logp+force kernel: {dt*1e6:.1f} us/eval -> {B/dt:.3e} walker-evals/s")
