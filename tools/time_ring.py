"""Ring kernels (LJ55 at the C5 per-GPU batch, DW4 at the C2 batch): logp+force launch, fused descent step, MALA step
(fused / per kernel, adaptive or not), with HIP-event timing."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import pita_amd as pa


def ev_time(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, e, B in (("LJ55", pa.LennardJonesEnergy(165, 55, 3), 32768), ("DW4", pa.MultiDoubleWellEnergy(8, 4, 2), 65536)):
    n, d = e.n_particles, e.n_spatial_dim
    if name == "LJ55":
        g = np.load("tests/golden/lj55_logp_force.npz")
        base = torch.as_tensor(g["x"][: int(g["n_cold"])], dtype=torch.float32)
        x = base.repeat((B + base.shape[0] - 1) // base.shape[0], 1)[:B].contiguous().cuda()
        dtm, flop = 2e-4, 1485 * 28 + 55 * 9
    else:
        x = (torch.tensor([[2.0, 2.0, -2.0, 2.0, -2.0, -2.0, 2.0, -2.0]]) + 0.3 * torch.randn(B, 8)).cuda()
        dtm, flop = 0.05, 150
    lp, f = torch.empty(B, device="cuda"), torch.empty_like(x)
    L, sp = pa._lib.lib(), pa._lib.stream_ptr()
    if name == "LJ55":
        raw = lambda: L.pita_lj_logp_force(x.data_ptr(), lp.data_ptr(), f.data_ptr(), B, n, d, 1.0, 1.0, 1e-6, 1.0, 1.0, 1.0, sp)
    else:
        raw = lambda: L.pita_dw_logp_force(x.data_ptr(), lp.data_ptr(), f.data_ptr(), B, n, d, 1.0, 0.9, -4.0, 0.0, 4.0, sp)
    us = ev_time(raw, 100)
    print(f"{name} B={B} logp+force: {us:.1f} us -> {B / us * 1e6:.3e} evals/s, {B * flop / us / 1e6:.1f} algorithmic TFLOP/s "
          f"({B * flop / us / 1e6 / 157.3:.3f} of fp32 VALU peak), {B * (2 * n * d + 1) * 4 / us / 1e3:.0f} GB/s", flush=True)
    S = 200
    integ = pa.WeightedSDEIntegrator(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                                     num_negative_time_steps=S, dt_negative_time=1e-6)
    for langevin in (False, True):
        integ.do_langevin = langevin
        for fused in (True, False):
            us = ev_time(lambda: integ.negative_time_descent(x, e, fused=fused), 2) / S
            print(f"{name} descent langevin={langevin} fused={fused}: {us:.1f} us/step -> {B / us * 1e6:.3e} evals/s", flush=True)
    steps = 20
    for adaptive in (False, True):
        for fused in (True, False):
            integ = pa.WeightedSDEIntegrator(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                                             post_mcmc_steps=steps, dt_negative_time=dtm, adaptive_mcmc=adaptive, seed=9)
            fn = (lambda: integ.metropolis_hastings_mala_adaptive(x.clone(), e, dt_init=dtm, return_acceptance_rate=True, fused=fused)) \
                if adaptive else (lambda: integ.metropolis_hastings_mala(x.clone(), e, return_acceptance_rate=True, fused=fused))
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{name} MALA adaptive={adaptive} fused={fused}: {dt / steps * 1e6:.1f} us/step, rates {out[1][0]:.2f}..{out[1][-1]:.2f}", flush=True)
