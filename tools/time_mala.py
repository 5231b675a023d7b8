"""Timing of the MALA chain on the LJ13 target: fused launch (pita_lj_mala) against the launch-per-kernel path."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = dict(np.load(os.path.join(ROOT, "tests/golden/post_lj13.npz")))
e = pa.LennardJonesEnergy(39, 13, 3)
base = torch.tensor(g["x0"])
x0 = (base[torch.arange(B) % base.shape[0]] + 0.02 * torch.randn(B, 39)).cuda()
for adaptive in (False, True):
    for fused in (True, False):
        integ = pa.WeightedSDEIntegrator(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                                         post_mcmc_steps=steps, dt_negative_time=3e-4, adaptive_mcmc=adaptive, seed=9)
        fn = (lambda: integ.metropolis_hastings_mala_adaptive(x0.clone(), e, dt_init=3e-4, fused=fused)) if adaptive else \
             (lambda: integ.metropolis_hastings_mala(x0.clone(), e, fused=fused))
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} steps={steps} adaptive={adaptive} fused={fused}: {dt / steps * 1e6:.1f} us per MALA step "
              f"({2 * B * steps / dt:.3e} target evaluations/s)")
