"""Whole NOT-debiased integrate_sde (the headline path through the plug-in class, per-step moments on) at 65 536 walkers:
per-step wall time beside the bare fused launch.  python tools/time_integrate_plain.py [walkers] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net), debias_inference=False)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
e = pita_amd.LennardJonesEnergy(39, 13, 3)
for rec in (False,):
    integ = pita_amd.WeightedSDEIntegrator(sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N,
                                           resampling_interval=-1, num_negative_time_steps=0, post_mcmc_steps=0)
    x1 = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
    integ.integrate_sde(x1, e, gam, inverse_temperature=1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = integ.integrate_sde(x1, e, gam, inverse_temperature=1.0); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"integrate_sde, not debiased, {N} steps, B={B}: {dt/N*1e3:.4f} ms per step = {B*N/dt:.3e} walker-steps/s")
