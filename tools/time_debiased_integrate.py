"""Whole debiased integrate_sde (Feynman-Kac weights, resampling every step) at 65 536 walkers: per-step wall time."""
import copy, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pita_amd
from pita_amd.energy_net import EnergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200  # a real grid: ten steps over the whole schedule throw walkers out of the f16 range
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)),
                            debias_inference=True)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
e = pita_amd.LennardJonesEnergy(39, 13, 3)
for interval in (-1, 1):
    integ = pita_amd.WeightedSDEIntegrator(sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N,
                                           resampling_interval=interval, num_negative_time_steps=0, post_mcmc_steps=0,
                                           batch_size=512)
    x1 = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
    integ.integrate_sde(x1, e, gam, inverse_temperature=1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, logw, uniq, _, _ = integ.integrate_sde(x1, e, gam, inverse_temperature=1.0); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"interval={interval}: {dt/N*1e3:.1f} ms per step, {B*N/dt:.3e} walker-steps/s, unique[-1]={uniq[-1]}")
