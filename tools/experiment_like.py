"""End-to-end run with the settings of experiment/lj13.yaml (debiased, resampling every step, chunks of 512,
resample_at_end, 5 adaptive MALA steps at dt = 1e-13) at a small size: a crash / finiteness check, not a benchmark."""
import copy, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
from pita_amd.energy_net import EnergyNet
w = dict(np.load(os.path.join(ROOT, "tests", "golden", "egnn_weights_trainedlike.npz")))
net = pa.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True, condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
sde = pa.VEReverseSDE(noise_schedule=sched, score_net=pa.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)), debias_inference=True)
gam = pa.ConstantAnnealingFactorSchedule(4 / 3)
e = pa.LennardJonesEnergy(39, 13, 3, temperature=3.0)
N, B = 50, 1024
integ = pa.WeightedSDEIntegrator(sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=40, resampling_interval=1,
                                 num_negative_time_steps=0, post_mcmc_steps=5, adaptive_mcmc=True, dt_negative_time=1e-13, batch_size=512,
                                 resample_at_end=True, do_langevin=False, should_mean_free=True)
scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
x1 = pa.Prior(scale=scale, n_particles=13, spatial_dim=3).sample(B)
t0 = time.time()
x, logw, uniq, terms, acc = integ.integrate_sde(x1, e, gam, inverse_temperature=1 / 3.0)
torch.cuda.synchronize()
print("ok", x.shape, logw.shape, len(uniq), uniq[:5], uniq[-3:], acc, torch.isfinite(x).all().item(), f"{time.time()-t0:.2f}s")
