"""End-to-end run of the integrator at the metric's batch (development aid): the reference's LJ13 experiment settings through
WeightedSDEIntegrator.integrate_sde -- not-debiased over the full 1000-step grid with descent + adaptive MALA, and the
debiased Feynman-Kac regime with resampling -- with wall times and sanity checks.  python tools/e2e_check.py [B]"""
import copy, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
from pita_amd.energy_net import EnergyNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pa.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                       condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pa.ConstantAnnealingFactorSchedule(4 / 3)
e = pa.LennardJonesEnergy(39, 13, 3)
scale = float((sched.h(torch.tensor(1.0)) / gam.gamma(torch.tensor(1.0))) ** 0.5)
x1 = pa.Prior(scale=scale, n_particles=13, spatial_dim=3, seed=1).sample(B)
for debias, N, extra in ((False, 1000, dict(num_negative_time_steps=0, post_mcmc_steps=0)),
                         (False, 1000, dict(num_negative_time_steps=100, post_mcmc_steps=5, adaptive_mcmc=True, dt_negative_time=1e-5)),
                         (True, 40, dict(num_negative_time_steps=10, post_mcmc_steps=2, adaptive_mcmc=True, dt_negative_time=1e-5,
                                         resampling_interval=10, resample_at_end=True))):
    sde = pa.VEReverseSDE(noise_schedule=sched, score_net=pa.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)),
                          debias_inference=debias)
    integ = pa.WeightedSDEIntegrator(sde=sde, num_integration_steps=N, start_resampling_step=0, end_resampling_step=N - 1,
                                     batch_size=512, seed=3, **extra)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, logw, uniq, terms, acc = integ.integrate_sde(x1, e, gam, inverse_temperature=1.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lp = e(x)
    v = x.reshape(B, 13, 3)
    dmin = (torch.cdist(v, v) + 1e3 * torch.eye(13, device=x.device)).amin(dim=(1, 2))
    print(f"   min pair distance: median {float(dmin.nanmedian()):.3f}, 1% quantile {float(torch.nan_to_num(dmin, nan=0.0).quantile(0.01)):.3f}")
    print(f"debias={debias}: N={N} B={B}: {dt:.2f} s; finite x {bool(torch.isfinite(x).all())}, finite logp "
          f"{int(torch.isfinite(lp).sum())}/{B}, median logp {float(lp.median()):.2f}, terms {len(terms)}, "
          f"unique min {min(uniq)}, acc {[round(a, 3) for a in acc]}", flush=True)
    # the SDE itself must keep every walker finite; descent / MALA on the LJ target may lose walkers whose particles
    # overlap (infinite forces), exactly like the reference -- the MALA chain sets them aside
    post = extra.get("num_negative_time_steps", 0) + extra.get("post_mcmc_steps", 0) > 0
    assert len(terms) == N and (torch.isfinite(x).all() if not post else torch.isfinite(x).all(dim=1).float().mean() > 0.99)
