"""Headline sampler only: walker-steps/s of 100-step launches (after 2 warm launches), 65 536 LJ13 walkers."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, 1001)[:-1], 1e-3, 1.0, 1.0).cuda()
torch.manual_seed(12345)
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
x = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=1).sample(B)
c = 100
for i in range(2):
    net.sampler_run(x, tab[i * c:(i + 1) * c].contiguous(), c, step0=i * c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(2, 6):
    net.sampler_run(x, tab[i * c:(i + 1) * c].contiguous(), c, step0=i * c)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
print(f"STAGGER={os.environ.get('PITA_EGNN_STAGGER', '0')}: {ms:.3f} ms per 100 steps -> {B * c / ms / 1e3:.4e} walker-steps/s, x finite {bool(torch.isfinite(x).all())}")
