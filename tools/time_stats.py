"""Cost of the per-step statistics in the fused sampler (development aid): 1000 steps at B walkers with and without stats_out."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(12345)
net = pa.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                       condition_time=True, condition_temperature=True, agg="sum")
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pa.ConstantAnnealingFactorSchedule(4 / 3)
times = torch.linspace(1.0, 0.0, 1001)[:-1]
tab = pa.sde_integration.build_step_table(sched, gam, times, 1e-3, 1.0, 1.0).cuda()
x = pa.Prior(scale=69.0, n_particles=13, spatial_dim=3, seed=1).sample(B)
for stats in (None, torch.zeros(1000, 4, dtype=torch.float64, device="cuda")):
    xx = x.clone()
    net.sampler_run(xx, tab[:100].contiguous(), 100, seed=1, stats_out=None if stats is None else stats[:100])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.sampler_run(xx, tab, 1000, seed=1, stats_out=stats)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"stats {'on' if stats is not None else 'off'}: {dt*1e3:.1f} ms for 1000 steps -> {B*1000/dt:.3e} walker-steps/s")
