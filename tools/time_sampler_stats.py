"""Fused sampler launch with and without the per-step moment accumulation (stats_out): python tools/time_sampler_stats.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
from pita_amd import sde_integration as SI
B, c = 65536, 100
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
times = torch.linspace(1.0, 0.0, 1001)[:-1]
tab = SI.build_step_table(sched, pita_amd.ConstantAnnealingFactorSchedule(4 / 3), times, 1e-3, 1.0, 1.0).cuda()
x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
for label, st in (("no stats", None), ("stats_out", torch.zeros(c, 4, dtype=torch.float64, device="cuda"))):
    for rep in range(2):
        xx = x.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in range(400, 700, c):
            net.sampler_run(xx, tab[s:s + c].contiguous(), c, seed=1, step0=s, remove_mean=True, stats_out=st)
        e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / 3:.3f} ms per 100 steps")
