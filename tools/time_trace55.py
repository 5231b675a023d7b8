"""Timing of pita_egnn_jacobian_trace alone (all N*dim directions) at a given batch: python tools/time_trace.py [B] [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(55, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
x = pita_amd.Prior(scale=3.0, n_particles=55, spatial_dim=3).sample(B)
h1 = torch.full((B,), 1.0).cuda(); b1 = torch.ones(B).cuda()
ref = net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(reps): out = net.jacobian_trace(h1, x, b1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"jacobian_trace (165 directions), B={B}: {dt*1e3:.2f} ms; checksum {float(out.double().sum()):.9e} "
      f"nonfinite {int((~torch.isfinite(out)).sum())}")
