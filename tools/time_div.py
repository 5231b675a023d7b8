import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pita_amd
B = 65536
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
h1 = torch.full((B,), 1.0).cuda(); b1 = torch.ones(B).cuda()
tr = net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(3): net.jacobian_trace(h1, x, b1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"jacobian_trace (39 directions): {dt*1e3:.2f} ms  checksum {tr.double().sum().item():.6f}")
