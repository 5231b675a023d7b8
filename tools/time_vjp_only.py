"""Only the reverse-mode launches (for rocprofv3 --pmc passes): 65 536 LJ13 walkers, 10 launches."""
import os, sys, copy
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
h1 = torch.full((B,), 1.0).cuda(); b1 = torch.ones(B).cuda()
for _ in range(3): net.vjp(h1, x, b1, want_dot_h=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): net.vjp(h1, x, b1, want_dot_h=True)
e1.record(); torch.cuda.synchronize()
print(f"vjp B={B}: {e0.elapsed_time(e1) / 10:.3f} ms per launch")
