import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pita_amd as pa
from pita_amd import mlp
torch.manual_seed(12345)
net = mlp.MyMLP(hidden_size=128, hidden_layers=3, emb_size=128, out_dim=2, input_dim=2)
B = 1 << 20
x = torch.randn(B, 2, device="cuda") * 80
t = torch.full((B,), 0.3, device="cuda")
for _ in range(3):
    y = net(t, x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): y = net(t, x)
torch.cuda.synchronize()
print((time.perf_counter() - t0) / 5 * 1e3, "ms per forward of", B)
