"""EGNN_dynamics_AD2_cat (22 atoms, hidden 64 x 5 layers) on the matrix-pipe kernel and (PITA_WIDE_NO_MFMA=1) the vector-pipe kernel:
walker-forwards/s."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat
torch.manual_seed(0)
net = EGNN_dynamics_AD2_cat(22, 3, condition_beta=True)
for mfma in (1, 0):
  if mfma: os.environ.pop("PITA_WIDE_NO_MFMA", None)
  else: os.environ["PITA_WIDE_NO_MFMA"] = "1"
  for B in (256, 1024, 2048, 4096, 16384, 65536):
      x = pita_amd.Prior(scale=3.0, n_particles=22, spatial_dim=3, seed=1).sample(B)
      t = torch.full((B,), 0.1).cuda(); b = torch.ones(B).cuda()
      for _ in range(2): net(t, x, b)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(5): net(t, x, b)
      e1.record(); torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / 5
      mac = 5 * (462 * (2 * 64 + 2) * 64 + 2 * 462 * 64 * 64 + 2 * 462 * 64 + 22 * 3 * 64 * 64)
      print(f"B={B}: {ms:.2f} ms per forward -> {B / ms * 1e3:.3e} walker-forwards/s, {2 * mac * B / ms / 1e9:.1f} algorithmic TFLOP/s ({'matrix' if mfma else 'vector'} pipe)")
