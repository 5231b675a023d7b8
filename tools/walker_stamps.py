"""Section profile of the walker-resident trace kernel (library built with PITA_EXTRA_HIPCC_FLAGS=-DPITA_WK_STAMPS):
    PITA_EXTRA_HIPCC_FLAGS=-DPITA_WK_STAMPS python -m pita_amd.build --force && python tools/walker_stamps.py [B]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PITA_DIV_WALKER"] = "1"
import pita_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
h1 = torch.full((B,), 1.0).cuda(); b1 = torch.ones(B).cuda()
net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(ROOT, "pita_amd/libpita_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
lib.pita_wk_stamps(buf, 1)
net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
lib.pita_wk_stamps(buf, 0)
names = ["loop top", "walker set-up", "layer start", "barrier 0", "primal tile", "edge loop", "dr/de coefficients", "S product",
         "(sum M)(Wa dH)", "node model + d pos", "barrier 1", "publish + last barrier", "", "", "", ""]
tot = sum(buf)
waves = 8 * 256
for k in range(12):
    print(f"{names[k]:>24s}: {buf[k] / waves / (B / 256):10.0f} cycles per wave and walker  ({100.0 * buf[k] / tot:5.1f} %)")
print(f"{'total':>24s}: {tot / waves / (B / 256):10.0f}")
