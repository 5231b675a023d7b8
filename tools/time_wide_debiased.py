"""Debiased (Feynman-Kac) step with the alanine-dipeptide backbone EGNN_dynamics_AD2_cat (hidden 64 x 5) for score and
energy net: forward-mode launches (pita_egnn_wide_jvp) for the trace, one reverse-mode launch (pita_egnn_wide_vjp), config C4's per-GPU shard by default.
python tools/time_wide_debiased.py [walkers]"""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat
from pita_amd.energy_net import EnergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(12345)
net = EGNN_dynamics_AD2_cat(22, 3, hidden_nf=64, n_layers=5, condition_beta=True)
sched = pa.ElucidatingNoiseSchedule(sigma_min=0.01, sigma_max=80.0, rho=7)
sde = pa.VEReverseSDE(noise_schedule=sched, score_net=pa.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)), debias_inference=True)
gam = pa.ConstantAnnealingFactorSchedule(4 / 3)
x = pa.Prior(scale=3.0, n_particles=22, spatial_dim=3, seed=7).sample(B)
t = torch.tensor(0.5, device="cuda")
h = torch.full((B,), float(sched.h(torch.tensor(0.5))), device="cuda")
b = torch.ones(B, device="cuda")
def ev(fn, reps=1):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
fwd = ev(lambda: net.edm(1, h, x, b), 5)
os.environ["PITA_WIDE_NO_MFMA"] = "1"
fwd_v = ev(lambda: net.edm(1, h, x, b), 2)
del os.environ["PITA_WIDE_NO_MFMA"]
one = ev(lambda: net.jvp(h, x, b, direction=3, want_primal=False, want_tangent=False, diag_acc=torch.zeros(B, device="cuda")), 2)
rev = ev(lambda: net.vjp(h, x, b, want_dot_h=True), 2)
step = ev(lambda: sde.f(t, x, 1.0, gam, None, None, resampling_interval=1))
print(f"B={B}: denoiser forward {fwd*1e3:.2f} ms (matrix pipe), {fwd_v*1e3:.2f} ms (vector pipe); one forward-mode launch "
      f"{one*1e3:.2f} ms; one reverse-mode launch {rev*1e3:.2f} ms; debiased step (67 forward-mode launches + 1 "
      f"reverse-mode launch) {step*1e3:.1f} ms = {B/step:.3e} walker-steps/s")
