"""Timing of the debiased (Feynman-Kac) regime: walker-steps/s of sde.f at a given batch and particle count.
    python tools/time_debiased.py [walkers = 8192] [particles = 13]
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/time_debiased.py 32768 55"""
import copy, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
from pita_amd.energy_net import EnergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 13
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
net = pita_amd.EGNN_dynamics(NP, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)),
                            debias_inference=True)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
x = pita_amd.Prior(scale=3.0, n_particles=NP, spatial_dim=3).sample(B)
t = torch.tensor(0.5).cuda()
sde.f(t, x, 1.0, gam, None, None); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 3 if NP <= 22 else 2
for _ in range(reps): terms = sde.f(t, x, 1.0, gam, None, None)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"debiased f: {NP} particles, B={B}: {dt*1e3:.1f} ms per step -> {B/dt:.3e} walker-steps/s (Jacobian trace + 1 reverse-mode launch + assembly)")
_, d = net.jvp(torch.full((B,), 1.0).cuda(), x, torch.ones(B).cuda(), direction=0); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(20 if NP <= 22 else 4): net.jvp(torch.full((B,), 1.0).cuda(), x, torch.ones(B).cuda(), direction=k, want_primal=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (20 if NP <= 22 else 4)
print(f"one JVP launch: {dt*1e3:.2f} ms -> {B/dt:.3e} walker-JVPs/s")

h1 = torch.full((B,), 1.0).cuda(); b1 = torch.ones(B).cuda()
net.vjp(h1, x, b1); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(10): net.vjp(h1, x, b1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"one VJP launch: {dt*1e3:.2f} ms -> {B/dt:.3e} walker-VJPs/s")
net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(3): net.jacobian_trace(h1, x, b1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"jacobian_trace ({3 * NP} directions): {dt*1e3:.2f} ms")
