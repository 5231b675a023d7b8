"""Quick on-GPU timing of the hot kernels (development aid; bench.py is the contract)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pita_amd

torch.cuda.set_device(0)
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
N = 1000
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, N + 1)[:-1], 1.0 / N, 1.0, 1.0).cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prior = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=1)
x = prior.sample(B)
for steps in (2, 10, 50):
    xs = x.clone()
    net.sampler_run(xs, tab[:steps].contiguous(), steps, seed=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.sampler_run(xs, tab[:steps].contiguous(), steps, seed=3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sampler B={B} steps={steps}: {dt*1e3:.2f} ms  -> {B*steps/dt:.3e} walker-steps/s  "
          f"({B*steps*4.197e6/dt/1e12:.1f} algorithmic TFLOP/s)")
e = pita_amd.LennardJonesEnergy(39, 13, 3)
for Bb in (65536, 1 << 21):
    xx = (torch.randn(Bb, 39, device="cuda") * 0.5 + torch.linspace(-2, 2, 39, device="cuda"))
    e(xx, return_force=True)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        e(xx, return_force=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"LJ13 logp+force B={Bb}: {dt*1e6:.1f} us -> {Bb/dt:.3e} evals/s, {Bb*316/dt/1e12:.2f} TB/s algorithmic")
