import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import pita_amd
w = dict(np.load("tests/golden/egnn_weights_trainedlike.npz"))
net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                             condition_time=True, condition_temperature=True, agg="sum")
net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
N = 12
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, N + 1)[:-1], 1.0 / N, 1.0, 1.0).cuda()
B = 4101
x0 = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=5).sample(B)
for steps in (1, 2, 12):
    for use_noise in (True, False):
        nz = torch.randn(steps, B, 39, device="cuda") if use_noise else None
        a = net.sampler_run(x0.clone(), tab[:steps].contiguous(), steps, seed=11, noise=nz)
        h = 2003
        c0 = net.sampler_run(x0[:h].clone(), tab[:steps].contiguous(), steps, seed=11, walker_offset=0, noise=None if nz is None else nz[:, :h].contiguous())
        c1 = net.sampler_run(x0[h:].clone(), tab[:steps].contiguous(), steps, seed=11, walker_offset=h, noise=None if nz is None else nz[:, h:].contiguous())
        c = torch.cat([c0, c1])
        d = (c - a).abs().max(dim=1).values
        bad = torch.nonzero(d > 0).flatten()
        print(f"steps={steps} noise_buf={use_noise}: n_bad={bad.numel()} first={bad[:10].tolist()} max={d.max().item():.3e}")
# forward-only check
t = torch.randn(B, device="cuda"); b = torch.rand(B, device="cuda") + 0.5
xa = torch.randn(B, 39, device="cuda")
fa = net(t, xa, b)
fb = torch.cat([net(t[:h], xa[:h], b[:h]), net(t[h:], xa[h:], b[h:])])
d = (fa - fb).abs().max(dim=1).values
print("forward: n_bad", int((d > 0).sum()), "max", d.max().item(), torch.nonzero(d > 0).flatten()[:10].tolist())
