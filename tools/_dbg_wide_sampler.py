import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pita_amd as pa
from pita_amd.egnn_dynamics_ad2_cat import EGNN_dynamics_AD2_cat
from oracle import pita_oracle as O
g = dict(np.load("tests/golden/egnn_ad2cat_h64_fwd.npz"))
w = {k[2:]: torch.tensor(v) for k, v in g.items() if k.startswith("w.")}
net = EGNN_dynamics_AD2_cat(22, 3, hidden_nf=64, n_layers=5, tanh=True, attention=True, condition_beta=True)
net.load_state_dict(w)
sched, gam = pa.ElucidatingNoiseSchedule(sigma_min=0.01, sigma_max=80.0, rho=7), pa.ConstantAnnealingFactorSchedule(4 / 3)
N, B = 8, 41
gen = torch.Generator().manual_seed(21)
x1 = O.remove_mean(torch.randn(B, 66, generator=gen) * 60.0, 22, 3).cuda()
tab = pa.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, N + 1)[:-1], 1.0 / N, 1.0, 1.3).cuda()
a = net.sampler_run(x1.clone(), tab, N, seed=9)
b = net.sampler_run(x1.clone(), tab, N, seed=9)
print("run-to-run equal:", torch.equal(a, b))
for cut in (1, 3, 7):
    p = x1.clone()
    net.sampler_run(p, tab[:cut].contiguous(), cut, seed=9)
    net.sampler_run(p, tab[cut:].contiguous(), N - cut, seed=9, step0=cut)
    d = (p - a).abs().amax(1)
    print("cut", cut, "equal", torch.equal(p, a), "walkers differing", int((d > 0).sum()), "max", float(d.max()))
p = x1.clone()
for s in range(N):
    net.sampler_run(p, tab[s:s+1].contiguous(), 1, seed=9, step0=s)
print("one-by-one equal", torch.equal(p, a), float((p-a).abs().max()))
os.environ["PITA_WIDE_NO_MFMA"] = "1"
av = net.sampler_run(x1.clone(), tab, N, seed=9)
p = x1.clone()
net.sampler_run(p, tab[:3].contiguous(), 3, seed=9); net.sampler_run(p, tab[3:].contiguous(), N - 3, seed=9, step0=3)
print("vector pipe: chunk equal", torch.equal(p, av), "vs mfma", float((av - a).abs().max()))
