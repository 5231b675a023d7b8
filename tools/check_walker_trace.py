"""Walker-resident trace kernel against the cached path and the fp64 oracle; then its time at the metric's batch.
    python tools/check_walker_trace.py [B_time] [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
from oracle import pita_oracle as O
from torch.func import jacrev, vmap

def make(wfile, **kw):
    w = dict(np.load(os.path.join(ROOT, "tests/golden", wfile)))
    cfg = dict(hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True, condition_time=True,
               condition_temperature=True, agg="sum")
    cfg.update(kw)
    net = pita_amd.EGNN_dynamics(13, 3, **cfg)
    net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return net, {k: torch.tensor(v).double() for k, v in w.items()}

def both(net, h, x, b):
    os.environ["PITA_DIV_WALKER"] = "1"
    tn, dn = net.jacobian_trace(h, x, b, want_denoiser=True)
    os.environ["PITA_DIV_WALKER"] = "0"
    to, do = net.jacobian_trace(h, x, b, want_denoiser=True)
    return tn, dn, to, do

for wfile in ("egnn_weights_trainedlike.npz", "egnn_weights_seed12345.npz"):
    net, wt = make(wfile)
    B = 40
    gen = torch.Generator().manual_seed(5)
    h = torch.tensor([0.0025, 0.01, 0.3, 2.0, 40.0, 900.0, 6400.0, 1.0])[torch.arange(B) % 8]
    x = O.remove_mean(torch.randn(B, 39, generator=gen) * (1 + h.sqrt())[:, None], 13, 3)
    beta = torch.rand(B, generator=gen) + 0.7
    tn, dn, to, do = both(net, h.cuda(), x.cuda(), beta.cuda())
    bb = lambda cn, xs, b: O.egnn_forward(wt, cn, xs, b, 13, 3)
    one = lambda hh, xx, b: O.denoiser(bb, hh[None], xx[None], b[None])[0]
    J = vmap(jacrev(one, argnums=1))(h.double(), x.double(), beta.double())
    want = torch.diagonal(J, dim1=1, dim2=2).sum(-1)
    scale = float(want.abs().mean()) + 1.0
    en = (tn.cpu().double() - want).abs() / (want.abs() + scale)
    eo = (to.cpu().double() - want).abs() / (want.abs() + scale)
    print(f"{wfile}: walker kernel max err {float(en.max()):.2e} (mean {float(en.mean()):.2e}); cached path {float(eo.max()):.2e} (mean {float(eo.mean()):.2e}); "
          f"denoiser new-vs-old rel {float((dn - do).norm() / do.norm()):.2e}; nonfinite {int((~torch.isfinite(tn)).sum())}")
    if float(en.max()) > 1e-3:
        print("   first walkers: new", tn[:6].cpu().numpy(), "\n   want", want[:6].numpy(), "\n   old", to[:6].cpu().numpy())

if len(sys.argv) > 1:
    Bt = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    net, _ = make("egnn_weights_trainedlike.npz")
    x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(Bt)
    h1 = torch.full((Bt,), 1.0).cuda(); b1 = torch.ones(Bt).cuda()
    for off in (False, True, False, True):
        os.environ["PITA_DIV_WALKER"] = "0" if off else "1"
        out = net.jacobian_trace(h1, x, b1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(reps): out = net.jacobian_trace(h1, x, b1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"{'cached path' if off else 'walker kernel'}: B={Bt}: {dt*1e3:.2f} ms per trace; checksum {float(out.double().sum()):.9e} nonfinite {int((~torch.isfinite(out)).sum())}")
