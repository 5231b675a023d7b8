"""Dev probe: per-component errors of the debiased weight drift at one recorded walker set of a long fixture."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd as pa
from tests import test_hip_parity as H
from oracle import pita_oracle as O

def golden(n):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", n)))

which, k = sys.argv[1], int(sys.argv[2])
g = golden(H._LONG[which][0]); w = golden(H._LONG[which][1])
truth = H._long_terms_truth(g, w)
sde, sched, gam = H._long_stack(pa, golden, H._LONG[which][1])
N, B, chunk = int(g["N"]), int(g["B"]), int(g["chunk"])
times = torch.linspace(1.0, 0.0, N + 1)[:-1]
s = int(g["at"][k])
terms = sde.f(times[s], torch.tensor(g["x_at"][k]).cuda(), 1.0, gam, None, pa.LennardJonesEnergy(39, 13, 3), 1, clamp_chunk=chunk)
print("step", s, "t", float(times[s]), "h", float(sched.h(times[s:s+1])[0]))
for nm in H._TERMS:
    t64 = truth[nm][k]; hip = getattr(terms, nm).cpu().numpy().astype(np.float64); ref = g[nm][s].astype(np.float64)
    print(f"{nm:16s} |truth| mean {np.abs(t64).mean():.4e}  hip err rms {np.sqrt(((hip-t64)**2).mean()):.3e} max {np.abs(hip-t64).max():.3e} (walker {np.abs(hip-t64).argmax()})"
          f"   ref err rms {np.sqrt(((ref-t64)**2).mean()):.3e} max {np.abs(ref-t64).max():.3e} (walker {np.abs(ref-t64).argmax()})")
gm = 4/3
for nm, src in (("hip", lambda n: getattr(terms, n).cpu().numpy().astype(np.float64)), ("ref", lambda n: g[n][s].astype(np.float64))):
    e = gm*gm*(src("cross_term")-truth["cross_term"][k]) + gm*(src("divergence_score")-truth["divergence_score"][k]) + gm*(src("dUt_dt")-truth["dUt_dt"][k])
    print(nm, "sum of component errors rms", np.sqrt((e**2).mean()), " corr(cross err, dUt err)", np.corrcoef(src("cross_term")-truth["cross_term"][k], src("dUt_dt")-truth["dUt_dt"][k])[0,1])
