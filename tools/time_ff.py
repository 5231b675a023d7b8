"""Force-field kernel (K4) timing at the BASELINE config C4 batch: 22 atoms, 16 384 walkers per 4 GPUs -> 4 096/GPU and 16 384."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import pita_amd
from pita_amd.alp_energy import ForceFieldEnergy
from tests._synthetic import synthetic_peptide
tabs, pos = synthetic_peptide()
rng = np.random.default_rng(3)
gbt = dict(tabs, gb_radius=rng.choice([0.12, 0.13, 0.15, 0.155, 0.17], 22), gb_scale=rng.choice([0.72, 0.79, 0.85], 22))
for name, t in (("bonded+nonbonded", tabs), ("+GB-OBC1", gbt)):
    e = ForceFieldEnergy(t, n_particles=22, temperature=300.0, data_normalization_factor=0.164, cutoff=2.0, rf_dielectric=1.0)
    for B in (4096, 16384, 262144):
        x = ((torch.tensor(pos.reshape(-1), dtype=torch.float32)[None] + 0.004 * torch.randn(B, 66)) / 0.164).cuda()
        e(x, return_force=True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): e(x, return_force=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name}: B={B}: {us:.1f} us/eval -> {B/us*1e6:.3e} walker-evals/s", flush=True)
