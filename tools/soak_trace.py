"""Soak test of the block-shared tangent stream (development aid): repeated traces at several batch sizes must give the
same bits every time (a race in the LDS ring / barrier protocol would show as run-to-run differences)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
bad = 0
for n, sizes in ((13, (1, 2, 3, 511, 4097, 20011, 65536)), (22, (1, 5, 4096, 9001)), (55, (1, 2, 257, 1031))):
    net = pita_amd.EGNN_dynamics(n, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                                 condition_time=True, condition_temperature=True, agg="sum")
    net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    for B in sizes:
        gen = torch.Generator().manual_seed(B)
        h = (torch.rand(B, generator=gen) * 3 + 0.05).cuda()
        x = pita_amd.data_utils.remove_mean((torch.randn(B, n * 3, generator=gen) * 2).cuda(), n, 3)
        b = torch.ones(B).cuda()
        ref = net.jacobian_trace(h, x, b).clone()
        t0 = time.perf_counter()
        diff = 0
        for _ in range(reps):
            diff += int(not torch.equal(net.jacobian_trace(h, x, b), ref))
        torch.cuda.synchronize()
        print(f"n={n} B={B}: {reps} repeats, {diff} differing, {(time.perf_counter()-t0)/reps*1e3:.2f} ms each, finite {bool(torch.isfinite(ref).all())}", flush=True)
        bad += diff
sys.exit(1 if bad else 0)
