"""Timeline of debiased steps from a rocprofv3 kernel trace: busy time, gaps, per-kernel totals per step.
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/debiased_timeline.py run 65536
    python3 tools/debiased_timeline.py parse <dir>"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    import copy
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    import pita_amd
    from pita_amd.energy_net import EnergyNet
    B = int(sys.argv[2])
    w = dict(np.load(os.path.join(ROOT, "tests/golden/egnn_weights_trainedlike.npz")))
    net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                                 condition_time=True, condition_temperature=True, agg="sum")
    net.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
    sde = pita_amd.VEReverseSDE(noise_schedule=sched, score_net=pita_amd.ScoreNet(net), energy_net=EnergyNet(copy.deepcopy(net)),
                                debias_inference=True)
    gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
    x = pita_amd.Prior(scale=3.0, n_particles=13, spatial_dim=3).sample(B)
    t = torch.tensor(0.5)
    L = pita_amd._lib.lib()
    for _ in range(6):
        terms = sde.f(t, x, 1.0, gam, None, None, clamp_chunk=512)
        L.pita_em_step(x.data_ptr(), terms.drift_X.data_ptr(), 0, B, 13, 3, 1e-3, 0.1, float(np.sqrt(1e-3)), 1, 0, 0, 1, 0,
                       pita_amd._lib.stream_ptr(x.device))
    torch.cuda.synchronize()
else:
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    rows.sort()
    # steps are delimited by the cache writer launch
    idx = [i for i, r in enumerate(rows) if "egnn_vjp_kernel" in r[2] and ", 2>" in r[2]]
    for a, b in zip(idx[2:-1], idx[3:]):
        seg = rows[a:b]
        span = seg[-1][1] - seg[0][0] if b >= len(rows) else rows[b][0] - seg[0][0]
        busy = sum(e - s for s, e, _ in seg)
        big = sum(e - s for s, e, n in seg if any(k in n for k in ("tangent_shared", "div_fast", "egnn_vjp_kernel<13, 3, 7, 4, true, 2>")))
        gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)]
        print(f"step: span {span/1e6:.3f} ms, kernels {len(seg)}, busy {busy/1e6:.3f} ms (big five {big/1e6:.3f}), idle {(span-busy)/1e6:.3f} ms")
        for g, n0, n1 in sorted(gaps, reverse=True)[:5]:
            print(f"      gap {g/1e3:8.1f} us  after {n0}  before {n1}")
