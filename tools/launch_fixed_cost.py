"""Per-launch fixed cost of the fused sampler: event-timed launches of c steps, c = 1..100, least-squares fit
t(c) = a + b c, for the f16x2 (backup copy + kernel + repair launch) and bf16x3 (one kernel) paths."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pita_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sched = pita_amd.ElucidatingNoiseSchedule(sigma_min=0.05, sigma_max=80.0, rho=7)
gam = pita_amd.ConstantAnnealingFactorSchedule(4 / 3)
tab = pita_amd.sde_integration.build_step_table(sched, gam, torch.linspace(1.0, 0.0, 1001)[:-1], 1e-3, 1.0, 1.0).cuda()
for prec in ("f16x2", "bf16x3"):
    torch.manual_seed(12345)
    net = pita_amd.EGNN_dynamics(13, 3, hidden_nf=32, n_layers=3, recurrent=True, tanh=True, attention=True,
                                 condition_time=True, condition_temperature=True, agg="sum", precision=prec)
    x = pita_amd.Prior(scale=69.28, n_particles=13, spatial_dim=3, seed=1).sample(B)
    cs, ts = [], []
    for c in (1, 2, 5, 10, 20, 50, 100):
        reps = max(3, 100 // c)
        net.sampler_run(x, tab[:c].contiguous(), c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tabs = [tab[(i * c) % 900:(i * c) % 900 + c].contiguous() for i in range(reps)]
        e0.record()
        for i in range(reps):
            net.sampler_run(x, tabs[i], c, step0=i * c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        cs.append(c)
        ts.append(ms)
        print(f"{prec} B={B} chunk={c:4d}: {ms:8.3f} ms/launch  {ms / c:7.4f} ms/step  {B * c / ms / 1e3:.3e} walker-steps/s")
    b, a = np.polyfit(cs, ts, 1)
    print(f"{prec}: fit t(c) = {a:.3f} ms + {b:.4f} ms x c")
