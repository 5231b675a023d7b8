"""Negative-time descent on LJ13: fused multi-step launch vs per-step (force kernel + em_step)."""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import pita_amd as pa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
e = pa.LennardJonesEnergy(39, 13, 3)
g = np.load("tests/golden/post_lj13.npz")
x0 = torch.as_tensor(g["x0"], dtype=torch.float32).cuda()
x0 = x0.repeat((B + x0.shape[0] - 1) // x0.shape[0], 1)[:B].contiguous()
for lang in (False, True):
    integ = pa.WeightedSDEIntegrator(sde=None, num_integration_steps=1, start_resampling_step=0, end_resampling_step=1,
                                     num_negative_time_steps=S, dt_negative_time=1e-4, do_langevin=lang)
    for fused in (True, False):
        integ.negative_time_descent(x0, e, fused=fused); torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = integ.negative_time_descent(x0, e, fused=fused); torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"B={B} S={S} langevin={lang} fused={fused}: {t*1e3:.2f} ms, {t/S*1e6:.2f} us/step, "
              f"{B*S/t:.3e} walker-evals/s, algorithmic {B*316*S/t/1e12:.2f} TB/s", flush=True)
