"""The reference's own batch sizes through the whole integrate_sde (bench.py small_batch_legs) as a table, or ONE size for
a kernel trace:
    python tools/small_batch.py                              # 512 / 2 048 / 5 000 / 16 384 / 65 536 walkers, both regimes
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/small_batch.py 2048 default
Reference operating points: configs/experiment/lj13.yaml:27,32 (inference chunks of 512, num_eval_samples 2 048),
configs/model/energytemp.yaml:69,122 (5 000 / 2 000)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pita_amd

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["lj13"]
net = bench.build_model(pita_amd, cfg["n"], cfg["d"])
if len(sys.argv) > 1:
    B, regime = int(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "both")
    r = bench.small_batch_legs(pita_amd, net, cfg, dev, sizes=(B,), n_plain=0 if regime == "default" else 1000,
                               n_default=0 if regime == "plain" else 100)
    print(json.dumps(r))
    sys.exit(0)
r = bench.small_batch_legs(pita_amd, net, cfg, dev, sizes=(512, 2048, 5000, 16384, 65536))
full = r["sizes"]["65536"]
print("LJ13, whole WeightedSDEIntegrator.integrate_sde per batch size (one MI355X)")
print(f"{'walkers':>8s} | {'not debiased, 1 000 steps':^44s} | {'default regime, 100 steps, chunks of 512':^58s} | sampler mapping")
print(f"{'':>8s} | {'ms/step':>8s} {'walker-steps/s':>15s} {'of 65 536':>9s} {'host ms':>8s} | {'ms/step':>8s} {'walker-steps/s':>15s} {'of 65 536':>9s} "
      f"{'host ms/step':>12s} {'MALA acc':>9s} | G, waves / slots")
for B, legs in r["sizes"].items():
    a, b, m = legs["not_debiased"], legs["default_regime"], legs["sampler_mapping"]
    print(f"{B:>8s} | {a['ms_per_step']:8.4f} {a['value']:15.3e} {a['value'] / full['not_debiased']['value']:9.3f} "
          f"{a['host_seconds'] * 1e3:8.1f} | {b['ms_per_step']:8.3f} {b['value']:15.3e} "
          f"{b['value'] / full['default_regime']['value']:9.3f} {b['host_seconds'] * 1e3 / b['steps']:12.3f} "
          f"{sum(b['mala_acceptance']) / max(1, len(b['mala_acceptance'])):9.3f} | "
          f"{m['walkers_per_wave_group']}, {m['waves']} / {m['resident_wave_slots']} = {m['wave_slot_fill']:.2f}")
print(json.dumps(r))
