"""Merge the divergence kernels' counter summary (tools/pmc_kernels.sh <tag> div tools/time_trace.py <B>: summary.json)
into profiles/pmc_sampler_current.json as `divergence_<config>`: per kernel the issue fractions, and the launch-time
weighted vector-issue fraction of the whole trace (what bench.py reports beside the debiased roofline).
usage: python tools/pmc_merge_div.py <summary.json> <config> <walkers>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summ, cfg, walkers = json.load(open(sys.argv[1])), sys.argv[2], int(sys.argv[3])
kern = {k: v for k, v in summ.items() if "egnn_div" in k}
tot = sum(v["kernel_cycles_per_xcd"] * v["launches"] for v in kern.values())
ent = {"walkers": walkers,
       "source": "tools/pmc_kernels.sh (rocprofv3 --kernel-trace --pmc, two passes) over tools/time_trace.py; tools/pmc_merge_div.py",
       "valu_issue_frac": sum(v.get("valu_issue_frac", 0.0) * v["kernel_cycles_per_xcd"] * v["launches"] for v in kern.values()) / tot,
       "kernels": {k: {q: v[q] for q in ("launches", "kernel_cycles_per_xcd", "valu_issue_frac", "mfma_pipe_busy_frac",
                                         "salu_issue_frac", "lds_issue_frac", "wave_wait_any_frac") if q in v}
                   for k, v in kern.items()}}
path = os.path.join(ROOT, "profiles", "pmc_sampler_current.json")
d = json.load(open(path))
d[f"divergence_{cfg}"] = ent
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(ent, indent=1))
