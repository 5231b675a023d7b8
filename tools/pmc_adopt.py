"""Adopt one collect_pmc.sh result (gpurun_out/<tag>/) into profiles/: the bench lines, the rocprofv3 kernel-stats
summary and the PMC summary of <config>, named per round, and merge the PMC summary into
profiles/pmc_sampler_current.json (what bench.py reads).   usage: python tools/pmc_adopt.py <tag> <config> <round>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(ROOT, "gpurun_out", tag)
prof = os.path.join(ROOT, "profiles")
for a, b in ((f"bench_{cfg}.json", f"{rnd}_bench_{cfg}.json"), (f"bench_under_rocprof_{cfg}.json", f"{rnd}_bench_under_rocprof_{cfg}.json"),
             (f"kernel_stats_{cfg}.csv", f"{rnd}_kernel_stats_{cfg}.csv")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(prof, b))
new = json.load(open(os.path.join(src, f"pmc_sampler_{cfg}.json")))
for name in ("pmc_sampler_current.json", f"{rnd}_pmc_summary.json"):
    path = os.path.join(prof, name)
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur.update(new)
    json.dump(cur, open(path, "w"), indent=1)
print("adopted", tag, cfg, "->", prof)
