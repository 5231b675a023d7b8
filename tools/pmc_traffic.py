"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic.json
Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KB; FETCH_SIZE
under-counts streaming reads by 2x (the correction is applied to kernels that stream their input once and is reported
uncorrected next to it, since it is calibrated for streaming reads only).
"""
import csv, json, sys
from collections import defaultdict


def load(path, name):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def short(k):
    k = k.replace("void pita::", "").replace("pita::", "")
    return k.split("(")[0]


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 100 --warmup 100 "
                 "--no-cpu-baseline (two separate passes)", "units": "bytes per launch (mean over launches)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "pita::" not in k:
        continue
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
    out["kernels"][short(k)] = {"launches": len(fetch.get(k, [])), "fetch_raw_KB": round(f, 1),
                                "fetch_bytes_x2_streaming_correction": f * 1024 * 2, "fetch_bytes_uncorrected": f * 1024,
                                "write_bytes": w * 1024}
print(json.dumps(out, indent=1))
