#!/bin/bash
# SQ counters of the wide EGNN kernels (tools/time_wide.py) through rocprofv3: tools/pmc_wide.sh <tag>
set -u
TAG="$1"
OUT="$PWD/gpurun_out/$TAG"
REPO="$PWD"
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc$i" -- python3 "$REPO/tools/time_wide.py" > "$OUT/run$i.log" 2> "$OUT/rocprof$i.err" )
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for i in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wide" not in k: continue
            k = ("mfma " if "wide64" in k else "vector ") + r.get("Grid_Size", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
    for k in sorted(acc):
        n = max(cnt[k], 1)
        print(f"pass {i} {k} ({n} dispatches):", {c: round(v / n) for c, v in acc[k].items()})
PY
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
