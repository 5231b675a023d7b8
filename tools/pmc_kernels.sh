#!/bin/bash
# SQ counters of the kernels of one command, three passes (development aid; run through gpurun from the repo root):
#   tools/pmc_kernels.sh <tag> <kernel-name-substring> <python script and args...>
TAG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG; mkdir -p $O; R=$PWD
( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $R/$* > $O/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/$* > $O/p2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/$* > $O/p3.log 2>&1 )
python3 - "$O" "$PAT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
out = {}
for k, v in acc.items():
    print(k)
    m = {c: sum(vals) / len(vals) for c, vals in v.items()}
    for c, vals in sorted(v.items()):
        print("    %-28s %.6g  (%d launches)" % (c, m[c], len(vals)))
    g = m.get("GRBM_GUI_ACTIVE")
    e = {"launches": max(len(x) for x in v.values()), "counters": m}
    if g:  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles
        cyc = g / 8.0
        e["kernel_cycles_per_xcd"] = cyc
        if "SQ_ACTIVE_INST_VALU" in m: e["valu_issue_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: e["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if "SQ_INSTS_SALU" in m: e["salu_issue_frac"] = m["SQ_INSTS_SALU"] / (256.0 * cyc)
        if "SQ_ACTIVE_INST_LDS" in m: e["lds_issue_frac"] = 4.0 * m["SQ_ACTIVE_INST_LDS"] / (1024.0 * cyc)
        if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: e["wave_wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    out[k.replace("void pita::", "")] = e
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
find $O -name "*.csv" -delete
