#!/bin/bash
# SQ counters of the kernels of one command, two passes (development aid; run through gpurun from the repo root):
#   tools/pmc_kernels.sh <tag> <kernel-name-substring> <python script and args...>
TAG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG; mkdir -p $O; R=$PWD
( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $R/$* > $O/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/$* > $O/p2.log 2>&1 )
python3 - "$O" "$PAT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("    %-28s %.6g  (%d launches)" % (c, sum(vals) / len(vals), len(vals)))
PY
find $O -name "*.csv" -delete
