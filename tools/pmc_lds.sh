#!/bin/bash
# LDS / scalar / instruction-fetch counters of the kernels of one command (development aid; through gpurun from the repo root):
#   tools/pmc_lds.sh <tag> <kernel-name-substring> <python script and args...>
TAG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG; mkdir -p $O; R=$PWD
( cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $R/$* > $O/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/$* > $O/p2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/$* > $O/p3.log 2>&1 )
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
        print("    %-32s %.6g  (%d launches)" % (c, sum(vals) / len(vals), len(vals)))
PY
find $O -name "*.csv" -delete
