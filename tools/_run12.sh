python -m pytest tests -m gpu -x -q -k "egnn or traj or sampler or sharding" 2>&1 | tail -4 > gpurun_out/r3_egnn_tests_v1.log
python tools/launch_fixed_cost.py 2>&1 | grep -E "chunk= 100|chunk=  20|fit" > gpurun_out/r3_egnn_ab_v1.log
for c in dw4 aldp22 lj55; do python bench.py --config $c --steps 200 --warmup 100 --no-debiased --no-cpu-baseline --force-evals 0 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['workload'][:30], d['value'])"; done >> gpurun_out/r3_egnn_ab_v1.log
export TMPDIR=/tmp; R=$PWD; (cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3_ws_v1 -- python3 $R/bench.py --steps 200 --warmup 100 --no-debiased --no-cpu-baseline --force-evals 0 > /dev/null 2>&1)
python3 - <<'PY' > gpurun_out/r3_ws_v1.log
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3_ws_v1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "egnn_kernel" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v)*1024/1e6, "MB written per launch", len(v))
PY
rm -rf gpurun_out/r3_ws_v1
PITA_EXTRA_HIPCC_FLAGS="-DPITA_EGNN_RECOMPUTE_COLS=0" python -m pita_amd.build --force > /dev/null 2>&1
python tools/launch_fixed_cost.py 2>&1 | grep -E "chunk= 100|chunk=  20|fit" > gpurun_out/r3_egnn_ab_v0.log
for c in dw4 aldp22 lj55; do python bench.py --config $c --steps 200 --warmup 100 --no-debiased --no-cpu-baseline --force-evals 0 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['workload'][:30], d['value'])"; done >> gpurun_out/r3_egnn_ab_v0.log
python -m pita_amd.build --force > /dev/null 2>&1
