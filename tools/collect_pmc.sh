#!/bin/bash
# Collect the rocprofv3 evidence of one bench configuration on the GPU box (run through gpurun from the repo root):
#   tools/collect_pmc.sh <tag> <config> [extra bench args]
# Writes gpurun_out/<tag>/: bench line, --kernel-trace --stats summary, and six separate --pmc passes (SQ a, SQ b,
# FETCH_SIZE and WRITE_SIZE at 100 steps per launch, FETCH_SIZE and WRITE_SIZE at 10 steps per launch); a counter pass
# carries --kernel-trace (kernel names) and no other trace domain.  Then the summary JSON with the traffic fit.
set -u
TAG="$1"; CFG="$2"; shift 2
OUT="$PWD/gpurun_out/$TAG"
REPO="$PWD"
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--config $CFG --steps 200 --warmup 100 --no-cpu-baseline --no-debiased --no-e2e --no-small-batch $*"
python3 bench.py --config "$CFG" "$@" > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
# the stats pass runs the bench's default protocol (1 000 timed steps in 100-step launches after 100 warm-up steps) so that
# its per-kernel average is over 11 launches, not dominated by the first, cold one
ARGS_STATS="--config $CFG --steps 1000 --warmup 100 --no-cpu-baseline --no-debiased --no-e2e --no-small-batch $*"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$CFG" -- python3 "$REPO/bench.py" $ARGS_STATS > "$OUT/bench_under_rocprof_$CFG.json" 2> "$OUT/rocprof_$CFG.err" )
WALK=$(python3 -c "import json;print(json.load(open('$OUT/bench_$CFG.json'))['config']['walkers_per_gpu'])")
P1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc${i}_$CFG" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2>> "$OUT/rocprof_$CFG.err" )
done
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc${i}_$CFG" -- python3 "$REPO/bench.py" $ARGS --chunk 10 > /dev/null 2>> "$OUT/rocprof_$CFG.err" )
done
python3 tools/pmc_summarise.py "$CFG" "$WALK" 100 "$OUT/pmc_sampler_$CFG.json" "$OUT/pmc1_$CFG" "$OUT/pmc2_$CFG" "$OUT/pmc3_$CFG" "$OUT/pmc4_$CFG" --second 10 "$OUT/pmc5_$CFG" "$OUT/pmc6_$CFG" > /dev/null
# keep what is judged small: the stats CSV and the summary; drop the raw per-dispatch counter dumps
find "$OUT/stats_$CFG" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$CFG.csv" \;
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
head -60 "$OUT/pmc_sampler_$CFG.json"
