bash tools/collect_pmc.sh r3_lj13 lj13 > gpurun_out/r3_collect_lj13.log 2>&1
