for c in dw4 aldp22 lj55; do bash tools/collect_pmc.sh r3_$c $c > gpurun_out/r3_collect_$c.log 2>&1; done
