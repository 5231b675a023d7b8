export TMPDIR=/tmp; R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_deb_stats -- python3 $R/tools/time_debiased.py 65536 > $R/gpurun_out/r3_deb_time.log 2>&1)
find gpurun_out/r3_deb_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_deb_kernel_stats.csv \;
find gpurun_out/r3_deb_stats -name "*kernel_trace.csv" -exec cp {} gpurun_out/r3_deb_kernel_trace.csv \;
rm -rf gpurun_out/r3_deb_stats
