python -m pytest tests -m gpu -x -q -k "streaming or lj_vs_oracle or lj_golden or full_size_lj13" 2>&1 | tail -4 > gpurun_out/r3_stream_tests.log
for b in 65536 262145 2097152 8388608; do python tools/time_lj.py $b; done > gpurun_out/r3_time_lj_stream.log 2>&1
for b in 262145 2097152 8388608; do PITA_LJ13_NO_STREAM=1 python tools/time_lj.py $b; done > gpurun_out/r3_time_lj_plain.log 2>&1
