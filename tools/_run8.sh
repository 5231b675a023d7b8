python -m pytest tests -m gpu -x -q -k "ring or lj_golden or dw4 or descent or mala or lj55 or full_size" 2>&1 | tail -5 > gpurun_out/r3_ring_tests4.log
python tools/time_ring.py 2>&1 | grep LJ55 | head -3 > gpurun_out/r3_time_ring_final.log
