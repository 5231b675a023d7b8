python -m pytest tests -m gpu -x -q -k "ring or lj_golden or dw4 or descent or mala or lj55 or full_size" 2>&1 | tail -5 > gpurun_out/r3_ring_tests3.log
python tools/time_ring.py 2>&1 | grep -v amdgpu > gpurun_out/r3_time_ring_pk1.log
PITA_EXTRA_HIPCC_FLAGS="-DRING_PK_GROUP=2" python -m pita_amd.build --force > /dev/null 2>&1; python tools/time_ring.py 2>&1 | grep LJ55 | head -3 > gpurun_out/r3_time_ring_pk2.log
python -m pita_amd.build --force > /dev/null 2>&1
python tools/time_mala.py 70001 20 > gpurun_out/r3_time_mala13.log 2>&1
