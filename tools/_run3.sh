python -m pytest tests -m gpu -x -q -k "ring or lj_golden or dw4 or descent or lj55" 2>&1 | tail -4 > gpurun_out/r3_ring_tests2.log
python tools/time_ring.py > gpurun_out/r3_time_ring_g2.log 2>&1
PITA_EXTRA_HIPCC_FLAGS="-DRING_DD_GROUP=3" python -m pita_amd.build --force > /dev/null 2>&1; python tools/time_ring.py 2>&1 | grep LJ55 > gpurun_out/r3_time_ring_g3.log
PITA_EXTRA_HIPCC_FLAGS="-DRING_DD_GROUP=1" python -m pita_amd.build --force > /dev/null 2>&1; python tools/time_ring.py 2>&1 | grep LJ55 > gpurun_out/r3_time_ring_g1.log
python -m pita_amd.build --force > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 --no-debiased --no-cpu-baseline --force-last > gpurun_out/r3_bench_forcelast.json 2>&1
python bench.py --steps 20 --warmup 5 --no-debiased --no-cpu-baseline > gpurun_out/r3_bench_forcefirst.json 2>&1
python bench.py --steps 20 --warmup 5 --no-debiased --no-cpu-baseline --force-evals 0 > gpurun_out/r3_bench_noforce.json 2>&1
