for E in 0 1 2 3; do
PITA_EXTRA_HIPCC_FLAGS="-DRING_EXP=$E" python -m pita_amd.build --force > /dev/null 2>&1; echo "RING_EXP=$E $(python tools/time_ring_force.py 2>&1 | tail -1)"
done > gpurun_out/r3_ring_exp.log 2>&1
python -m pita_amd.build --force > /dev/null 2>&1
