for V in "-DPITA_WIDE64_PIPE=0" "-DPITA_WIDE64_RG=1" "-DPITA_WIDE64_RG=2" "-DPITA_WIDE64_RG=4"; do
  PITA_EXTRA_HIPCC_FLAGS="$V" python -m pita_amd.build --force > /tmp/b1.log 2>&1 && echo "$V: $(python tools/time_wide.py 2>&1 | grep matrix | tail -1)"
done
python -m pita_amd.build --force > /tmp/b2.log 2>&1
timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "ad2cat" 2>&1 | tail -3
