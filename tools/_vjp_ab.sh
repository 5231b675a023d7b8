for V in "" "-DPITA_VJP_AGPR_WEIGHTS=1" "-DPITA_VJP_AGPR_WEIGHTS=1 -mllvm -amdgpu-mfma-vgpr-form"; do
  PITA_EXTRA_HIPCC_FLAGS="$V" python -m pita_amd.build --force > /tmp/b1.log 2>&1 && echo "[$V] $(python tools/time_vjp_only.py 2>&1 | tail -1)"
done
python -m pita_amd.build --force > /tmp/b2.log 2>&1
