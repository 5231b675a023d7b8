#!/bin/bash
# Same-box A/B of ONE kernel source against an earlier revision of it (how the cuts of profiles/r04_tangent_kernel_cuts.txt
# were measured: boxes of this pool differ by up to 10 %, so both objects are built and timed on the box of one gpurun call).
#   here:   git show <rev>:pita_amd/csrc/<file>.hip > pita_amd/csrc/_old_<file>.hip      (not tracked; travels with the snapshot)
#   there:  gpurun -- 'bash tools/ab_same_box.sh <file> "<timing command>" ["<second timing command>"]'
# e.g.      bash tools/ab_same_box.sh egnn_div_kernel "python tools/time_trace.py 65536 5" "python tools/time_trace55.py 4096 2"
cd "${GRAFT_REPO_ROOT:-.}"
F=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off"
case $F in egnn_vjp_kernel|egnn_wide_mfma_kernel|egnn_wide_mfma_jvp_kernel) FL="$FL -mllvm -amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc $FL -c pita_amd/csrc/_old_$F.hip -o /tmp/old_$F.o &
cp pita_amd/csrc/$F.o /tmp/new_$F.o
wait
for rep in 1 2; do
  for v in old new; do
    cp /tmp/${v}_$F.o pita_amd/csrc/$F.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pita_amd/libpita_hip.so pita_amd/csrc/*.o
    echo "=== $v"
    for cmd in "$@"; do $cmd 2>&1 | grep -v amdgpu.ids | tail -1; done
  done
done
