#!/bin/bash
# round 4: row-split small-batch kernel with the pinned k-step schedule (sx_gemm) against the previous build
# (build/exp/libset_amd_noprobe.so = the same tree before that change): parity tests, soak, latency table
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_split.log; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split or split or latency or small or ragged or persistent" 2>&1 | tail -3 >> $OUT
SOAK_N=200 timeout 600 python tools/soak_x3_probe.py 2>&1 | grep -v amdgpu.ids | tail -10 >> $OUT
for i in 1 2; do
  echo "== previous build, run $i" >> $OUT; SET_AMD_LIB=build/exp/libset_amd_noprobe.so SIZES=1x800,2x800,4x800,8x800,32x800 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -8 >> $OUT
  echo "== pinned schedule, run $i" >> $OUT; SIZES=1x800,2x800,4x800,8x800,32x800 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -8 >> $OUT
done
cat $OUT
