#!/bin/bash
# round 4: step-boundary kernel (two-piece fp16 operands) with the pinned k-step schedule against the previous build
# (build/exp/libset_amd_prevb.so): boundary / loop parity tests, then the latency table on both
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_boundary.log; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "boundary or full_inference or full_size or loop" 2>&1 | tail -3 >> $OUT
for i in 1 2; do
  echo "== previous build, run $i" >> $OUT; SET_AMD_LIB=build/exp/libset_amd_prevb.so SIZES=1x800,2x800,8x800,32x800 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -8 >> $OUT
  echo "== pinned boundary GEMMs, run $i" >> $OUT; SIZES=1x800,2x800,8x800,32x800 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -8 >> $OUT
done
cat $OUT
