#!/bin/bash
# round 4: range-masked epilogue stores in the HiFi-GAN / fp32 conv kernels: parity tests, then the vocoder with the previous build beside it
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_epi2.log; : > $OUT
if [ "${SKIP_TESTS:-0}" != 1 ]; then timeout 900 python -m pytest tests/test_gpu_x2conv.py tests/test_gpu_parity.py -x -q -m gpu -k "conv or hifigan or resblock or transpose or config3 or x2" 2>&1 | tail -3 >> $OUT; fi
for i in 1; do
  echo "== previous build, run $i" >> $OUT; SET_AMD_LIB=build/exp/libset_amd_prevepi.so HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\|stage" >> $OUT
  echo "== range-masked stores, run $i" >> $OUT; HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\|stage" >> $OUT
done
cat $OUT
