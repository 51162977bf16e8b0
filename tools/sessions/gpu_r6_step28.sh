#!/bin/bash
# round 6, step 28: conv1d_x2 with 128-frame wave tiles: parity of the x2 convs, V1 forward A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_cxwide.so timeout 900 python -m pytest tests/test_gpu_x2conv.py -q -x > $OUT/step28_tests.log 2>&1; echo "tests (wide-N conv build) rc=$?" | tee -a $OUT/step28_tests.log; tail -3 $OUT/step28_tests.log | cut -c1-200
for rep in 1 2; do
  timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN V1" | sed 's/^/shipped (pair kernel wide-N): /'
  SET_AMD_LIB=$PWD/build/exp/libset_amd_cxwide.so timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN V1" | sed 's/^/+ conv1d_x2 wide-N:           /'
done | tee $OUT/cx_wide_ab.log
