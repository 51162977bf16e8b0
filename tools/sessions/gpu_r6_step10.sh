#!/bin/bash
# round 6, step 10: x3w parity tests on the shipped build (nt loads), then nt stores A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w or x3_stack or full800" > $OUT/step10_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step10_tests.log; tail -2 $OUT/step10_tests.log
timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_ntst_ab.log 2>&1; grep "variant" $OUT/x3w_ntst_ab.log
SET_AMD_LIB=$PWD/build/exp/libset_amd_ntst.so timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_ntst_ab_ntst.log 2>&1; grep "variant" $OUT/x3w_ntst_ab_ntst.log
