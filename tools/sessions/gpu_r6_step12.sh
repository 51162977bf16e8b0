#!/bin/bash
# round 6, step 12: dynamic XCD-aware chunk claiming in the Winograd kernel: parity tests with it on, A/B (chunks of 32 / 64 / 16 tasks)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_X3_XCD=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w or full800_inside" > $OUT/step12_tests.log 2>&1; echo "tests (XCD chunks on) rc=$?" | tee -a $OUT/step12_tests.log; tail -2 $OUT/step12_tests.log
timeout 300 python tools/loop_ab_probe.py 6 env:xcd_chunks32:SET_AMD_X3_XCD=1 > $OUT/x3w_xcd2_ab.log 2>&1; grep "variant" $OUT/x3w_xcd2_ab.log | cut -c1-320
SET_AMD_LIB=$PWD/build/exp/libset_amd_chunk64.so timeout 300 python tools/loop_ab_probe.py 6 only-extra env:xcd_chunks64:SET_AMD_X3_XCD=1 > $OUT/x3w_xcd2_ab64.log 2>&1; grep "variant" $OUT/x3w_xcd2_ab64.log | tail -1 | cut -c1-320
SET_AMD_LIB=$PWD/build/exp/libset_amd_chunk16.so timeout 300 python tools/loop_ab_probe.py 6 only-extra env:xcd_chunks16:SET_AMD_X3_XCD=1 > $OUT/x3w_xcd2_ab16.log 2>&1; grep "variant" $OUT/x3w_xcd2_ab16.log | tail -1 | cut -c1-320
