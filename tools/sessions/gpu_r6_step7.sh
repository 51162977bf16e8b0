#!/bin/bash
# round 6, step 7: Winograd form: next-task conditioner prefetch A/B (shipped build vs -DSET_X3W_NOPF), parity tests
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w" > $OUT/step7_x3w_tests.log 2>&1; echo "x3w tests rc=$?" | tee -a $OUT/step7_x3w_tests.log; tail -2 $OUT/step7_x3w_tests.log
timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd_prefetch:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1 > $OUT/x3w_pf_ab.log 2>&1; grep variant $OUT/x3w_pf_ab.log
SET_AMD_LIB=$PWD/build/exp/libset_amd_nopf.so timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd_no_prefetch:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1 > $OUT/x3w_pf_ab_nopf.log 2>&1; grep variant $OUT/x3w_pf_ab_nopf.log
