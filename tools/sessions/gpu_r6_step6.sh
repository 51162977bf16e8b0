#!/bin/bash
# round 6, step 6: Winograd form with XCD-aware chunked claiming: parity tests, A/B (global counter vs chunks per XCD group)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w" > $OUT/step6_x3w_tests.log 2>&1; echo "x3w tests rc=$?" | tee -a $OUT/step6_x3w_tests.log; tail -2 $OUT/step6_x3w_tests.log
timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd_global_counter:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1,SET_AMD_X3_XCD=0 env:x3_winograd_xcd_chunks:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1,SET_AMD_X3_XCD=1 > $OUT/x3w_xcd_ab.log 2>&1; grep variant $OUT/x3w_xcd_ab.log
