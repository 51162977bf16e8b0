#!/bin/bash
# round 6, step 17: 96-frame Winograd tiles on the 16-wide matrix instruction (diffnet_stack_x3v_kernel): parity tests, A/B against the 64-frame form
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w" > $OUT/step17_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step17_tests.log; tail -15 $OUT/step17_tests.log | cut -c1-300
timeout 600 python tools/loop_ab_probe.py 6 env:x3v_96:SET_AMD_X3_WINO=2 env:x3v_96_grid240:SET_AMD_X3_WINO=2,SET_AMD_STACK_GRID=240 env:x3v_96_grid213:SET_AMD_X3_WINO=2,SET_AMD_STACK_GRID=213 env:x3w_64_again:SET_AMD_X3_WINO=1 env:x3v_96_again:SET_AMD_X3_WINO=2 > $OUT/x3v_ab.log 2>&1; grep "variant\|identical" $OUT/x3v_ab.log | cut -c1-330
