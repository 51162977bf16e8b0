#!/bin/bash
# round 6, step 3: the Winograd form of the split-operand kernel: parity tests, then A/B at the metric's configuration with power / clock
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "x3w" > $OUT/step3_x3w_tests.log 2>&1; echo "x3w tests rc=$?" | tee -a $OUT/step3_x3w_tests.log
grep -h "stack (\|Winograd)\|rc=\|Error\|assert" $OUT/step3_x3w_tests.log | tail -30
timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1 > $OUT/x3w_ab.log 2>&1; tail -6 $OUT/x3w_ab.log
