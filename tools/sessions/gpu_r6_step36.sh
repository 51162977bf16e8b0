#!/bin/bash
# round 6, step 36: split-operand stack on heavy-tailed weights / outlier activations against fp64
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "heavy_tailed" > $OUT/step36_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step36_tests.log; grep -h "heavy-tailed\|passed\|failed\|Error\|assert" $OUT/step36_tests.log | cut -c1-250 | head -20
