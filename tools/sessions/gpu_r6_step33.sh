#!/bin/bash
# round 6, step 33: shape sweep of the Winograd kernel against the direct form
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3v_shape_sweep" > $OUT/step33_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step33_tests.log; tail -15 $OUT/step33_tests.log | cut -c1-250
