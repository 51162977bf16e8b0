#!/bin/bash
# round 6, step 26: stacked conditioner projection in the loop: the cache test, the goldens, loop time
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "step_table or full_inference or smoke" > $OUT/step26_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step26_tests.log; tail -3 $OUT/step26_tests.log | cut -c1-200
timeout 300 python tools/loop_ab_probe.py 5 > $OUT/step26_ab.log 2>&1; grep "variant" $OUT/step26_ab.log | cut -c1-330
