#!/bin/bash
# round 6, step 21: the x3v kernel as the only Winograd form (x3w removed, image without the 32-wide planes): parity tests of the stack kernels, loop sanity
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/step21_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step21_tests.log; tail -3 $OUT/step21_tests.log | cut -c1-200
timeout 300 python tools/loop_ab_probe.py 5 > $OUT/step21_ab.log 2>&1; grep "variant" $OUT/step21_ab.log | cut -c1-330
