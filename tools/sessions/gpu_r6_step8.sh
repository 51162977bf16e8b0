#!/bin/bash
# round 6, step 8: Winograd form as the default: whole GPU suite (no -x: list every test that assumed the direct form), A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_step8.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_step8.log; grep "^FAILED\|passed\|failed" $OUT/pytest_gpu_step8.log | tail -30
timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_default_ab.log 2>&1; grep "variant\|bit-identical" $OUT/x3w_default_ab.log
