#!/bin/bash
# round 6, step 2: whole GPU suite on the committed tree (after the leaf-cap fix)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_step2.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_step2.log; tail -5 $OUT/pytest_gpu_step2.log
grep -h "leaf operand" $OUT/pytest_gpu_step2.log | head -2
