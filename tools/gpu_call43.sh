#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu_full.log 2>&1
tail -3 $OUT/pytest_gpu_full.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu_full.log | head
