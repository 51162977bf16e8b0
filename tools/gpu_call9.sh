#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -q -m gpu -p no:cacheprovider -rA -k "full_size or full_length" > gpurun_out/r02/pytest_call9.log 2>&1
grep -E "passed|failed" gpurun_out/r02/pytest_call9.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/r02/pytest_call9.log | head -20
