#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 280 python -m pytest tests/test_gpu_dist.py -q -m gpu -p no:cacheprovider -x --tb=short > gpurun_out/r02/pytest_call11.log 2>&1
tail -40 gpurun_out/r02/pytest_call11.log | cut -c1-250
