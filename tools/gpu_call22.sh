#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_edit_caller.py -x -q -m gpu > gpurun_out/r02/pytest_call22.log 2>&1
tail -4 gpurun_out/r02/pytest_call22.log
SIZES=1x400,1x800,2x800,4x800,5x800,8x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B=" > gpurun_out/r02/latency_split.log
cat gpurun_out/r02/latency_split.log
