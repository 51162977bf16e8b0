#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split or persistent_stack_bit or dependency_timeout" > gpurun_out/r02/pytest_call20.log 2>&1
tail -4 gpurun_out/r02/pytest_call20.log
SIZES=1x400,1x800,2x800,4x800,5x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B=" > gpurun_out/r02/latency_split.log
SET_AMD_SPLIT=0 SIZES=1x800,2x800,4x800,5x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B=" > gpurun_out/r02/latency_nosplit.log
echo split; cat gpurun_out/r02/latency_split.log; echo nosplit; cat gpurun_out/r02/latency_nosplit.log
