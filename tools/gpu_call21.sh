#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split" > gpurun_out/r02/pytest_call21.log 2>&1
tail -2 gpurun_out/r02/pytest_call21.log
timeout 200 python tools/split_phase_probe.py 2>&1 | grep "B=" > gpurun_out/r02/split_phase.log
cat gpurun_out/r02/split_phase.log
