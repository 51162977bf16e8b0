#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split" 2>&1 | tail -1
timeout 200 python tools/split_phase_probe.py 2>&1 | grep "B=" > gpurun_out/r02/split_phase.log
cat gpurun_out/r02/split_phase.log
SIZES=1x800,2x800,4x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="
