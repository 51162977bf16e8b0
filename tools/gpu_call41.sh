#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split or dependency_timeout" 2>&1 | tail -4
SIZES=1x400,1x800,2x800,3x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="
SET_AMD_SPLIT_F32=1 SIZES=1x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="
