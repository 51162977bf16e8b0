#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3 or split_operand or f16x2" 2>&1 | tail -3
SIZES=3x800,4x800,8x800,12x800,16x800,32x800 timeout 300 python tools/latency_probe.py 2>&1 | grep "B="
SET_AMD_X3_TILE=64 SIZES=4x800,8x800,12x800 timeout 300 python tools/latency_probe.py 2>&1 | grep "B="
