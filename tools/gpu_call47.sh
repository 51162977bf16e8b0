#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "full_size_100" 2>&1 | grep "max|dmel\|passed\|failed" | tail -3
