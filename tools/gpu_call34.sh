#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_x2conv.py -x -q -s 2>&1 | grep -v "^$" | tail -16
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hifigan or vocoder" 2>&1 | tail -3
timeout 300 python tools/hifigan_bench.py 2>&1 | grep -v Warn | tail -20 > $OUT/hifigan_stages_x2.log; cat $OUT/hifigan_stages_x2.log
