#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN\|stage" > $OUT/hifigan_stages_x2.log; cat $OUT/hifigan_stages_x2.log
