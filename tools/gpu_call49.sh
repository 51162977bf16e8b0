#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
SOAK_N=300 timeout 500 python tools/soak_x3_probe.py 2>&1 | grep "launches" > $OUT/soak_x3.log; cat $OUT/soak_x3.log
