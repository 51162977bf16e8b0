#!/bin/bash
# round 6, step 11: worker count of the Winograd kernel (8000 tasks: 256 workers = 31.25 rounds, 250 = 32.0)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/loop_ab_probe.py 5 only-extra env:grid250:SET_AMD_STACK_GRID=250 env:grid256:SET_AMD_STACK_GRID=256 env:grid242:SET_AMD_STACK_GRID=242 env:grid267:SET_AMD_STACK_GRID=267 env:grid320:SET_AMD_STACK_GRID=320 > $OUT/x3w_grid_ab.log 2>&1; grep "variant" $OUT/x3w_grid_ab.log | cut -c1-330
