#!/bin/bash
# round 6, step 25: conditioner projections of the reverse loop: 20 launches against one stacked launch
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/condproj_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/condproj_probe.log
