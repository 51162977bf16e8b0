#!/bin/bash
# round 6, step 22: what the reverse loop spends outside the stack kernel: generic conv launches by shape
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/infer_conv_shapes.py 2>&1 | grep -v amdgpu.ids | tee $OUT/infer_conv_shapes.log
