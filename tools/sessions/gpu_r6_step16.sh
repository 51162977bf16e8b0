#!/bin/bash
# round 6, step 16: operand-stream ceilings of the Winograd GEMM 1 at 64- / 96- (16-wide MFMAs) / 128-frame tiles under the power cap
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/ceiling_w_probe.py 6 2>&1 | grep -v amdgpu.ids | tee $OUT/ceiling_w.log
