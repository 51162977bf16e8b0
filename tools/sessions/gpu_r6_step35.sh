#!/bin/bash
# round 6, step 35: conv-loop ceilings (HiFi-GAN wave tile) on the 32-wide and the 16-wide instruction under the power cap
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/ceiling_w_probe.py 6 5,6,5,6 2>&1 | grep -v amdgpu.ids | tee $OUT/ceiling_conv.log
