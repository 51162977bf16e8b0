#!/bin/bash
# round 5, step 23: where the host time of a training step goes (cProfile over 5 steps)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s23; mkdir -p $OUT
MODEL=spec_denoiser timeout 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/host_spec.log
MODEL=campnet timeout 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/host_campnet.log
head -4 $OUT/host_spec.log; head -4 $OUT/host_campnet.log
