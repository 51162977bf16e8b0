#!/bin/bash
# round 6, step 37: host profile of the spec_denoiser bf16 step
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
MODEL=spec_denoiser DTYPE=bf16 timeout 300 python tools/host_profile.py > $OUT/host_profile_spec.log 2>&1; head -45 $OUT/host_profile_spec.log | cut -c1-200
