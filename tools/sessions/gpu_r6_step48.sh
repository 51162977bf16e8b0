#!/bin/bash
# round 6, step 48: host profile of the spec_denoiser and CampNet bf16 steps
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
MODEL=spec_denoiser DTYPE=bf16 timeout 300 python tools/host_profile.py > $OUT/host_profile_spec.log 2>&1; head -60 $OUT/host_profile_spec.log | cut -c1-220
MODEL=campnet DTYPE=bf16 timeout 300 python tools/host_profile.py > $OUT/host_profile_campnet.log 2>&1; head -12 $OUT/host_profile_campnet.log | cut -c1-220
