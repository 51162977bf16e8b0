#!/bin/bash
# round 6, step 40: timeline of one x3v task per block (SGPR stamps)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tl.so timeout 300 python tools/x3_timeline_probe.py 2>&1 | tee $OUT/x3v_timeline.log | cut -c1-200
