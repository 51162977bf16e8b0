#!/bin/bash
# round 6, step 5: Winograd form A/B: shipped build vs no-setprio build (power / clock per variant)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1 > $OUT/x3w_ab2.log 2>&1; grep variant $OUT/x3w_ab2.log
SET_AMD_LIB=$PWD/build/exp/libset_amd_noprio.so timeout 600 python tools/loop_ab_probe.py 6 only-extra env:x3_winograd_noprio:SET_AMD_LOOP_LAUNCH=0,SET_AMD_X3_WINO=1 > $OUT/x3w_ab2_noprio.log 2>&1; grep variant $OUT/x3w_ab2_noprio.log
