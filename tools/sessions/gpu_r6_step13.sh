#!/bin/bash
# round 6, step 13: bound on what the weight-fragment stream of the Winograd GEMM 1 costs (experiment build without it: wrong results)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3w_noa_ab.log 2>&1; grep "variant" $OUT/x3w_noa_ab.log | cut -c1-330
SET_AMD_LIB=$PWD/build/exp/libset_amd_noa.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3w_noa_ab_noa.log 2>&1; grep "variant" $OUT/x3w_noa_ab_noa.log | cut -c1-330
