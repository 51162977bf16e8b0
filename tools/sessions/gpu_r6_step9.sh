#!/bin/bash
# round 6, step 9: nt (evict-first) hints on the read-once activation streams of the Winograd kernel, A/B against the shipped build
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_nt_ab.log 2>&1; grep "variant" $OUT/x3w_nt_ab.log
SET_AMD_LIB=$PWD/build/exp/libset_amd_nt.so timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_nt_ab_nt.log 2>&1; grep "variant" $OUT/x3w_nt_ab_nt.log
