#!/bin/bash
# round 6, step 50: x3v -- ring refills issued at the MFMA burst's priority: GEMM 2 only (e5), GEMM 1 and 2 (e6), against the current sources (e3b)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
  for tag in e3b e5 e6; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_e5_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_e5_ab_${tag}$rep.log | grep -v "identical\|sha256" | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_e5_ab.log
