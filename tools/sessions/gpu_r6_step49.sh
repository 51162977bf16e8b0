#!/bin/bash
# round 6, step 49: x3v GEMM 1 -- ring slots refilled row block by row block inside the MFMA burst (e4) against e3
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tle4.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_tle4.log 2>&1
grep -A22 "^wave 7" $OUT/x3v_timeline_tle4.log | cut -c1-160
for rep in 1 2 3; do
  for tag in e3 e4; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_e4_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_e4_ab_${tag}$rep.log | grep -v identical | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_e4_ab.log
