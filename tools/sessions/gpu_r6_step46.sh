#!/bin/bash
# round 6, step 46: x3v -- second staging pass's loads in front of the barrier + gate as one quotient (e3; lead thread 0 again) against the previous commit (wp)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tle3.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_tle3.log 2>&1
grep -A22 "^wave 7" $OUT/x3v_timeline_tle3.log | cut -c1-160
for rep in 1 2 3; do
  for tag in wp e3; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_e3_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_e3_ab_${tag}$rep.log | grep -v identical | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_e3_ab.log
