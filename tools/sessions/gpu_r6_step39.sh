#!/bin/bash
# round 6, step 39: x3v staging / gate splits as pairs (v_cvt_pk_f16_f32): A/B against HEAD's source, interleaved, mel hashes
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for tag in head pk; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3v_pk_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_pk_ab_${tag}$rep.log | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_pk_ab.log
