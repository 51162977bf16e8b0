#!/bin/bash
# round 6, step 41: x3v -- step offsets staged in front of the dependency wait (d), per-wave rate cap (s_sleep before every k-step's MFMA burst: s5, s7, s07 = GEMM 2 only);
# A/B against HEAD's source interleaved, mel hashes; timelines of d and s7
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for tag in tld tls7; do
  SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_$tag.log 2>&1
  grep -A22 "^wave 7" $OUT/x3v_timeline_$tag.log | cut -c1-160
done
for rep in 1 2; do
  for tag in head d s5 s7 s07; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_slp_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_slp_ab_${tag}$rep.log | grep -v identical | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_slp_ab.log
