#!/bin/bash
# round 6, step 47: on the new task balance -- GEMM 2 ring 4 deep (f_pf24), skip rows fetched before GEMM 2 (f_ske), both, against e3 (= current sources)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for tag in e3 f_pf24 f_ske f_both; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_f_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_f_ab_${tag}$rep.log | grep -v identical | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_f_ab.log
