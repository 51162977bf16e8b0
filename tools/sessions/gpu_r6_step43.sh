#!/bin/bash
# round 6, step 43: x3v -- finished tiles published per wave (flag counts waves) against the block-wide publish; timeline; parity file on the new form
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tlwp.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_tlwp.log 2>&1
grep -A22 "^wave 7" $OUT/x3v_timeline_tlwp.log | cut -c1-160
for rep in 1 2; do
  for tag in nowp wp; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_wp_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_wp_ab_${tag}$rep.log | grep -v identical | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_wp_ab.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest_parity_step43.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_parity_step43.log | cut -c1-250
