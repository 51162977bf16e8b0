#!/bin/bash
# round 6, step 45: x3v -- claim / dependency look by a late wave + second staging pass's loads in front of the barrier (e1), + gate as one quotient (e2), against the previous commit (wp)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tle2.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_tle2.log 2>&1
grep -A22 "^wave 7" $OUT/x3v_timeline_tle2.log | cut -c1-160
for rep in 1 2; do
  for tag in wp e1 e2; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_e_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_e_ab_${tag}$rep.log | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_e_ab.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest_parity_step45.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_parity_step45.log | cut -c1-250
