#!/bin/bash
# round 6, step 44: x3v -- bias + conditioner projection added in the gate (accumulators start at zero, chunks fetched one ahead) against the accumulator-start form
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
SET_AMD_LIB=$PWD/build/exp/libset_amd_tllcp.so timeout 300 python tools/x3_timeline_probe.py > $OUT/x3v_timeline_tllcp.log 2>&1
grep -A22 "^wave 7" $OUT/x3v_timeline_tllcp.log | cut -c1-160
for rep in 1 2; do
  for tag in nolcp lcp; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_lcp_ab_${tag}$rep.log 2>&1
    grep -h "x3_winograd_default" $OUT/x3v_lcp_ab_${tag}$rep.log | sed "s/^/$tag: /" | cut -c1-330
  done
done | tee $OUT/x3v_lcp_ab.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest_parity_step44.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_parity_step44.log | cut -c1-250
