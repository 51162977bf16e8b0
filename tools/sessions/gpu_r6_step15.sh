#!/bin/bash
# round 6, step 15: every tile of an x3w launch in the Winograd form (column blocks of a tile carry their own utterance): parity tests, then
# same-box A/B against the previous build (straddling tiles through the direct form), alternating
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w or full800 or x3_stack or ragged" > $OUT/step15_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step15_tests.log; tail -3 $OUT/step15_tests.log
for rep in 1 2; do
  SET_AMD_LIB=$PWD/build/exp/libset_amd_head.so timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_straddle_ab_head$rep.log 2>&1; grep "x3_winograd_default" $OUT/x3w_straddle_ab_head$rep.log | sed 's/^/head: /' | cut -c1-330
  timeout 300 python tools/loop_ab_probe.py 6 > $OUT/x3w_straddle_ab_new$rep.log 2>&1; grep "x3_winograd_default" $OUT/x3w_straddle_ab_new$rep.log | sed 's/^/new:  /' | cut -c1-330
done
