#!/bin/bash
# round 6, step 20: the x3v kernel on 64-frame tiles (NB = 2) against the 32-wide 64-frame form (x3w) at shapes without a 96-frame chain per CU; parity of form 3
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3w" > $OUT/step20_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step20_tests.log; tail -3 $OUT/step20_tests.log | cut -c1-200
for B in 24 16 12; do
  AB_B=$B timeout 300 python tools/loop_ab_probe.py 4 env:x3w_64:SET_AMD_X3_WINO=1 env:x3v_nb2_64:SET_AMD_X3_WINO=3 env:x3v_nb3_96:SET_AMD_X3_WINO=2 env:x3w_64_again:SET_AMD_X3_WINO=1 env:x3v_nb2_64_again:SET_AMD_X3_WINO=3 > $OUT/x3v_nb2_ab_B$B.log 2>&1
  grep "variant" $OUT/x3v_nb2_ab_B$B.log | sed "s/^/B=$B /" | cut -c1-200
done
