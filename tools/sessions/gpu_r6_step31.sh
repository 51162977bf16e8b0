#!/bin/bash
# round 6, step 31: 128-frame tiles (NB = 4, 15 spilled registers) against 96-frame tiles at B = 64 and B = 48
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for B in 64 48; do
  AB_B=$B SET_AMD_LIB=$PWD/build/exp/libset_amd_nb4.so timeout 400 python tools/loop_ab_probe.py 5 env:x3v_nb4_128:SET_AMD_X3_WINO=4 env:x3v_nb3_again:SET_AMD_X3_WINO=3 env:x3v_nb4_again:SET_AMD_X3_WINO=4 > $OUT/x3v_nb4_ab_B$B.log 2>&1
  grep "variant" $OUT/x3v_nb4_ab_B$B.log | sed "s/^/B=$B /" | cut -c1-260
done
