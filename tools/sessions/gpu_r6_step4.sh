#!/bin/bash
# round 6, step 4: phase shares of block 0 of the split-operand kernel, direct vs Winograd form (probe build with s_memtime stamps)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
export SET_AMD_LIB=$PWD/build/exp/libset_amd_probe.so
(SET_AMD_X3_WINO=0 timeout 300 python tools/x3_phase_probe.py; SET_AMD_X3_WINO=1 timeout 300 python tools/x3_phase_probe.py) > $OUT/x3w_phase_probe.log 2>&1
cat $OUT/x3w_phase_probe.log | tail -8
