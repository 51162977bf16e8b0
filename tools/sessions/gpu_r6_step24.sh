#!/bin/bash
# round 6, step 24: transposed 16-wide GEMM 2 with 16-byte epilogue accesses (T % 4 == 0 only: experiment), A/B against the 32-wide GEMM 2 build
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  SET_AMD_LIB=$PWD/build/exp/libset_amd_g2w.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_g2v16_ab_w$rep.log 2>&1; grep "x3_winograd_default" $OUT/x3v_g2v16_ab_w$rep.log | sed 's/^/gemm2 32-wide: /' | cut -c1-340
  timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_g2v16_ab_v$rep.log 2>&1; grep "x3_winograd_default" $OUT/x3v_g2v16_ab_v$rep.log | sed 's/^/gemm2 16-wide: /' | cut -c1-340
done
