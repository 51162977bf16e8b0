#!/bin/bash
# round 6, step 23: GEMM 2 of the x3v kernel on the 16-wide instruction, transposed (8-byte epilogue): parity, A/B against the 32-wide GEMM 2 build
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3v or full800 or ragged" > $OUT/step23_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step23_tests.log; tail -4 $OUT/step23_tests.log | cut -c1-200
for rep in 1 2; do
  SET_AMD_LIB=$PWD/build/exp/libset_amd_g2w.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_g2v_ab_w$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_g2v_ab_w$rep.log | sed 's/^/gemm2 32-wide: /' | cut -c1-340
  timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_g2v_ab_v$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_g2v_ab_v$rep.log | sed 's/^/gemm2 16-wide: /' | cut -c1-340
done
AB_B=24 SET_AMD_LIB=$PWD/build/exp/libset_amd_g2w.so timeout 300 python tools/loop_ab_probe.py 4 > $OUT/x3v_g2v_ab_B24w.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_g2v_ab_B24w.log | sed 's/^/B=24 gemm2 32-wide: /' | cut -c1-240
AB_B=24 timeout 300 python tools/loop_ab_probe.py 4 > $OUT/x3v_g2v_ab_B24v.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_g2v_ab_B24v.log | sed 's/^/B=24 gemm2 16-wide: /' | cut -c1-240
