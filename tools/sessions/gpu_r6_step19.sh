#!/bin/bash
# round 6, step 19: x3v phase probe; A/B of two register-for-latency trades (skip rows fetched before GEMM 2; GEMM 2 ring 4 deep)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
X3_NAMES="claim+init+wait,stageA,gemmA,EO+stageB,gemmB,publish,-,-,xres+gate,gemm2,epilogue" SET_AMD_LIB=$PWD/build/exp/libset_amd_probe.so timeout 300 python tools/x3_phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/x3v_phase_probe.log
for rep in 1 2; do
  timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_ab2_base$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_ab2_base$rep.log | sed 's/^/base:    /' | cut -c1-340
  SET_AMD_LIB=$PWD/build/exp/libset_amd_skearly.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_ab2_skearly$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_ab2_skearly$rep.log | sed 's/^/skearly: /' | cut -c1-340
  SET_AMD_LIB=$PWD/build/exp/libset_amd_pf24.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_ab2_pf24$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_ab2_pf24$rep.log | sed 's/^/pf2=4:   /' | cut -c1-340
done
