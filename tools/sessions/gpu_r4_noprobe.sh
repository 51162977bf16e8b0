#!/bin/bash
# round 4: the split-operand stack kernel built without its run-time phase stamps (-DSET_X3_PROBE=0) against the shipped build:
# the parity file on both, then the sustained per-launch time of both (tools/x3_phase_probe.py prints us per launch)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_noprobe.log; : > $OUT
L=build/exp/libset_amd_noprobe.so
echo "== shipped build: x3 / row-split tests" >> $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3 or split or full_size or latency or small" 2>&1 | tail -3 >> $OUT
echo "== -DSET_X3_PROBE=0: whole parity file" >> $OUT
SET_AMD_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 >> $OUT
for i in 1 2; do
  echo "== shipped, run $i" >> $OUT; timeout 200 python tools/x3_phase_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT
  echo "== noprobe, run $i" >> $OUT; SET_AMD_LIB=$L timeout 200 python tools/x3_phase_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT
done
SET_AMD_LIB=$L timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $OUT
cat $OUT
