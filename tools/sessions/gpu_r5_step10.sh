#!/bin/bash
# round 5, step 10: what the 2-byte memory instructions of the layer backward cost: measurement builds without the d_o stores (1), without the dy
# stores (2), without the y loads (4), without all three (7) -- timing only, results are wrong
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s10; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for v in "" 1 2 4 7; do
  lib=""; [ -n "$v" ] && lib="$R/build/exp/libset_amd_bwdexp$v.so"
  SET_AMD_LIB=$lib timeout 300 python bench.py --mode train --model spec_denoiser --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('SET_BWD_EXP=${v:-0}: layer bwd %.2f us per launch, step %.3f ms' % (1e3*r['launch_ms'], d['ms_per_step']))" | tee -a $OUT/bwd_exp.log
done
