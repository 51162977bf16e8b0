#!/bin/bash
# round 5, step 16: shorter weight-gradient blocks (more split-K slices) so that the compute stream's kernels find free slots sooner
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s16; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "spec_denoiser bf16" "campnet bf16"; do set -- $cfg; for x in 1 2 4 8; do
  SET_AMD_WGRAD_TARGET_X=$x timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 wgrad_target_x=$x: %.3f ms/step  loss %.6f' % (d['ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
done; done; done
