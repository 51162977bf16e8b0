#!/bin/bash
# round 5, step 15: embedding-table gradients and the step-projection weight gradients on the leaf stream: tests, A/B on one box
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s15; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x -k "small_op or golden or match_reference or bit_stable or leaf_stream or reduces_loss or full_size or preln" 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/pytest.log; cat $OUT/pytest.log
for rep in 1 2; do for cfg in "spec_denoiser bf16" "campnet bf16" "spec_denoiser f32"; do set -- $cfg; for x in 1 0; do
  SET_AMD_LEAF_EXTRA=$x timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 leaf_extra=$x: %.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/train_ab.log
done; done; done
