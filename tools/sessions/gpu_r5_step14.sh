#!/bin/bash
# round 5, step 14: fp32 training step -- 16-byte gate / res-skip backward, every fp32 weight image re-packed in one launch after the optimizer
# step, the DiffNet stack images in one launch per family: tests, then the step time (f32 and, for regressions, bf16)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s14; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -s -k "vector_and_scalar or batched_fp32 or small_op or golden or match_reference or bit_stable or leaf_stream" 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/pytest.log; cat $OUT/pytest.log
for cfg in "spec_denoiser f32" "spec_denoiser bf16" "campnet bf16"; do set -- $cfg
  timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2: %.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/train.log
done
