#!/bin/bash
# round 5, step 12: queued second stages of the gradient reductions (set_rows_sum_defer_*): equality tests, the training suites, step-time A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s12; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x -s -k "deferred or leaf_stream or bit_stable or golden or preln or reduces_loss or full_size" 2>&1 | grep -v amdgpu.ids | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
for model in spec_denoiser campnet; do for dt in bf16 f32; do for d in 1 0; do
  [ $model = campnet ] && [ $dt = f32 ] && continue
  SET_AMD_DEFER_SUMS=$d timeout 300 python bench.py --mode train --model $model --dtype $dt --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$model $dt defer=$d: %.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/train_ab.log
done; done; done
