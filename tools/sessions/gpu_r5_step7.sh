#!/bin/bash
# round 5, step 7: the residual join inside the LayerNorm backward launch (set_layernorm_ch_bwd_add): training / CampNet tests, step times
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2; do
  for model in spec_denoiser campnet; do
    timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/train.log
  done
done
