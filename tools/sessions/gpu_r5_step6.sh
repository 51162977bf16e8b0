#!/bin/bash
# round 5, step 6: the leaf stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask) -- does keeping CUs free for the compute
# stream's short kernels beat sharing all of them?  same-box A/B at 0 (no mask) / 224 / 192 / 160 / 128 CUs, both steps, twice
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s6; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x 2>&1 | tail -2 | tee $OUT/pytest.log
for rep in 1 2; do
for cus in 0 224 192 160 128; do
  for model in spec_denoiser campnet; do
    SET_AMD_LEAF_CUS=$cus timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('leaf_cus=$cus $model: %.3f ms/step, host enqueue %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $OUT/cu_mask_ab.log
  done
done
done
