#!/bin/bash
# round 5, step 24: host side -- plain-int device pointers / stream handles, weight lookup through the module tables: tests + same-box A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s24; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -2 | tee $OUT/pytest.log
for rep in 1 2; do for cfg in "spec_denoiser bf16" "campnet bf16"; do set -- $cfg; for old in 0 1; do
  SET_AMD_HOST_OLD=$old timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 host_old=$old: %.3f ms/step, host enqueue %.3f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $OUT/train_ab.log
done; done; done
