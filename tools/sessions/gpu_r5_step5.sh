#!/bin/bash
# round 5, step 5: the pre-LN FFN sub-block as one tape node (fused vs per-op tape: cross-check test, training tests, A/B of both steps);
# the store-data hazard probe with the data produced by packed / plain multiplies right in front of the store
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s5; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"

timeout 1200 python -X faulthandler -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py tests/test_gpu_bf16.py -q -x 2>&1 | grep -v '^  File "/usr/l' | tail -12 | tee $OUT/pytest.log
for rep in 1 2; do
for fused in 1 0; do
  for model in spec_denoiser campnet; do
    SET_AMD_FUSED_NODES=$fused timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused_nodes=$fused $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f | dominant %s' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss'], json.dumps(d['roofline'])[:260]))" | tee -a $OUT/train_ab.log
  done
done
done
MODEL=campnet timeout 300 python tools/host_profile.py 2>&1 | grep "host phases" | tee $OUT/host_campnet.log
MODEL=spec_denoiser timeout 300 python tools/host_profile.py 2>&1 | grep "host phases" | tee $OUT/host_spec.log
