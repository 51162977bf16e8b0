#!/bin/bash
# round 5, step 3: (a) training tests on the new tree (low-priority leaf stream, in-launch ordered reductions), A/B of the step times;
# (b) the fp32 conv fast-path store: the range-masked build (conv1d.hip -DSET_CONV_V2_MASKED_STORE=1) and the same store behind a full
# wait (=2), each through tools/grad_stability_probe.py with the per-call trace -- which launch does the run-to-run difference enter through?
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s3; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 900 python -X faulthandler -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py tests/test_gpu_dist.py tests/test_gpu_bf16.py -q -x 2>&1 | grep -v '^  File "/usr/l' | tail -25 | tee $OUT/pytest.log
for cfg in "1 low" "1 0" "0 low"; do
  set -- $cfg
  for model in spec_denoiser campnet; do
    SET_AMD_LEAF_STREAM=$1 SET_AMD_LEAF_PRIORITY=$2 timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('leaf=$1 prio=$2 $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
  done
done
timeout 300 python bench.py --mode train --model spec_denoiser --dtype f32 --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 spec_denoiser: %.3f ms/step, host enqueue %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $OUT/train_ab.log
for v in shipped masked masked_wait; do
  lib=""; [ $v != shipped ] && lib="$R/build/exp/libset_amd_$v.so"
  echo "== $v" | tee -a $OUT/stability.log
  SET_AMD_LIB=$lib DTYPE=f32 REPEAT=4 TRACE=1 timeout 400 python tools/grad_stability_probe.py 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-400 | tee -a $OUT/stability.log
done
