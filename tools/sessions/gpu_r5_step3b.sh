#!/bin/bash
# round 5, step 3b: same-box A/B of the in-launch ordered reductions against the previous two-launch form (build/exp/libset_amd_prevred.so =
# this tree with csrc/train.hip of the commit before); the first differing conv call of the range-masked fp32 fast-path store builds
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s3; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in "" "$R/build/exp/libset_amd_prevred.so"; do
  for model in spec_denoiser campnet; do
    SET_AMD_LIB=$lib timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=${lib##*/} $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/reduce_ab.log
  done
done
done
for v in masked masked_wait; do
  echo "== $v" | tee -a $OUT/stability2.log
  SET_AMD_LIB=$R/build/exp/libset_amd_$v.so DTYPE=f32 REPEAT=3 TRACE=1 timeout 400 python tools/grad_stability_probe.py 2>&1 | grep "FIRST\|differing .* of .* elements\|conv calls\|bit-identical\|parameters with" | cut -c1-600 | tee -a $OUT/stability2.log
done
