#!/bin/bash
# round 5, step 2: leaf forks as one C call; where does a step's time go?  host phases (empty queue), cProfile of the forward, GPU timeline of
# one step by stream (rocprofv3 kernel trace), A/B of the leaf stream; training tests first
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s2; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py tests/test_gpu_dist.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.log
for leaf in 1 0; do
  for model in spec_denoiser campnet; do
    SET_AMD_LEAF_STREAM=$leaf timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('leaf=$leaf $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
  done
done
MODEL=spec_denoiser timeout 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/host_spec.log; grep "host phases" $OUT/host_spec.log
MODEL=campnet timeout 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/host_campnet.log; grep "host phases" $OUT/host_campnet.log
tl() {  # name, model
  rm -rf $OUT/prof
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace -d "$R/$OUT/prof" -o t -- python "$R/bench.py" --mode train --model $2 --dtype bf16 --steps 8 --warmup 3 > "$R/$OUT/rocprof_$1.log" 2>&1)
  local db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_timeline.py $db $OUT/${1}_timeline.csv 2>&1 | tee $OUT/${1}_timeline.log
  rm -rf $OUT/prof
}
tl spec spec_denoiser
tl campnet campnet
