#!/bin/bash
# round 5, step 1: the leaf stream (weight / bias gradients off the compute stream), one-launch layer image pack and backward reduce.
# full -m gpu suite, then the two bf16 training steps with the leaf stream on / off, then the per-kernel stats of both steps.
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s1; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
T0=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/pytest_gpu.log; echo "wall $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
for leaf in 1 0; do
  for model in spec_denoiser campnet; do
    SET_AMD_LEAF_STREAM=$leaf timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('leaf=$leaf $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
  done
done
SET_AMD_GRAPH_STEP=1 timeout 300 python bench.py --mode train --model spec_denoiser --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph spec_denoiser: %.3f ms/step, host enqueue %.2f ms, replays %s loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['graph_replays'], d['loss']))" | tee -a $OUT/train_ab.log
SET_AMD_GRAPH_STEP=1 timeout 300 python bench.py --mode train --model campnet --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph campnet: %.3f ms/step, host enqueue %.2f ms, replays %s loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['graph_replays'], d['loss']))" | tee -a $OUT/train_ab.log
timeout 300 python bench.py --mode train --model spec_denoiser --dtype f32 --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 spec_denoiser: %.3f ms/step, host enqueue %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $OUT/train_ab.log
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- "$@" > "$R/$OUT/rocprof_$name.log" 2>&1)
  local db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $OUT/${name}_kernel_stats.csv > /dev/null 2>&1
  rm -rf $OUT/prof
  head -4 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
prof train_bf16 python "$R/bench.py" --mode train --dtype bf16 --steps 10 --warmup 3
prof campnet_bf16 python "$R/bench.py" --mode train --model campnet --dtype bf16 --steps 10 --warmup 3
