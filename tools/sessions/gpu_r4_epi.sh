#!/bin/bash
# round 4: branch-free (out-of-range-masked) epilogue stores: parity tests of the kernels touched, then the training steps
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_epi.log; : > $OUT
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_bf16.py tests/test_gpu_campnet.py tests/test_gpu_training.py} -x -q -m gpu 2>&1 | tail -4 >> $OUT
for m in spec_denoiser campnet; do
  timeout 300 python bench.py --mode train --model $m --dtype bf16 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m bf16: %.2f ms/step, host enqueue %.2f ms, loss %.5f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" >> $OUT
done
if [ "${F32:-0}" = 1 ]; then timeout 300 python bench.py --mode train --dtype f32 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec_denoiser f32: %.2f ms/step, loss %.5f' % (d['ms_per_step'], d['loss']))" >> $OUT; fi
cat $OUT
