#!/bin/bash
# round 5, step 20: LayerNorm backward with raw-buffer addressing and the 512-thread shape: training suites + step times
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s20; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
timeout 200 python tools/ln_bwd_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ln_bwd_probe4.log
for cfg in "spec_denoiser bf16" "campnet bf16" "spec_denoiser f32"; do set -- $cfg
  timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2: %.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/train.log
done
