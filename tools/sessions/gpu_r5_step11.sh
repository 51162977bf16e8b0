#!/bin/bash
# round 5, step 11: which host lines issue the ATen kernels of a training step (tools/aten_sites.py), both models; step times after
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s11; mkdir -p $OUT; export TMPDIR=/tmp
MODEL=spec_denoiser timeout 300 python tools/aten_sites.py 2>&1 | grep -v amdgpu.ids > $OUT/aten_spec.log; head -3 $OUT/aten_spec.log
MODEL=campnet timeout 300 python tools/aten_sites.py 2>&1 | grep -v amdgpu.ids > $OUT/aten_campnet.log; head -3 $OUT/aten_campnet.log
for model in spec_denoiser campnet; do
  timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$model: %.3f ms/step, launches %s' % (d['ms_per_step'], d.get('launches_per_step')))" | tee -a $OUT/train.log
done
