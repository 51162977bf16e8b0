#!/bin/bash
# round 5, step 21: LDS-tiled SSIM filter / backward against the per-pixel gather (previous train.hip as build/exp/libset_amd_prevtrain.so)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s21; mkdir -p $OUT; R="$GRAFT_REPO_ROOT"
echo "== LDS tiles"; timeout 200 python tools/ssim_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ssim_new.log
echo "== per-pixel gather (previous build)"; SET_AMD_LIB=$R/build/exp/libset_amd_prevtrain.so timeout 200 python tools/ssim_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ssim_prev.log
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x -k "losses or golden or match_reference or bit_stable" 2>&1 | tail -2 | tee $OUT/pytest.log
for cfg in "spec_denoiser bf16" "campnet bf16"; do set -- $cfg; for lib in "" $R/build/exp/libset_amd_prevtrain.so; do
  SET_AMD_LIB=$lib timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 lib=${lib:-new}: %.3f ms/step  loss %.6f' % (d['ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
done; done
