#!/bin/bash
# round 5, step 18: bf16 conv epilogue with all residual / bias loads up front, 1x1 one-shot kernel with its weight slice in registers, narrow
# LayerNorm-backward blocks: kernel A/B alone on the GPU (previous bf16.hip as build/exp/libset_amd_prevbf16.so), tests, step times
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s18; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
echo "== new"; timeout 200 python tools/conv_res_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_res_new.log
echo "== previous bf16.hip"; SET_AMD_LIB=$R/build/exp/libset_amd_prevbf16.so timeout 200 python tools/conv_res_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_res_prev.log
timeout 200 python tools/ln_bwd_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ln_bwd_probe3.log
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2; do for cfg in "spec_denoiser bf16" "campnet bf16"; do set -- $cfg; for lib in "" $R/build/exp/libset_amd_prevbf16.so; do
  SET_AMD_LIB=$lib timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 lib=${lib:-new}: %.3f ms/step  loss %.6f' % (d['ms_per_step'], d['loss']))" | tee -a $OUT/train_ab.log
done; done; done
