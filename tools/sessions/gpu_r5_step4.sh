#!/bin/bash
# round 5, step 4: the store-data hazard on the bare hardware (tools/hw/store_data_hazard), the shipped fp32 fast-path store (soffset 0)
# through 50 repeats of the full-size fp32 step, fp32 / bf16 training times, the training + parity test files
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s4; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 300 tools/hw/store_data_hazard 2>&1 | tee $OUT/store_data_hazard.log
DTYPE=f32 REPEAT=50 timeout 600 python tools/grad_stability_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee $OUT/stability_f32_50.log
DTYPE=bf16 REPEAT=20 timeout 600 python tools/grad_stability_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee $OUT/stability_bf16_20.log
for dt in f32 bf16; do
  timeout 300 python bench.py --mode train --model spec_denoiser --dtype $dt --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spec_denoiser $dt: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/train.log
done
SET_AMD_LIB=$R/build/exp/libset_amd_prevred.so timeout 300 python bench.py --mode train --model spec_denoiser --dtype f32 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spec_denoiser f32, branch store (previous build): %.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/train.log
timeout 300 python bench.py --mode train --model campnet --dtype bf16 --steps 30 --warmup 8 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('campnet bf16: %.3f ms/step, host enqueue %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $OUT/train.log
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_campnet.py -q -x 2>&1 | tail -4 | tee $OUT/pytest.log
