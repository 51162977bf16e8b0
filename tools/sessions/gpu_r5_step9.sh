#!/bin/bash
# round 5, step 9: layer backward on 64-frame tiles, two blocks per CU (SET_AMD_BWD_TILE=64) against the 128-frame shape: bf16 tests with
# both tiles, same-box A/B of the spec_denoiser step; the leaf-stream soak with explicit diffusion steps
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s9; mkdir -p $OUT; export TMPDIR=/tmp
for tile in 64 128; do
  SET_AMD_BWD_TILE=$tile timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py -q -x 2>&1 | tail -2 | sed "s/^/bwd tile $tile: /" | tee -a $OUT/pytest.log
done
for rep in 1 2 3; do
for tile in 128 64; do
  SET_AMD_BWD_TILE=$tile timeout 300 python bench.py --mode train --model spec_denoiser --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bwd tile $tile: %.3f ms/step, host %.2f ms, layer bwd %.2f us per launch = %.3f of HBM, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], 1e3*r['launch_ms'], r['frac'], d['loss']))" | tee -a $OUT/bwd_tile_ab.log
done
done
for cfg in "spec_denoiser bf16 60" "spec_denoiser f32 20" "campnet bf16 60"; do set -- $cfg; MODEL=$1 DTYPE=$2 STEPS=$3 timeout 600 python tools/leaf_soak_probe.py 2>&1 | grep -v amdgpu.ids | tail -2; done | tee $OUT/leaf_soak.log
