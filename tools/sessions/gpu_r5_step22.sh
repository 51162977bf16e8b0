#!/bin/bash
# round 5, step 22: SSIM tiles without per-tap range tests, against the per-pixel gather of the previous train.hip (bit-level checksums)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s22; mkdir -p $OUT; R="$GRAFT_REPO_ROOT"
echo "== LDS tiles, unrolled"; timeout 200 python tools/ssim_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ssim_new.log
echo "== per-pixel gather (previous build)"; SET_AMD_LIB=$R/build/exp/libset_amd_prevtrain.so timeout 200 python tools/ssim_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ssim_prev.log
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x -k "losses or golden or match_reference or bit_stable" 2>&1 | tail -2 | tee $OUT/pytest.log
