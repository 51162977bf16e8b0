#!/bin/bash
# round 5, step 26: training at ragged sizes (B = 3, T = 77, T_txt = 19, padded tails) against the reference's losses and gradients: fp32 per-op and
# fused stack, bf16 per-op and fused layer kernels within the bf16 tolerances
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s26; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py -q -x -s -k "all_gradients_match_reference or training_golden_within" 2>&1 | grep -v amdgpu.ids | grep -v "^   rel" | tail -14 | tee $OUT/pytest.log
