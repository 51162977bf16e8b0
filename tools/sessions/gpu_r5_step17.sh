#!/bin/bash
# round 5, step 17: LayerNorm backward alone on the GPU, wide (32 frames x 8 channel groups) vs narrow (16 x 16) blocks
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s17; mkdir -p $OUT
for n in 0 1; do echo "SET_AMD_LNB_NARROW=$n"; SET_AMD_LNB_NARROW=$n timeout 200 python tools/ln_bwd_probe.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/ln_bwd_probe2.log
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_campnet.py -q -x -k "small_op or golden or match_reference or preln or bit_stable" 2>&1 | tail -3 | tee $OUT/pytest.log
