#!/bin/bash
# round 5, step 19: LayerNorm backward, 32 frames x 16 channel groups in 512-thread blocks (two per CU) against the 16 x 16 blocks
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s19; mkdir -p $OUT
for v in "" 1; do echo "SET_AMD_LNB_512=${v:-unset}"; ( [ -n "$v" ] && export SET_AMD_LNB_512=1; timeout 200 python tools/ln_bwd_probe.py 2>&1 | grep -v amdgpu.ids ); done | tee $OUT/ln_bwd_512.log
SET_AMD_LNB_512=1 timeout 300 python -m pytest tests/test_gpu_training.py -q -x -k "small_op or match_reference" 2>&1 | tail -2 | tee -a $OUT/ln_bwd_512.log
