#!/bin/bash
# round 5, step 25: the ragged (T = 77, T_txt = 19, three padded tails) and short (T = 7, T_txt = 3) whole-model goldens through the default, split-operand and Winograd paths
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s25; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -s -k "ragged or short" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/pytest.log
