#!/bin/bash
# (historical: SET_AMD_BF16_REG_VARIANT / SET_AMD_BF16_T128_NSKR were round-4 measurement switches; the shipped library keeps only the chosen instantiations,
#  the logs are profiles/r04_bf16_ab.log and profiles/r04_t128_exp.log)
# round 4: A/B of the fused-layers kernels at B = 32, T = 800: SET_AMD_BF16_FUSE_TILE = 64 | 128 (tile width), SET_AMD_BF16_REG_VARIANT bit 0 =
# static priority skew, bit 1 = A ring of 8 k-steps (64-frame shape) / double-buffered B fragments in GEMM 1 (128-frame shape):
# bit-identity tests, time per group launch + phase shares
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_bf16_ab.log; : > $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "fused_layer_groups or inference_loop" 2>&1 | tail -5 >> $OUT
for t in ${TILES:-128 64}; do for v in ${VARIANTS:-0 1 2 3}; do
  SET_AMD_BF16_REG_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "fused_layer_groups" 2>&1 | tail -1 >> $OUT
  SET_AMD_BF16_FUSE_TILE=$t SET_AMD_BF16_REG_VARIANT=$v NLS=${NLS:-5,10} timeout 300 python tools/bf16_layers_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT
done; done
cat $OUT
