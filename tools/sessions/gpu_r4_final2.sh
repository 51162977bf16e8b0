#!/bin/bash
# round 4, final tree: bf16 loop tests first (stop there on a failure), PMC passes of the layer-group kernels on these sources,
# then soak + full suite + smoke + the driver's default bench command
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "inference_loop or fused_layer" 2>&1 | tail -3 | tee gpurun_out/r4_bf16_loop_tests.log
grep -q "failed" gpurun_out/r4_bf16_loop_tests.log && exit 1
TILE=128 NLS=10 bash tools/sessions/gpu_pmc_bf16_layers.sh > gpurun_out/r04/pmc_bf16_t128.log 2>&1; cp gpurun_out/pmc_bf16_layers_128.json gpurun_out/r04/pmc_bf16_layers.json
TILE=64 NLS=5 bash tools/sessions/gpu_pmc_bf16_layers.sh > gpurun_out/r04/pmc_bf16_t64.log 2>&1; cp gpurun_out/pmc_bf16_layers_64.json gpurun_out/r04/pmc_bf16_layers_tile64.json
cp gpurun_out/r04/pmc_bf16_layers.json profiles/r04_pmc_bf16_layers.json   # (so that the bench line below checks the sha of THIS build)
SOAK_N=${SOAK_N:-50} bash tools/sessions/gpu_r4_final.sh
