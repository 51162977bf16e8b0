#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "persistent or groups or full_size or full_inference or fused" 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
for cfg in "1 512" "1 576" "1 640" "1 768" "2 256"; do
  set -- $cfg
  echo "== NCB $1 grid $2"
  SET_AMD_STACK_NCB=$1 SET_AMD_STACK_GRID=$2 timeout 120 python tools/stack_probe.py 2>&1 | grep -A1 "persistent\|per-layer" | tee -a gpurun_out/stack_sweep.log
done
