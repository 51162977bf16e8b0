#!/bin/bash
# quick iteration: parity subset + phase probe + bench (no cpu baseline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x ${PYTEST_ARGS:-} 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 300 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_probe.log
PB=16 timeout 300 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_probe_b16.log
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench.log
