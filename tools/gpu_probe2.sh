#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_probe.log
PB=16 timeout 300 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_probe_b16.log
