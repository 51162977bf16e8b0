#!/bin/bash
# round 4, final tree: soak of the persistent kernels (the stack kernel ships without phase stamps now), full -m gpu suite, smoke(),
# the driver's default bench command with its wall time
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SOAK_N=${SOAK_N:-300} timeout 600 python tools/soak_x3_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r4_soak.log; cat gpurun_out/r4_soak.log
T0=$(date +%s); timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r4_pytest_gpu.log; echo "wall $(( $(date +%s) - T0 )) s" >> gpurun_out/r4_pytest_gpu.log; cat gpurun_out/r4_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r4_smoke.log; cat gpurun_out/r4_smoke.log
bash tools/sessions/gpu_r4_bench_full.sh 2>&1 | cut -c1-900
