#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r02/bench_default.json
cut -c1-400 gpurun_out/r02/bench_default.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_bench.log" 2>&1)
tail -1 gpurun_out/r02/rocprof_bench.log | cut -c1-300
(timeout 300 python tools/campnet_bench.py 2>&1 | tail -2) > gpurun_out/r02/campnet_f32.log
(timeout 300 env SET_AMD_DTYPE=bf16 python tools/campnet_bench.py 2>&1 | tail -2) > gpurun_out/r02/campnet_bf16.log
cat gpurun_out/r02/campnet_f32.log gpurun_out/r02/campnet_bf16.log | cut -c1-300
(timeout 300 python tools/e2e_bench.py 2>&1 | tail -2) > gpurun_out/r02/e2e.log; cut -c1-400 gpurun_out/r02/e2e.log
