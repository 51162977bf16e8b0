#!/bin/bash
# first GPU call of round 2: tests, default bench (new CPU protocol), train bench, HiFi-GAN bench + kernel trace
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/r02/pytest_gpu.log
(timeout 600 python bench.py 2>&1 | tail -5) > gpurun_out/r02/bench_default.log
(timeout 300 python bench.py --mode train --steps 5 --warmup 2 2>&1 | tail -3) > gpurun_out/r02/bench_train_f32.log
(timeout 300 env HSTAGES=1 python tools/hifigan_bench.py 2>&1 | tail -30) > gpurun_out/r02/hifigan_stages.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_hifigan" -o hifi -- python "$GRAFT_REPO_ROOT/tools/hifigan_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_hifigan.log" 2>&1)
find gpurun_out/r02/prof_hifigan -name "*kernel_stats*" | head -3
tail -3 gpurun_out/r02/pytest_gpu.log
cat gpurun_out/r02/bench_train_f32.log | tail -1 | cut -c1-600
