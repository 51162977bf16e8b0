#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_train_bf16" -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --mode train --dtype bf16 --steps 4 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_train_bf16.log" 2>&1)
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/rocprof_train_bf16.log
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r02/pytest_gpu_full.log 2>&1
tail -3 gpurun_out/r02/pytest_gpu_full.log
