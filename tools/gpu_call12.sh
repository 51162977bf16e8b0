#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_train_bf16" -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --mode train --dtype bf16 --steps 6 --warmup 3 > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_train_bf16.log" 2>&1)
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/rocprof_train_bf16.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_train_f32" -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --mode train --dtype f32 --steps 6 --warmup 3 > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_train_f32.log" 2>&1)
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/rocprof_train_f32.log
