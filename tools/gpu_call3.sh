#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_dist.py tests/test_gpu_training.py -q -m gpu -p no:cacheprovider -rA -k "tolerance or bit_stable or two_ranks or trainer or accumulation" > gpurun_out/r02/pytest_call3.log 2>&1
grep -E "passed|failed" gpurun_out/r02/pytest_call3.log | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_train_bf16" -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --mode train --dtype bf16 --steps 4 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_train_bf16.log" 2>&1)
ls gpurun_out/r02/prof_train_bf16 | head
