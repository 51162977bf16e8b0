#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py tests/test_gpu_campnet.py tests/test_gpu_dist.py -q -m gpu -p no:cacheprovider -rA > gpurun_out/r02/pytest_call10.log 2>&1
grep -E "passed|failed" gpurun_out/r02/pytest_call10.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/r02/pytest_call10.log | head
(timeout 300 python bench.py --mode train --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1) > gpurun_out/r02/bench_train_bf16.log
(timeout 300 python bench.py --mode train --dtype f32 --steps 10 --warmup 3 2>&1 | tail -1) > gpurun_out/r02/bench_train_f32.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/bench_train_bf16.log gpurun_out/r02/bench_train_f32.log
