#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "x3 or split_operand" > gpurun_out/r02/pytest_call25.log 2>&1
grep -v "^$" gpurun_out/r02/pytest_call25.log | tail -25
(timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --steps 2 2>&1 | tail -1) > gpurun_out/r02/bench_x3.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/r02/bench_x3.json').read())
print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['launch_ms'])
P
