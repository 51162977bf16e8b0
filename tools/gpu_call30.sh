#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_edit_caller.py -x -q -m gpu > gpurun_out/r02/pytest_call30.log 2>&1
tail -5 gpurun_out/r02/pytest_call30.log
(timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --steps 2 2>&1 | tail -1) > gpurun_out/r02/bench_x3.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/r02/bench_x3.json').read())
print(j['value'], j['ms_per_step'], j['dtype']); print(json.dumps(j['roofline'])[:900]); print(j.get('native_fp32_loop'))
P
