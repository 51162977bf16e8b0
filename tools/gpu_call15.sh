#!/bin/bash
set -u
mkdir -p gpurun_out/r02
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -p no:cacheprovider -rA -k "inference_loop" > gpurun_out/r02/pytest_call15.log 2>&1
grep -E "passed|failed|bf16 loop vs|^E  " gpurun_out/r02/pytest_call15.log | head
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/r02/bench_bf16loop.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_bf16loop.json').read())
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("bf16_operand_loop"), indent=1)[:1500])
PY
