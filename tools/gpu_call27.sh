#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "x3 or split_operand or f16x2" > gpurun_out/r02/pytest_call27.log 2>&1
grep -v "^$" gpurun_out/r02/pytest_call27.log | grep -v Warn | tail -32
for m in f16x2 bf16x3; do
SET_AMD_SPLIT_OPERAND=$m timeout 200 python tools/x3_phase_probe.py 2>&1 | grep "B=\|ticks"
(SET_AMD_SPLIT_OPERAND=$m timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --steps 2 2>&1 | tail -1) > gpurun_out/r02/bench_$m.json
python - $m <<'P'
import json,sys
j=json.loads(open('gpurun_out/r02/bench_%s.json'%sys.argv[1]).read())
print(sys.argv[1], j['value'], j['ms_per_step'], j['roofline']['launch_ms'])
P
done
