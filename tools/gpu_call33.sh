#!/bin/bash
# round-end evidence, part 1: the default bench line (full CPU protocol) + small-batch / e2e numbers
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python bench.py 2>$OUT/bench_default.err | tail -1) > $OUT/bench_default.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/r02/bench_default.json').read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'], j.get('native_fp32_loop',{}).get('value'), (j.get('bf16_operand_loop') or {}).get('value'))
P
SIZES=1x800,2x800,4x800,8x800,16x800,32x800,64x800 timeout 300 python tools/latency_probe.py 2>&1 | grep "B=" > $OUT/latency.log; cat $OUT/latency.log
(EB=64 timeout 300 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e.log; cat $OUT/e2e.log | cut -c1-400
(EB=1 ESTEPS=8 timeout 200 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e_b1_8.log; cat $OUT/e2e_b1_8.log | cut -c1-300
(EB=1 ESTEPS=100 timeout 200 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e_b1_100.log; cat $OUT/e2e_b1_100.log | cut -c1-300
