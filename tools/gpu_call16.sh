#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
for G in 1 2 4; do
  (SET_AMD_GROUPS=$G timeout 300 python bench.py --no-cpu-baseline --steps 2 2>&1 | tail -1) > gpurun_out/r02/bench_g$G.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r02/bench_g$G.json').read())
b=d["bf16_operand_loop"]
print("groups $G: f32 %.0f frames/s | bf16 loop %.0f frames/s, %.2f ms/step-layers, hbm frac %.3f" % (d["value"], b["value"], b["roofline"]["layers_span_ms"], b["roofline"]["frac"]))
PY
done
