#!/bin/bash
# round 4, checkpoint 1: full GPU suite + the bf16 loop line of bench.py with the 128- and 64-frame layer groups
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_pytest_gpu.log
for t in 128 64; do
  SET_AMD_BF16_FUSE_TILE=$t timeout 600 python bench.py --steps 3 --warmup 1 --cpu-baseline off --no-native-fp32 --no-bf16x3-loop 2>/dev/null | tail -1 > gpurun_out/r4_bench_bf16_tile$t.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4_bench_bf16_tile$t.json"))
b = d.get("bf16_operand_loop", {})
print("tile $t: headline %.0f frames/s %.1f ms | bf16 loop %.0f frames/s, %.1f ms/step, layers span %.4f ms, hbm frac %.3f, mcd %.3f" % (
    d["value"], d["ms_per_step"], b.get("value", 0), b.get("ms_per_step", 0), b.get("roofline", {}).get("layers_span_ms", 0), b.get("roofline", {}).get("frac", 0), b.get("mcd_vs_f32_path", 0)))
PY
done
cat gpurun_out/r4_pytest_gpu.log
