#!/bin/bash
# round 4, checkpoint 2: full GPU suite, sustained time of the split-operand stack kernel (tools/power_probe.py x3), default bench line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_pytest_gpu.log
timeout 300 python tools/power_probe.py x3 bf128 2>&1 | grep "^{\"kernel" > gpurun_out/r4_power_x3.log
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r4_bench_quick.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_quick.json"))
print("headline %.0f frames/s %.2f ms/step; x3 launch %.4f ms frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
for k in ("native_fp32_loop", "bf16x3_operand_loop", "bf16_operand_loop"):
    b = d.get(k) or {}
    print(k, {kk: b.get(kk) for kk in ("value", "ms_per_step", "mcd_vs_f32_path")}, (b.get("roofline") or {}).get("frac"), (b.get("roofline") or {}).get("layers_span_ms"))
PY
cat gpurun_out/r4_power_x3.log gpurun_out/r4_pytest_gpu.log
