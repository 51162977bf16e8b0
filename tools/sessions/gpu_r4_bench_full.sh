#!/bin/bash
# the default bench line as the driver runs it (--steps 20 --warmup 5), wall time included
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > gpurun_out/r4_bench_full.json 2> gpurun_out/r4_bench_full.err
echo "rc=$? wall $(( $(date +%s) - T0 )) s"
tail -5 gpurun_out/r4_bench_full.err | grep -v "amdgpu.ids\|Warning\|WeightNorm"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_full.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.2f frac %.3f frac_alg %.3f mcd_vs_oracle %s max|dmel| %s ints %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_algorithmic"], d.get("mcd_vs_oracle"), d.get("max_abs_dmel_vs_oracle"), d.get("integers_equal_vs_oracle")))
print("quality_sample", d.get("quality_sample"))
b = d["bf16_operand_loop"]; print("bf16 loop", b["value"], b["roofline"]["frac"], b["roofline"]["layers_span_ms"], b["mcd_vs_f32_path"], b["roofline"]["kernel"], b["roofline"]["traffic_note"])
for k in ("e2e_b64_vocoder", "train_bf16", "campnet_train_bf16"):
    print(k, json.dumps(d.get(k))[:700])
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"], "traffic", d["roofline"]["traffic_note"])
PY
