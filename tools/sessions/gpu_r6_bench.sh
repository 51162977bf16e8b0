#!/bin/bash
# round 6: the driver's default bench command -> profiles/r06_bench.json (+ a one-screen digest)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s); timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_wall.log
tail -3 $OUT/bench.err
cp $OUT/bench.json profiles/r06_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06/bench.json") if l.startswith("{")][-1])
r = d["roofline"]
print("headline %.0f frames/s, %.2f ms/step, kernel %s, launch %.3f ms, frac %.3f (alg %.3f), traffic %s (x%.2f), mfma busy %s" % (
    d["value"], d["ms_per_step"], r["kernel"], r["launch_ms"], r["frac"], r["frac_algorithmic"], r["traffic"], r["traffic_over_algorithmic_bytes"] or 0, r["pmc_mfma_busy_frac_of_simd_cycles"]))
print("quality", d.get("mcd_vs_oracle"), d.get("max_abs_dmel_vs_oracle"), d.get("quality_sample"))
for k in ("native_fp32_loop", "bf16x3_operand_loop", "bf16_operand_loop"):
    print(k, d[k]["value"], d[k].get("roofline", {}).get("frac"))
for k in ("train_bf16", "campnet_train_bf16", "train_f32"):
    print(k, d[k].get("ms_per_step"), d[k].get("launches_per_step"), (d[k].get("roofline") or {}).get("kernel"), d[k].get("error"))
print("e2e", d.get("e2e_b64_vocoder"))
c = d["cpu_baseline"]
print("cpu", c["value"], c["cores"], c.get("best_of"), "usable", c.get("host_cpus_usable"), "quota", c.get("host_cgroup_quota_cores"), "os", c.get("host_cpus"))
print("cpu sweep", c.get("thread_sweep_frames_per_s_B8"), "all", c.get("all_cores_multiprocess"), "speedup", d["speedup_vs_cpu_baseline"])
PY
