#!/bin/bash
# round 6: PMC passes (separate --pmc runs, kernel trace only) of the headline kernel (Winograd form of the split-operand stack) at B=32, T=800 on the
# current sources -> profiles/r06_pmc_x3.json (stamped with the sha256 of its kernel sources); rocprofv3 --kernel-trace --stats of the inference
# loop -> profiles/r06_kernel_stats.csv; then the driver's default bench command -> profiles/r06_bench.json
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_* $OUT/trace_infer
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_fetch" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_write" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_write.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_write -name "*counter_collection.csv" | head -1)
U=$(find $OUT/pmc_x3_util -name "*counter_collection.csv" | head -1)
python tools/pmc_x3_summary.py "$F" "$W" "$U" $OUT/pmc_x3.json | tail -24
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*.csv" -delete
cp $OUT/pmc_x3.json profiles/r06_pmc_x3.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace_infer" -o infer -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-native-fp32 --no-bf16-loop --no-bf16x3-loop --no-quality --no-secondary > "$R/$OUT/trace_infer.log" 2>&1)
S=$(find $OUT/trace_infer -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -12 "$S" | cut -c1-220 && cp "$S" profiles/r06_kernel_stats.csv
find $OUT/trace_infer -name "*.csv" -size +2M -delete
T0=$(date +%s); timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_wall.log
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
