#!/bin/bash
# round 5: PMC passes (separate --pmc runs, kernel trace only) of the headline split-operand stack kernel at B=32, T=800 on the final sources
# -> gpurun_out/r05/pmc_x3.json (-> profiles/r05_pmc_x3.json, stamped with the sha256 of its kernel sources), then the driver's default
# bench command so that the line quotes it
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_*
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_fetch" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_write" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_write.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_write -name "*counter_collection.csv" | head -1)
U=$(find $OUT/pmc_x3_util -name "*counter_collection.csv" | head -1)
python tools/pmc_x3_summary.py "$F" "$W" "$U" $OUT/pmc_x3.json | tail -12
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*.csv" -delete
cp $OUT/pmc_x3.json profiles/r05_pmc_x3.json
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_wall.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench.json") if l.startswith("{")][-1])
print("headline %.0f frames/s, %.2f ms/step, launch %.3f ms, frac %.3f (alg %.3f), traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["frac_algorithmic"], d["roofline"]["traffic"]))
for k in ("bf16_operand_loop",):
    print(k, d[k]["value"], d[k].get("roofline", {}).get("frac"), d[k].get("roofline", {}).get("traffic"))
for k in ("train_bf16", "campnet_train_bf16", "train_f32"):
    print(k, d[k]["ms_per_step"], d[k].get("launches_per_step"), (d[k].get("roofline") or {}).get("kernel"))
print("e2e", d.get("e2e_b64_vocoder"))
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
