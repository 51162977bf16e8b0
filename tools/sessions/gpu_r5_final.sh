#!/bin/bash
# round 5, evidence run on FINAL sources -> gpurun_out/r05/ (copied to profiles/r05_*):
#   1. PMC passes (separate --pmc runs, kernel trace only) of the two bf16 training steps -> pmc_train.json (HBM bytes per launch of the layer /
#      conv kernels, launches per step, sha256 of the kernel sources); PMC of the bf16 layer groups -> pmc_bf16_layers.json
#   2. per-kernel stats + by-stream timeline of both training steps
#   3. full -m gpu suite, smoke(), the driver's default bench command with its wall time
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
pmc() {  # tag model counter
  rm -rf $OUT/pmc_$1_$3
  (cd /tmp && timeout 300 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d "$R/$OUT/pmc_$1_$3" -o pmc -- python "$R/bench.py" --mode train --model $2 --dtype bf16 --steps 3 --warmup 2 > "$R/$OUT/pmc_$1_$3.log" 2>&1)
  find $OUT/pmc_$1_$3 -name "*counter_collection.csv" | head -1
}
stats() {  # tag model [dtype]
  rm -rf $OUT/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- python "$R/bench.py" --mode train --model $2 --dtype ${3:-bf16} --steps 10 --warmup 3 > "$R/$OUT/rocprof_$1.log" 2>&1)
  local db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $OUT/$1_kernel_stats.csv > /dev/null 2>&1
  python tools/rocpd_timeline.py $db $OUT/$1_timeline.csv 2>&1 | grep -v "kernels columns" | tee $OUT/$1_timeline.log
  rm -rf $OUT/prof
}
stats train_bf16 spec_denoiser
stats campnet_bf16 campnet
stats train_f32 spec_denoiser f32
FS=$(pmc spec spec_denoiser FETCH_SIZE); WS=$(pmc spec spec_denoiser WRITE_SIZE)
FC=$(pmc camp campnet FETCH_SIZE); WC=$(pmc camp campnet WRITE_SIZE)
python tools/pmc_train_summary.py $OUT/pmc_train.json spec_denoiser_bf16 "$FS" "$WS" $OUT/train_bf16_kernel_stats.csv campnet_bf16 "$FC" "$WC" $OUT/campnet_bf16_kernel_stats.csv | head -60
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cp $OUT/pmc_train.json profiles/r05_pmc_train.json   # so that the bench line below quotes it
TILE=128 NLS=10 bash tools/sessions/gpu_pmc_bf16_layers.sh > $OUT/pmc_bf16_t128.log 2>&1; cp gpurun_out/pmc_bf16_layers_128.json $OUT/pmc_bf16_layers.json; cp $OUT/pmc_bf16_layers.json profiles/r05_pmc_bf16_layers.json
for cfg in "spec_denoiser bf16 60" "spec_denoiser f32 20" "campnet bf16 60"; do set -- $cfg; MODEL=$1 DTYPE=$2 STEPS=$3 timeout 600 python tools/leaf_soak_probe.py 2>&1 | grep -v amdgpu.ids | tail -2; done | tee $OUT/leaf_soak.log
T0=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $OUT/pytest_gpu.log; echo "wall $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/smoke.log; cat $OUT/smoke.log
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_wall.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench.json") if l.startswith("{")][-1])
print("headline %.0f frames/s, %.2f ms/step, launch %.3f ms, frac %.3f (alg %.3f), traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["frac_algorithmic"], d["roofline"]["traffic"]))
for k in ("bf16_operand_loop", "native_fp32_loop", "bf16x3_operand_loop"):
    if k in d: print(k, d[k]["value"], d[k].get("roofline", {}).get("frac"))
for k in ("train_bf16", "campnet_train_bf16", "train_f32"):
    print(k, json.dumps(d.get(k))[:700])
print("e2e", d.get("e2e_b64_vocoder"))
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
