#!/bin/bash
# round 6, evidence run on FINAL sources -> gpurun_out/r06/ (copied to profiles/r06_* by the caller; profiles/ is not merged back from the box):
#   1. per-kernel stats + by-stream timeline of the three training steps (bf16 spec_denoiser / CampNet, fp32 spec_denoiser)
#   2. PMC passes (separate --pmc runs, kernel trace only) of the three training steps -> pmc_train.json (HBM bytes per launch of the layer /
#      conv kernels -- the fp32 step included --, launches per step, sha256 of the kernel sources)
#   3. PMC passes of the headline kernel -> pmc_x3.json; of the bf16 layer groups -> pmc_bf16_layers.json; kernel stats of the inference loop
#   4. full -m gpu suite, smoke(), the driver's default bench command with its wall time (quoting the PMC files of this very run)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
pmc() {  # tag model counter dtype
  rm -rf $OUT/pmc_$1_$3
  (cd /tmp && timeout 300 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d "$R/$OUT/pmc_$1_$3" -o pmc -- python "$R/bench.py" --mode train --model $2 --dtype ${4:-bf16} --steps 3 --warmup 2 > "$R/$OUT/pmc_$1_$3.log" 2>&1)
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
FF=$(pmc f32 spec_denoiser FETCH_SIZE f32); WF=$(pmc f32 spec_denoiser WRITE_SIZE f32)
python tools/pmc_train_summary.py $OUT/pmc_train.json spec_denoiser_bf16 "$FS" "$WS" $OUT/train_bf16_kernel_stats.csv campnet_bf16 "$FC" "$WC" $OUT/campnet_bf16_kernel_stats.csv \
    spec_denoiser_f32 "$FF" "$WF" $OUT/train_f32_kernel_stats.csv | head -40
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cp $OUT/pmc_train.json profiles/r06_pmc_train.json   # so that the bench line below quotes it
TILE=128 NLS=10 bash tools/sessions/gpu_pmc_bf16_layers.sh > $OUT/pmc_bf16_t128.log 2>&1; cp gpurun_out/pmc_bf16_layers_128.json $OUT/pmc_bf16_layers.json; cp $OUT/pmc_bf16_layers.json profiles/r06_pmc_bf16_layers.json   # quoted by the bench line below
# headline kernel PMC + kernel stats of the inference loop
rm -rf $OUT/pmc_x3_* $OUT/trace_infer
for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_$c" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_$c.log" 2>&1); done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_WRITE_SIZE -name "*counter_collection.csv" | head -1); U=$(find $OUT/pmc_x3_util -name "*counter_collection.csv" | head -1)
python tools/pmc_x3_summary.py "$F" "$W" "$U" $OUT/pmc_x3.json | grep -A3 "traffic_bytes\|mfma_busy" | head -12
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cp $OUT/pmc_x3.json profiles/r06_pmc_x3.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace_infer" -o infer -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-native-fp32 --no-bf16-loop --no-bf16x3-loop --no-quality --no-secondary > "$R/$OUT/trace_infer.log" 2>&1)
S=$(find $OUT/trace_infer -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv && head -5 $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/trace_infer
T0=$(date +%s); timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.log; echo "wall $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/smoke.log; cat $OUT/smoke.log
bash tools/sessions/gpu_r6_bench.sh
# soak of the persistent kernels (every launch compared bit for bit with the first; incl. the Winograd form on 64- / 96-frame tiles)
SOAK_N=200 timeout 900 python tools/soak_x3_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak.log | cut -c1-200
