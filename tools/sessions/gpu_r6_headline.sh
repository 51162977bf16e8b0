#!/bin/bash
# round 6: headline evidence only (PMC passes of the stack kernel -> pmc_x3.json, kernel stats of the inference loop, the default bench line quoting them)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
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
bash tools/sessions/gpu_r6_bench.sh
