#!/bin/bash
# Round 4 evidence run on FINAL sources: PMC passes (separate runs, --kernel-trace only) of the split-operand stack kernel and of the
# fused bf16 layer groups at B=32, T=800 (each JSON carries the sha256 of its kernel sources: bench.py quotes the traffic only when
# it matches), then the kernel trace of the default bench command.  Outputs under gpurun_out/r04/ (copied to profiles/r04_*).
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_*
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_fetch" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_write" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_write.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_write -name "*counter_collection.csv" | head -1)
U=$(find $OUT/pmc_x3_util -name "*counter_collection.csv" | head -1)
python tools/pmc_x3_summary.py "$F" "$W" "$U" $OUT/pmc_x3.json | tail -30
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*.csv" -delete
TILE=128 NLS=10 bash tools/sessions/gpu_pmc_bf16_layers.sh > $OUT/pmc_bf16_t128.log 2>&1; cp gpurun_out/pmc_bf16_layers_128.json $OUT/pmc_bf16_layers.json
TILE=64 NLS=5 bash tools/sessions/gpu_pmc_bf16_layers.sh > $OUT/pmc_bf16_t64.log 2>&1; cp gpurun_out/pmc_bf16_layers_64.json $OUT/pmc_bf16_layers_tile64.json
rm -rf $OUT/prof_bench
(cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_bench" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-native-fp32 --no-bf16x3-loop --no-secondary --no-quality --steps 3 > "$R/$OUT/rocprof_bench.log" 2>&1)
tail -1 $OUT/rocprof_bench.log | cut -c1-300
python tools/rocpd_summary.py $(find $OUT/prof_bench -name "*.db" | head -1) $OUT/kernel_stats.csv 2>&1 | tail -3
head -8 $OUT/kernel_stats.csv
rm -rf $OUT/prof_bench
cat $OUT/pmc_bf16_layers.json
