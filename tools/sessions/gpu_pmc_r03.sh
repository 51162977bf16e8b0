#!/bin/bash
# Round 3: PMC passes (HBM traffic, MFMA utilisation; separate runs, --kernel-trace only) of the SHIPPED split-operand stack
# kernel at B=32, T=800 -> gpurun_out/r03/pmc_x3.json (copied to profiles/r03_pmc_x3.json), then the kernel trace of the default
# bench command -> gpurun_out/r03/kernel_stats.csv
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r03; mkdir -p $OUT; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_*
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_fetch" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_write" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_write.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_write -name "*counter_collection.csv" | head -1)
U=$(find $OUT/pmc_x3_util -name "*counter_collection.csv" | head -1)
python tools/pmc_x3_summary.py "$F" "$W" "$U" $OUT/pmc_x3.json | tail -40
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*kernel_trace.csv" -delete
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*counter_collection.csv" -delete
rm -rf $OUT/prof_bench
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_bench" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-bf16-loop --no-native-fp32 --no-bf16x3-loop --steps 3 > "$R/$OUT/rocprof_bench.log" 2>&1)
tail -1 $OUT/rocprof_bench.log | cut -c1-400
python tools/rocpd_summary.py $(find $OUT/prof_bench -name "*.db" | head -1) $OUT/kernel_stats.csv 2>&1 | tail -3
head -8 $OUT/kernel_stats.csv
rm -rf $OUT/prof_bench
