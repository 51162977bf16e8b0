#!/bin/bash
# PMC passes (HBM traffic, MFMA utilisation) + kernel trace for the split-operand stack kernel at B=32, T=800
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_*
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_fetch" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_write" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_write.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_util" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_util.log" 2>&1)
F=$(find $OUT/pmc_x3_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_x3_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" $OUT/traffic_x3.json | tail -12
python - <<'PY'
import csv, glob, collections, json
f = glob.glob('gpurun_out/r02/pmc_x3_util/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'diffnet_stack_x3_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in acc.items()}
cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
m['derived_cycles_per_launch'] = cyc
m['derived_mfma_busy_frac_of_simd_cycles'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc) if cyc else None
json.dump({'diffnet_stack_x3_kernel<SplitF16x2>': m}, open('gpurun_out/r02/pmc_x3_util_summary.json', 'w'), indent=1)
print(json.dumps(m, indent=1))
PY
find $OUT/pmc_x3_fetch $OUT/pmc_x3_write $OUT/pmc_x3_util -name "*kernel_trace.csv" -delete
# kernel trace of the default bench command (short)
rm -rf $OUT/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_bench" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-bf16-loop --no-native-fp32 --steps 2 > "$R/$OUT/rocprof_bench.log" 2>&1)
tail -1 $OUT/rocprof_bench.log | cut -c1-300
python tools/rocpd_summary.py $(find $OUT/prof_bench -name "*.db" | head -1) $OUT/kernel_stats.csv 2>&1 | tail -3
head -6 $OUT/kernel_stats.csv
