#!/bin/bash
# PMC evidence for the dominant kernel: MFMA busy cycles, wave cycles, waits, clock (separate pass from traffic).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/pmc_util
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_util" -o pmc -- python "$OLDPWD/tools/wino_probe.py" > "$OLDPWD/$OUT/pmc_util.log" 2>&1 )
python - <<'PY'
import csv, glob, collections, json
f = glob.glob('gpurun_out/pmc_util/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    for key in ('diffnet_stack_wino_kernel', 'diffnet_stack_kernel', 'diffnet_layer_kernel'):
        if key in r['Kernel_Name'] and 'pack_' not in r['Kernel_Name']:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0           # summed over the 8 XCDs
    m['derived_cycles_per_launch'] = cyc
    m['derived_mfma_busy_frac_of_simd_cycles'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc) if cyc else None
    out[k] = m
json.dump(out, open('gpurun_out/pmc_util_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT/pmc_util -name "*kernel_trace.csv" -delete
