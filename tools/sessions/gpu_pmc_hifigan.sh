#!/bin/bash
# PMC pass over the HiFi-GAN forward: MFMA busy fraction and LDS bank conflicts per conv kernel instantiation.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/pmc_hifigan
( cd /tmp && HB=16 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_hifigan" -o pmc -- python "$OLDPWD/tools/hifigan_bench.py" > "$OLDPWD/$OUT/pmc_hifigan.log" 2>&1 )
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_hifigan/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'conv1d' not in n: continue
    key = (n.split('(')[0].replace('void ', ''), r.get('Grid_Size', r.get('Grid_Size_X', '')))
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[key] += 1
lines = ["HiFi-GAN V1 forward (B=16, T=800), PMC per conv kernel instantiation x grid; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /"
         " (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"]
tot_busy = tot_cyc = 0.0
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    cyc = d['GRBM_GUI_ACTIVE'] / 8.0
    tot_busy += d['SQ_VALU_MFMA_BUSY_CYCLES']; tot_cyc += cyc
    lines.append("%-30s grid %-9s calls %3d  cycles %5.1f%%  mfma_busy %.1f%%  lds_conflict/lds_active %.2f  wait_inst/wave_cycles %.2f" % (
        k[0], k[1], cnt[k], 0.0, 100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc) if cyc else 0,
        d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1), d['SQ_WAIT_INST_ANY'] / max(d['SQ_WAVE_CYCLES'], 1)))
lines.append("all conv kernels: mfma_busy %.1f%% of the SIMD cycles" % (100 * tot_busy / (1024.0 * tot_cyc)))
open('gpurun_out/pmc_hifigan_summary.txt', 'w').write("\n".join(lines[:14] + lines[-1:]) + "\n")
print("\n".join(lines[:10] + lines[-1:]))
PY
find $OUT/pmc_hifigan -name "*.csv" -size +5M -delete
