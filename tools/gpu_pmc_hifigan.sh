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
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:8]:
    cyc = d['GRBM_GUI_ACTIVE'] / 8.0
    print("%-30s grid %-9s calls %3d  mfma_busy %.1f%%  lds_conflict/lds_active %.2f  wait_inst/wave_cycles %.2f" % (
        k[0], k[1], cnt[k], 100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc) if cyc else 0,
        d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1), d['SQ_WAIT_INST_ANY'] / max(d['SQ_WAVE_CYCLES'], 1)))
PY
find $OUT/pmc_hifigan -name "*.csv" -size +5M -delete
