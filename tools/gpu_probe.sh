#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== phase probe"
timeout 300 python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_probe.log
echo "== counters"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|TCC_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|[A-Za-z]*MFMA[A-Za-z0-9_]*)\b" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt; grep -iE "mfma|GUI_ACTIVE|FETCH_SIZE|WRITE_SIZE|BUSY_CYCLES|WAVE_CYCLES|WAIT_INST_ANY|WAIT_ANY|ACTIVE_INST" $OUT/counters.txt | tr '\n' ' '
echo
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && PB=32 PT=800 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_$tag" -o pmc -- python "$OLDPWD/tools/phase_probe.py" > "$OLDPWD/$OUT/pmc_$tag.log" 2>&1 )
  f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "-- $set -> $f"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'diffnet_layer' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'n=%d' % len(v), 'mean=%.4g' % (sum(v) / len(v)), 'last=%.4g' % v[-1])
PY
done
