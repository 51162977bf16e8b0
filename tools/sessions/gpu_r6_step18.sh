#!/bin/bash
# round 6, step 18: 96-frame form as the default: whole GPU suite, smoke, default bench line
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s); timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $OUT/step18_pytest.log; echo "wall $(( $(date +%s) - T0 )) s" >> $OUT/step18_pytest.log; cat $OUT/step18_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/step18_bench.json 2> $OUT/step18_bench.err; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r06/step18_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['frac'], r['frac_algorithmic'], d.get('max_abs_dmel_vs_oracle'), d.get('mcd_vs_oracle'))
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
P
