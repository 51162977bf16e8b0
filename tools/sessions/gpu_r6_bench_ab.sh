#!/bin/bash
# round 6: bench.py's own headline number, direct form vs Winograd form of the split-operand kernel, back to back on one box (3 x 100 steps each, no extras)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
F="--steps 5 --warmup 2 --no-cpu-baseline --no-native-fp32 --no-bf16-loop --no-bf16x3-loop --no-quality --no-secondary"
for rep in 1 2; do
for w in 0 1; do
  SET_AMD_X3_WINO=$w timeout 600 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('WINO=$w rep $rep: %.0f frames/s  %.2f ms/step  launch %.4f ms  %s  traffic x%s  mfma busy %s' % (d['value'], d['ms_per_step'], r['launch_ms'], r['kernel'][:28], r['traffic_over_algorithmic_bytes'], r['pmc_mfma_busy_frac_of_simd_cycles']))"
done; done | tee $OUT/bench_ab_wino.log
