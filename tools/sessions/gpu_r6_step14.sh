#!/bin/bash
# round 6, step 14: host cost of the leaf-stream operand accounting: training steps with it (default) and without (SET_AMD_LEAF_KEEP_MB=0), same box
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for mb in 4096 0; do
  for cfg in "campnet bf16" "spec_denoiser bf16" "spec_denoiser f32"; do set -- $cfg
    SET_AMD_LEAF_KEEP_MB=$mb timeout 300 python bench.py --mode train --model $1 --dtype $2 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('keep_mb=$mb rep $rep %-14s %-4s %.3f ms/step  host enqueue %.3f ms' % ('$1', '$2', d['ms_per_step'], d['host_enqueue_ms_per_step']))"
  done
done; done | tee $OUT/leaf_accounting_ab.log
