#!/bin/bash
# round 5, step 3c: the fp32 fast-path store, hypothesis = data-register hazard of a 16-byte buffer store with an SGPR soffset (the compiler
# rewrites the store's data registers in the next instruction; it inserts the wait state itself only when soffset is NOT a register):
# masked store + s_nop 1 behind it (3), + full wait behind it (4), whole offset in the VGPR / soffset = 0 (5); then the reverted reductions A/B
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s3; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for v in masked3 masked4 masked5 masked; do
  echo "== $v" | tee -a $OUT/stability3.log
  SET_AMD_LIB=$R/build/exp/libset_amd_$v.so DTYPE=f32 REPEAT=5 TRACE=1 timeout 400 python tools/grad_stability_probe.py 2>&1 | grep "FIRST\|differing .* of .* elements\|conv calls\|bit-identical\|parameters with" | cut -c1-300 | tee -a $OUT/stability3.log
done
for model in spec_denoiser campnet; do
  timeout 300 python bench.py --mode train --model $model --dtype bf16 --steps 40 --warmup 10 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('reverted reductions $model: %.3f ms/step, host enqueue %.2f ms, loss %.6f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss']))" | tee -a $OUT/reduce_ab.log
done
