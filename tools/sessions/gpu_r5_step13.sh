#!/bin/bash
# round 5, step 13: per-kernel stats + by-stream timeline of the fp32 training step (B = 32, T = 800)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s13; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- python "$R/bench.py" --mode train --model spec_denoiser --dtype f32 --steps 8 --warmup 3 > "$R/$OUT/rocprof_f32.log" 2>&1)
db=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $db $OUT/train_f32_kernel_stats.csv > /dev/null 2>&1
python tools/rocpd_timeline.py $db $OUT/train_f32_timeline.csv 2>&1 | grep -v "kernels columns" | tee $OUT/train_f32_timeline.log
rm -rf $OUT/prof
grep '^{' $OUT/rocprof_f32.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 under rocprof: %.3f ms/step' % d['ms_per_step'])"
