#!/bin/bash
# Round 4 end state: per-kernel stats (rocprofv3 --kernel-trace --stats) of the two bf16 training steps, the per-stage table of the vocoder,
# the small-batch latency table -> gpurun_out/r04p/ (copied to profiles/r04_*)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- "$@" > "$R/$OUT/rocprof_$name.log" 2>&1)
  local db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $OUT/${name}_kernel_stats.csv > /dev/null 2>&1
  rm -rf $OUT/prof
  head -6 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
prof train_bf16 python "$R/bench.py" --mode train --dtype bf16 --steps 10 --warmup 3
prof campnet_bf16 python "$R/bench.py" --mode train --model campnet --dtype bf16 --steps 10 --warmup 3
HSTAGES=1 timeout 200 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\|stage" > $OUT/hifigan_stages.log; head -1 $OUT/hifigan_stages.log
SIZES=1x800,2x800,4x800,8x800,16x800,32x800,64x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B=" > $OUT/latency.log; cat $OUT/latency.log
