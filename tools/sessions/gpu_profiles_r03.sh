#!/bin/bash
# Round 3 end state: per-kernel stats (rocprofv3 --kernel-trace --stats) of the two training benches and of the HiFi-GAN V1 forward,
# per-(kernel, grid) traces of the training steps, the per-stage table of the vocoder -> gpurun_out/r03p/ (copied to profiles/r03_*)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r03p; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
prof() {  # name, command...
  local name=$1; shift
  rm -rf $OUT/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- "$@" > "$R/$OUT/rocprof_$name.log" 2>&1)
  local db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $db $OUT/${name}_kernel_stats.csv > /dev/null 2>&1
  python tools/rocpd_by_grid.py $db $OUT/${name}_by_grid.csv 3 | tail -3
  rm -rf $OUT/prof
}
prof train_bf16 python "$R/bench.py" --mode train --dtype bf16 --steps 10 --warmup 3
prof campnet_bf16 python "$R/bench.py" --mode train --model campnet --dtype bf16 --steps 10 --warmup 3
prof hifigan python "$R/tools/hifigan_bench.py"
HSTAGES=1 timeout 200 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\|stage" > $OUT/hifigan_stages.log
tail -17 $OUT/hifigan_stages.log | head -3
timeout 100 python tools/small_conv_probe.py > $OUT/small_conv_probe.log 2>&1
SIZES=1x800,2x800,4x800,8x800,16x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B=" > $OUT/latency.log; cat $OUT/latency.log
