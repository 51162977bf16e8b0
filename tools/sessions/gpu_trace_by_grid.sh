#!/bin/bash
# Per-(kernel, grid) time of a training step: rocprofv3 kernel trace of `bench.py --mode train` -> gpurun_out/trace/<model>_by_grid.csv
# usage: tools/sessions/gpu_trace_by_grid.sh [campnet|spec_denoiser] [bf16|f32] [rows to print]
set -u
MODEL=${1:-campnet}; DT=${2:-bf16}; TOP=${3:-60}
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/trace; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o t -- python "$R/bench.py" --mode train --model $MODEL --dtype $DT --steps 10 --warmup 3 > "$R/$OUT/rocprof_$MODEL.log" 2>&1)
tail -1 $OUT/rocprof_$MODEL.log | cut -c1-300
python tools/rocpd_by_grid.py $(find $OUT/prof -name "*.db" | head -1) $OUT/${MODEL}_${DT}_by_grid.csv $TOP
rm -rf $OUT/prof
