set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o camp -- python "$R/bench.py" --mode train --model campnet --dtype bf16 --steps 10 --warmup 3 > "$R/$OUT/rocprof_camp.log" 2>&1)
tail -1 $OUT/rocprof_camp.log | cut -c1-300
python tools/rocpd_by_grid.py $(find $OUT/prof -name "*.db" | head -1) $OUT/campnet_by_grid.csv 60
rm -rf $OUT/prof
