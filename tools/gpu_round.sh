#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Logs land in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx950|Compute Unit" > $OUT/device.txt
nproc >> $OUT/device.txt
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -60 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --gpus 1 --steps ${BENCH_STEPS:-3} --warmup 1 2>&1 | tail -5 | tee $OUT/bench.log
echo "== rocprofv3"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > "$OLDPWD/$OUT/rocprof_bench.log" 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -15 "$f" | tee $OUT/kernel_stats_head.txt; done
# keep the merged output small: drop the raw per-dispatch trace
find $OUT/prof -name "*kernel_trace*" -size +8M -delete
