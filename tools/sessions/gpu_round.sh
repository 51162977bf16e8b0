#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace (+ optional PMC traffic passes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/device.txt; rocminfo 2>/dev/null | grep -m1 gfx950 >> $OUT/device.txt
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --gpus 1 --steps ${BENCH_STEPS:-3} --warmup 1 2>&1 | tail -1 | tee $OUT/bench.log
echo "== rocprofv3 kernel trace"
rm -rf $OUT/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > "$OLDPWD/$OUT/rocprof_bench.log" 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" -size +6M -delete
if [ "${PMC:-0}" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_$c
    ( cd /tmp && PB=32 PT=800 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_$c" -o pmc -- python "$OLDPWD/tools/wino_probe.py" > "$OLDPWD/$OUT/pmc_$c.log" 2>&1 )
  done
  python tools/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/traffic_bytes_per_launch.json | tail -12
  find $OUT -name "*kernel_trace.csv" -size +2M -delete
fi
