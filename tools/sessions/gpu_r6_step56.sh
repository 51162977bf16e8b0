# how much of a training step's host time is the HIP runtime itself: rocprofv3 --hip-runtime-trace --stats of the bf16 spec_denoiser / CampNet steps
mkdir -p gpurun_out/r06; cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for m in spec_denoiser campnet; do
  rm -rf gpurun_out/r06/hip_$m
  (cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d "$R/gpurun_out/r06/hip_$m" -o h -- python "$R/bench.py" --mode train --model $m --dtype bf16 --steps 20 --warmup 3 > "$R/gpurun_out/r06/hip_$m.log" 2>&1)
  S=$(find gpurun_out/r06/hip_$m -name "*hip_api_stats.csv" | head -1)
  echo "== $m (20 + 3 steps; + model build / warm-up calls)"; tail -1 gpurun_out/r06/hip_$m.log | cut -c1-300
  [ -n "$S" ] && head -14 "$S" | cut -c1-160
  find gpurun_out/r06/hip_$m -name "*.csv" ! -name "*stats*" -delete
done
