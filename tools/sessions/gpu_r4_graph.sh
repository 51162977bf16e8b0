#!/bin/bash
# round 4: the graphed training step -- bit-identity test, then the training benches (spec_denoiser bf16 / f32, CampNet bf16) graphed and eager
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_graph.log; : > $OUT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "graphed" 2>&1 | grep -v Warning | tail -25 >> $OUT
for cfg in "spec_denoiser bf16" "spec_denoiser f32" "campnet bf16"; do set -- $cfg
  for g in 1 0; do
    SET_AMD_GRAPH_STEP=$g timeout 600 python bench.py --mode train --model $1 --dtype $2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4_train_$1_$2_graph$g.json
    python - <<PY >> $OUT
import json
d = json.load(open("gpurun_out/r4_train_$1_$2_graph$g.json"))
print("$1 $2 graph=$g: %.2f ms/step, %.0f samples/s, host enqueue %.2f ms, replays %s, loss %.5f, frac %.3f" % (d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"], d.get("graph_replays"), d["loss"], d["roofline"]["frac"]))
PY
  done
done
cat $OUT
