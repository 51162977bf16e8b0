#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -p no:cacheprovider -rA -k "fused or tolerance or inference_loop" > gpurun_out/r02/pytest_call18.log 2>&1
grep -E "passed|failed|^E  |^FAILED" gpurun_out/r02/pytest_call18.log | head
for TILE in 64 128; do
(SET_AMD_BF16_TILE=$TILE timeout 300 python bench.py --no-cpu-baseline --steps 2 2>&1 | tail -1) > gpurun_out/r02/bench_bf16loop_$TILE.json
python - <<PY
import json
d=json.loads(open('gpurun_out/r02/bench_bf16loop_$TILE.json').read())
b=d["bf16_operand_loop"]
print("tile $TILE: bf16 loop %.0f frames/s, %.3f ms/step-layers, hbm frac %.3f mcd %.3f" % (b["value"], b["roofline"]["layers_span_ms"], b["roofline"]["frac"], b["mcd_vs_f32_path"]))
PY
(SET_AMD_BF16_TILE=$TILE timeout 300 python bench.py --mode train --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1) > gpurun_out/r02/bench_train_bf16_$TILE.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/bench_train_bf16_$TILE.log
done
