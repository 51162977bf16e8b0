#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3 or split_operand or f16x2 or row_split or full_size or ragged or full_inference or timeout" 2>&1 | tail -3
for k in 1 2; do (timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --no-native-fp32 --steps 3 2>&1 | tail -1) | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"; done
