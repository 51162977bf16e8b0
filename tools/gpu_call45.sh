#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "x3_stack" 2>&1 | tail -1
for k in 1 2; do (timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --no-native-fp32 --steps 3 2>&1 | tail -1) | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"; done
