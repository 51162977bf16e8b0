#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "boundary or full_size or full_inference or persistent_equals or f16x2" 2>&1 | tail -3
for v in 1 0 1 0; do
(SET_AMD_BOUNDARY_X2=$v timeout 300 python bench.py --no-cpu-baseline --no-bf16-loop --no-native-fp32 --steps 3 2>&1 | tail -1) | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('boundary_x2=$v', j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"
done
SIZES=1x800,4x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="
SET_AMD_BOUNDARY_X2=0 SIZES=1x800,4x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="
