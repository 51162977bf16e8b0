#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_x2conv.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_split" 2>&1 | tail -1
for d in f32; do
(timeout 300 python bench.py --mode train --dtype $d --steps 10 --warmup 3 2>&1 | tail -1) > $OUT/bench_train_$d.json
python -c "import json,sys; j=json.loads(open('$OUT/bench_train_$d.json').read()); print('train $d', j['ms_per_step'], j['value'], j.get('host_enqueue_ms_per_step'))"
done
