#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_x2conv.py -x -q -s 2>&1 | grep "convT\|passed\|failed\|Error\|assert" | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q -m gpu -k "hifigan or vocoder or training_losses" 2>&1 | tail -2
HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN\|  up" > $OUT/hifigan_up.log; cat $OUT/hifigan_up.log
(timeout 300 python bench.py --mode train --dtype f32 --steps 10 --warmup 3 2>&1 | tail -1) | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('train f32', j['ms_per_step'])"
