# word ids through a lookup table, loss total as one stack + sum: signature, tests, step
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py tests/test_gpu_campnet.py -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss','launches_per_step')})"
done
python tools/aten_sites.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-200
