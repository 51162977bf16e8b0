# DiffNet bf16 stack: per-layer addresses as integers in the forward loop and the backward sweep (host time): signature, tests, same-box A/B by swapping the file
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
for v in old new; do
  cp build/ab/autograd_ops_$v.py speech-editing-toolkit_amd/autograd_ops.py
  python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss')})"
done; done
cp build/ab/autograd_ops_new.py speech-editing-toolkit_amd/autograd_ops.py
