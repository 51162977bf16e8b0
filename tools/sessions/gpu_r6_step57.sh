# quad-interleaved layout of the bf16 tensors between the fused layer kernels and the weight gradients: bit-level signature (must equal
# profiles/r06_train_sha_before.log), the loaders' tests, the training tests, kernel times
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py -q -m gpu -x 2>&1 | tail -4
python -m pytest tests/test_gpu_training.py -q -m gpu -x 2>&1 | tail -4
python tools/wgrad_group_probe.py 2>&1 | grep -v amdgpu.ids | tail -6
python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','host_enqueue_ms_per_step','value','loss')}); print(d['roofline']['kernel'], d['roofline']['launch_ms'])"
