# bf16 conv with a pre-activation output (pre-LN FFN node: conv + activation in one launch): test, signature, A/B by SET_AMD_CONV_PRE_OUT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "pre_activation" 2>&1 | tail -3
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py tests/test_gpu_campnet.py -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do for v in 0 1; do for m in spec_denoiser campnet; do
  SET_AMD_CONV_PRE_OUT=$v python bench.py --mode train --model $m --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pre_out=$v $m', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss')})"
done; done; done
