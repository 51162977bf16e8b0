# 16-byte forms of the elementwise training kernels (conv epilogue backward, activations, add + mask, dropout): bit-level signature, tests, same-box A/B
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_training.py tests/test_gpu_bf16.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$PWD/speech-editing-toolkit_amd/libset_amd_before.so; else unset SET_AMD_LIB; fi
  for m in spec_denoiser campnet; do
  python bench.py --mode train --model $m --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $m', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step')})"
  done
done; done
unset SET_AMD_LIB
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$PWD/speech-editing-toolkit_amd/libset_amd_before.so; else unset SET_AMD_LIB; fi
  python bench.py --mode train --dtype f32 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v f32', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step')})"
done
