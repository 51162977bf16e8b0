# backward sweep: dx rows / dskip copy quad-interleaved fp32 (SET_AMD_BWD_Q4=0|1): signature, tests, same-box A/B
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
for v in 0 1; do
  SET_AMD_BWD_Q4=$v python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q4=$v train', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss')}, d['roofline']['kernel'], round(d['roofline']['launch_ms']*1e3,1), 'us')"
done; done
