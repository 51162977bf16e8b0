# per-layer bf16 forward: biases through LDS (accumulator starts as 16-byte LDS reads): signature, tests, same-box A/B
mkdir -p gpurun_out
python tools/train_grad_sha.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16 or fused_layer_groups or layer_group" 2>&1 | tail -2
for i in 1 2 3; do
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$PWD/speech-editing-toolkit_amd/libset_amd_before.so; else unset SET_AMD_LIB; fi
  python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss')})"
done; done
unset SET_AMD_LIB
cd /tmp && export TMPDIR=/tmp
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$GRAFT_REPO_ROOT/speech-editing-toolkit_amd/libset_amd_before.so; else unset SET_AMD_LIB; fi
  rm -rf /tmp/prof_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --dtype bf16 --steps 10 --warmup 3 > /tmp/prof_$v.log 2>&1
  S=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; grep "layer_fwd\|layer_bwd" "$S" | cut -c1-140
done
