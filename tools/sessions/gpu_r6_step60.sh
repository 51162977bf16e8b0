# forward gate of the bf16 layer kernels as one quotient (3 transcendentals per value): tests, bf16 inference loop A/B, training step A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_training.py -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16 or fused_layer_groups or layer_group" 2>&1 | tail -3
for i in 1 2; do
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$PWD/speech-editing-toolkit_amd/libset_amd_before.so; else unset SET_AMD_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-native-fp32 --no-bf16x3-loop --no-quality --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bf16_operand_loop']; print('$v bf16 loop', round(b['value']), 'frames/s', round(b['ms_per_step'],2), 'ms, frac', round(b['roofline']['frac'],4), 'layers span ms', round(b['roofline']['layers_span_ms'],4), 'mcd', round(b['mcd_vs_f32_path'],4), 'max|d|', round(b['max_abs_dmel_vs_f32_path'],4))"
  python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step','loss')})"
done; done
