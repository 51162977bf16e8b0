# conv1d_bf16_kernel: 16-byte-unit input staging against one frame per load (bit identity + times per shape)
python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "unit_staging or conv1d_bf16" 2>&1 | tail -5
for u in 0 1 0 1; do echo "== SET_AMD_CONV_BF16_UNITS=$u"; SET_AMD_CONV_BF16_UNITS=$u SHAPES=16x192x768x9x800,16x768x192x9x800,16x192x384x5x800,32x256x1024x9x800,32x1024x256x9x800,32x192x192x5x800,32x256x256x5x800 python tools/small_conv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; done
