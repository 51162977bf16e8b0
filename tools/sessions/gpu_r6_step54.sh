# conv1d_bf16_kernel: all 9 taps in one stage for the 64-row blocks (experiment)
for u in 0 1 0 1; do echo "== SET_AMD_CONV_TG9=$u"; SET_AMD_CONV_TG9=$u SHAPES=16x768x192x9x800,32x1024x256x9x800,16x192x192x9x800 python tools/small_conv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; done
