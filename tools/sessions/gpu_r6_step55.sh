# conv1d_bf16_kernel: tap loop unrolled (fragment reads of the next tap under the MFMAs of this one); compare with profiles/r06_conv_units_ab.log (units=1)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "unit_staging or conv1d_bf16" 2>&1 | tail -3
for u in 1 2; do SHAPES=16x192x768x9x800,16x768x192x9x800,16x192x384x5x800,32x256x1024x9x800,32x1024x256x9x800,32x192x192x5x800,32x256x256x5x800 python tools/small_conv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75; done
