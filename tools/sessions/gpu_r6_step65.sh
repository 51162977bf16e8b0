# conv1d_bf16_kernel (16-byte-unit staging): masked-out LDS writes to a spare row instead of twelve exec-masked blocks; reference: build/exp/libset_amd_burst.so (= before)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "conv1d_bf16" 2>&1 | tail -2
S=16x192x768x9x800,16x768x192x9x800,16x192x384x5x800,32x256x1024x9x800,32x1024x256x9x800,32x192x192x5x800,32x256x256x5x800,32x256x512x3x800,32x512x256x3x800
for i in 1 2; do
echo "== before"; SET_AMD_LIB=$PWD/build/exp/libset_amd_burst.so SHAPES=$S python tools/small_conv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
echo "== spare row"; SHAPES=$S python tools/small_conv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
done
