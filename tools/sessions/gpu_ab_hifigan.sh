# A/B on ONE box (boxes differ by ~5 %): per-stage times of the HiFi-GAN V1 forward under the settings given as arguments
# usage: tools/sessions/gpu_ab_hifigan.sh "ENV=.. ENV=.." "ENV=.." ...   (an argument "head" = the build/exp/libset_amd_head.so library)
for rep in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg"
    if [ "$cfg" == "head" ]; then cfg="SET_AMD_LIB=build/exp/libset_amd_head.so"; fi
    env $cfg HSTAGES=1 timeout 200 python tools/hifigan_bench.py 2>&1 | grep "ms/forward\|stage [23]" | cut -c1-100
  done
done
