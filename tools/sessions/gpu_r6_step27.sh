#!/bin/bash
# round 6, step 27: ResBlock-pair kernel with 128-frame wave tiles (every weight fragment meets 4 column blocks of one wave): V1 forward A/B, stages
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN V1" | sed 's/^/shipped: /'
  SET_AMD_LIB=$PWD/build/exp/libset_amd_rpwide.so timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN V1" | sed 's/^/wide-N:  /'
done | tee $OUT/rp_wide_ab.log
HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "stage" > $OUT/rp_stages_shipped.log
HSTAGES=1 SET_AMD_LIB=$PWD/build/exp/libset_amd_rpwide.so timeout 300 python tools/hifigan_bench.py 2>&1 | grep "stage" > $OUT/rp_stages_wide.log
paste -d'|' <(cut -c1-75 $OUT/rp_stages_shipped.log) <(cut -c50-75 $OUT/rp_stages_wide.log)
