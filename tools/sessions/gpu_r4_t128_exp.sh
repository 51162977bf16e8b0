#!/bin/bash
# (historical: SET_AMD_BF16_REG_VARIANT / SET_AMD_BF16_T128_NSKR were round-4 measurement switches; the shipped library keeps only the chosen instantiations,
#  the logs are profiles/r04_bf16_ab.log and profiles/r04_t128_exp.log)
# round 4: where the time of the 128-frame fused-layers kernel goes -- measurement builds (tools/build_exp.sh t128eN diffnet_bf16.hip
# -DSET_T128_EXP=N; bit 0: no weight-fragment refills, bit 1: no B-fragment reads, bit 2: no skip traffic between the layers, bit 3: plain
# instead of streaming accesses of the private skip copy) next to the product build
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/r4_t128_exp.log; : > $OUT
export SET_AMD_BF16_FUSE_TILE=128 NLS=10
for v in ${VARIANTS:-0 1}; do
  echo "== product build, variant $v" >> $OUT
  SET_AMD_BF16_REG_VARIANT=$v timeout 300 python tools/bf16_layers_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT
  for e in ${EXPS:-4 7}; do
    [ -f build/exp/libset_amd_t128e$e.so ] || continue
    echo "== SET_T128_EXP=$e, variant $v" >> $OUT
    SET_AMD_LIB=build/exp/libset_amd_t128e$e.so SET_AMD_BF16_REG_VARIANT=$v timeout 300 python tools/bf16_layers_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
cat $OUT
