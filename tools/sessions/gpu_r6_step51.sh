#!/bin/bash
# round 6, step 51: bf16 layer groups (128-frame tiles, 10 layers per launch) with a per-wave rate cap in front of every k-step's MFMAs (s_sleep 1 / 2 / 3), interleaved with the shipped form
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for tag in t128s0 t128s1 t128s2 t128s3; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so SET_AMD_BF16_FUSE_TILE=128 NLS=10 REPS=1500 timeout 300 python tools/bf16_layers_probe.py 2>&1 | grep "us per launch" | sed "s/^/$tag: /" | cut -c1-200
  done
done | tee $OUT/bf16_t128_sleep_ab.log
