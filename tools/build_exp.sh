#!/bin/bash
# Experimental builds of the library for A/B measurements: build/exp/libset_amd_<tag>.so, loaded with SET_AMD_LIB=<path>.
# usage: [SRC_OVERRIDE=path] tools/build_exp.sh <tag> <file.hip> [extra hipcc flags...]   -- rebuilds <file.hip> with the flags, links with the
# other objects (compiled once into build/exp/obj/).
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
CS=speech-editing-toolkit_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I include"
mkdir -p build/exp/obj
for f in conv1d conv_x2 resblock_x2 glue diffnet diffnet_x3 train attention attention_fused bf16 diffnet_bf16; do
  if [ ! -f build/exp/obj/$f.o ] || [ $CS/$f.hip -nt build/exp/obj/$f.o ] || [ $CS/common.h -nt build/exp/obj/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $CS/$f.hip -o build/exp/obj/$f.o &
  fi
done
wait
# SRC_OVERRIDE=<path>: compile that file in place of $CS/$SRC (an older or patched version of the same module)
/opt/rocm/bin/hipcc $FLAGS -I $CS "$@" -c ${SRC_OVERRIDE:-$CS/$SRC} -o build/exp/obj/${SRC%.hip}_$TAG.o
OBJS=""
for f in conv1d conv_x2 resblock_x2 glue diffnet diffnet_x3 train attention attention_fused bf16 diffnet_bf16; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS build/exp/obj/${f}_$TAG.o"; else OBJS="$OBJS build/exp/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o build/exp/libset_amd_$TAG.so
echo build/exp/libset_amd_$TAG.so
