#!/bin/bash
# Build a variant of libcompact_hip.so with extra preprocessor flags (tuning runs: A/B of two builds on one box).
#   tools/ab_build.sh <name> <file.hip> "<extra flags>"   ->  build_ab/libcompact_hip_<name>.so
# Use on the GPU box through tools/ab_run.sh, which puts the variant in the loader's place for one command.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/2023-compact-image-compression_amd/csrc
name=$1; file=$2; flags=$3
mkdir -p $ROOT/build_ab/obj_$name
for f in encode_kernels.hip encode_tiles.hip encode_pipe.hip encode_stream.hip decode_kernels.hip deflate_kernels.hip inflate_kernels.hip packbits_kernels.hip sched_kernels.hip api.cpp api_comm.cpp api_packbits.cpp curve.cpp; do
  if [ "$f" == "$file" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c $C/$f -o $ROOT/build_ab/obj_$name/$f.o
  else
    cp $C/build/$f.o $ROOT/build_ab/obj_$name/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,libcompact_hip.so -o $ROOT/build_ab/libcompact_hip_$name.so $ROOT/build_ab/obj_$name/*.o -lz -lpthread -ldl
rm -rf $ROOT/build_ab/obj_$name
echo built $ROOT/build_ab/libcompact_hip_$name.so
