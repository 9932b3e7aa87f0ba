#!/bin/bash
# tuning: rebuild decode_kernels.hip with phase stamps ON THE GPU BOX and print the phase times of one decode workgroup
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc
make -s CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCCT_DEC_PROF" build/decode_kernels.hip.o -B 2>&1 | grep error
make -s 2>&1 | grep error
cd $GRAFT_REPO_ROOT && timeout -k 10 200 python tools/prof_codec.py --reps 2 --what dec 2>&1 | grep -E "decode prof|^dec" | tail -4
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc && make -s build/decode_kernels.hip.o -B && make -s
