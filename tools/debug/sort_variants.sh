#!/bin/bash
# tuning: rebuild deflate_kernels.hip ON THE GPU BOX with other sort tile sizes (elements per lane and tile) and print the kernel times
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc
for e in "$@"; do
  make -s CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCCT_SORT_E=$e" build/deflate_kernels.hip.o -B 2>&1 | grep error
  make -s 2>&1 | grep error
  echo "== E=$e"
  (cd $GRAFT_REPO_ROOT && bash tools/gpu_prof_codec.sh --what enc 2>&1 | grep -E "sort_pass|match_kernel" ; python -m pytest tests/test_gpu_deflate.py -q -m gpu -x -k "random_alphabets or token_payloads" 2>&1 | tail -1)
done
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc && make -s build/deflate_kernels.hip.o -B && make -s
