#!/bin/bash
# tuning: dfl_match_kernel without its scattered result store (CCT_MATCH_PROBE=1; results invalid) against the real one
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc
for v in 1 0; do
  if [ $v = 0 ]; then make -s build/deflate_kernels.hip.o -B 2>&1 | grep error; else make -s CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCCT_MATCH_PROBE=$v" build/deflate_kernels.hip.o -B 2>&1 | grep error; fi
  make -s 2>&1 | grep error
  echo "== probe $v"
  (cd $GRAFT_REPO_ROOT && bash tools/gpu_prof_codec.sh --what enc 2>&1 | grep -E "match_kernel")
done
