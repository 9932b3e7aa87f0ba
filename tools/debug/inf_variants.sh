#!/bin/bash
# tuning: rebuild the INFLATE kernel ON THE GPU BOX with other constants (NAME=VALUE,NAME=VALUE ... per variant) and print
# the phase profile of each
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc
cp inflate_kernels.hip /tmp/inf_keep.hip
for cfg in "$@"; do
  cp /tmp/inf_keep.hip inflate_kernels.hip
  for kv in ${cfg//,/ }; do
    k=${kv%%=*}; v=${kv##*=}
    sed -i -e "s/^constexpr int $k = [0-9]*;/constexpr int $k = $v;/" inflate_kernels.hip
  done
  make -s 2>&1 | grep -E "error"
  echo "== $cfg"
  (cd $GRAFT_REPO_ROOT && CCT_INF_PROF=1 timeout -k 10 200 python tools/prof_codec.py --reps 1 --what dec 2>&1 | grep "inflate prof" | head -1
   timeout -k 10 200 python tools/prof_codec.py --reps 3 --what dec 2>&1 | grep "^dec" | tail -1)
done
cp /tmp/inf_keep.hip inflate_kernels.hip
make -s
