#!/bin/bash
# tuning: the bench with the INFLATE kernel at 256 and at 512 lanes per stream, on ONE box (boxes differ by +- 4 %)
cd $GRAFT_REPO_ROOT/2023-compact-image-compression_amd/csrc
cp inflate_kernels.hip /tmp/inf_keep.hip
for cfg in "NT=512,INF_IN=32768,ROUND_OUT_BUDGET=28672" "NT=256,INF_IN=16384,ROUND_OUT_BUDGET=24576" "NT=512,INF_IN=32768,ROUND_OUT_BUDGET=28672" "NT=256,INF_IN=16384,ROUND_OUT_BUDGET=24576"; do
  cp /tmp/inf_keep.hip inflate_kernels.hip
  for kv in ${cfg//,/ }; do
    k=${kv%%=*}; v=${kv##*=}
    sed -i -e "s/^constexpr int $k = [0-9]*;/constexpr int $k = $v;/" inflate_kernels.hip
  done
  make -s 2>&1 | grep -E "error"
  echo "== $cfg"
  (cd $GRAFT_REPO_ROOT && for slots in 1 2; do python bench.py --no-cpu-baseline --no-slot-comparison --encode-slots $slots 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots', d['config']['encode_slots'], d['value'], d['ms_per_step'], d['stages']['ms'])"; done)
done
cp /tmp/inf_keep.hip inflate_kernels.hip
make -s
