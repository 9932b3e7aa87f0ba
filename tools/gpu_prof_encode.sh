#!/bin/bash
# run on the GPU box: stage (i) alone, event timing, then per-kernel times from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="$@"
timeout -k 10 200 python tools/prof_encode.py $ARGS > gpurun_out/prof_encode.log 2>&1 || { tail -20 gpurun_out/prof_encode.log; exit 1; }
cat gpurun_out/prof_encode.log
rm -rf gpurun_out/prof_enc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc -- python tools/prof_encode.py --paths 1 $ARGS > gpurun_out/prof_encode_traced.log 2>&1
find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/enc_kernel_stats.csv
cut -d, -f1-4,6,7 gpurun_out/enc_kernel_stats.csv | head -12
rm -rf gpurun_out/prof_enc
