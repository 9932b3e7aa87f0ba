#!/bin/bash
# GPU box: default bench (with the CPU baseline), then the rocprofv3 kernel-trace summary of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rm -rf gpurun_out/prof_final
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python bench.py --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/bench_under_rocprof.err
find gpurun_out/prof_final -name "*kernel_stats.csv" | xargs -I{} cp {} gpurun_out/final_kernel_stats.csv
timeout -k 10 300 python bench.py --no-cpu-baseline --no-overlap > gpurun_out/bench_no_overlap.json 2> gpurun_out/bench_no_overlap.err
cat gpurun_out/bench_default.json
