#!/bin/bash
# run on the GPU box: deflate tests, then a kernel-stats profile of the non-overlapped bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/t_deflate.log 2>&1 || { tail -30 gpurun_out/t_deflate.log; exit 1; }
tail -2 gpurun_out/t_deflate.log
rm -rf gpurun_out/prof_run
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_run -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
find gpurun_out/prof_run -name "*kernel_stats.csv" | xargs -I{} cp {} gpurun_out/run_kernel_stats.csv
