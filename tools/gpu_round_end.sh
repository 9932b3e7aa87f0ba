#!/bin/bash
# GPU box: everything profiles/ holds for a round, in one call (about six minutes).
#   default bench (with the CPU baseline) -> the same command under rocprofv3 --kernel-trace --stats ->
#   two encode slots, no overlap, configs 4 and 5 -> HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and SQ
#   counters of the transform+pack kernels.  Copy what is kept from gpurun_out/round_end/ into profiles/ (tools/collect_profiles.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/round_end
rm -rf $O && mkdir -p $O
set -e
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "default bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --no-slot-comparison > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
echo "traced bench done"
timeout -k 10 300 python bench.py --no-cpu-baseline --encode-slots 2 > $O/bench_two_slots.json 2> $O/bench_two_slots.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python bench.py --no-cpu-baseline --no-slot-comparison --encode-slots 2 --steps 20 > $O/bench_two_slots_under_rocprof.json 2> $O/bench_two_slots_under_rocprof.err
find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_two_slots_kernel_stats.csv
rm -rf $O/prof1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-overlap > $O/bench_no_overlap.json 2> $O/bench_no_overlap.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-slot-comparison --no-pipeline-scheduling > $O/bench_no_pipeline_scheduling.json 2> $O/bench_no_pipeline_scheduling.err
echo "two slots / no overlap done"
timeout -k 10 600 python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
echo "configs 4, 5 done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python tools/prof_encode.py --paths 1 --reps 8 > $O/pmc_$c.log 2>&1
  find $O/pmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_${c}_counter_collection.csv
  rm -rf $O/pmc_$c
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_sq -- python tools/prof_encode.py --paths 1 --reps 5 > $O/pmc_sq.log 2>&1
find $O/pmc_sq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_sq_counter_collection.csv
rm -rf $O/pmc_sq
timeout -k 10 200 python tools/prof_encode.py --paths 1,2 --reps 30 > $O/prof_encode.log 2>&1
# every stage alone (serial calls, nothing else on the device), and the INFLATE kernel's phase profile for both geometries
timeout -k 10 200 python tools/prof_codec.py --reps 5 > $O/prof_codec.log 2>&1
CCT_INF_PROF=1 timeout -k 10 200 python tools/prof_codec.py --reps 1 --what dec 2>&1 | grep "inflate prof" | head -1 >> $O/prof_codec.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python tools/prof_codec.py --reps 5 > $O/prof_codec_traced.log 2>&1
find $O/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/codec_serial_kernel_stats.csv
rm -rf $O/prof2
# the other device stages (DEFLATE chain, INFLATE, decode kernel): HBM traffic and SQ counters per kernel, serial encode / decode calls
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcc_$c -- python tools/prof_codec.py --reps 3 > $O/pmc_codec_$c.log 2>&1
  find $O/pmcc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_codec_${c}_counter_collection.csv
  rm -rf $O/pmcc_$c
done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmcc_sq -- python tools/prof_codec.py --reps 3 > $O/pmc_codec_sq.log 2>&1
find $O/pmcc_sq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_codec_sq_counter_collection.csv
rm -rf $O/pmcc_sq
# phase stamps of the streaming kernel (diagnostic build of the same source)
CCT_STREAM_STAMPS=1 timeout -k 10 200 python tools/prof_encode.py --paths 1 --reps 1 2>&1 | grep -A16 "stream stamps" | head -17 > $O/stream_stamps.log
echo "pmc done"
python -c "
import json
for f in ('bench','bench_two_slots','bench_no_overlap','bench_no_pipeline_scheduling','bench_config4','bench_config5'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])
"
