#!/bin/bash
# run on the GPU box: per-kernel times of serial encode_batch / decode_batch calls (no overlap) from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c -- python tools/prof_codec.py --reps 5 "$@" > gpurun_out/prof_codec_traced.log 2>&1
f=$(find gpurun_out/prof_c -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/codec_kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/codec_kernel_stats.csv")))
tot = 0
for r in rows[:24]:
    name = r["Name"].replace("cct::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{name:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):6.2f}")
PY
rm -rf gpurun_out/prof_c
