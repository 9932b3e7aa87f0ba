#!/bin/bash
# run on the GPU box: per-kernel times of the bench loop (encode and decode overlapped, as measured) from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python bench.py --steps 20 --warmup 3 --no-slot-comparison --no-cpu-baseline "$@" > gpurun_out/bench_traced.json 2> gpurun_out/bench_traced.err
f=$(find gpurun_out/prof_b -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/bench_kernel_stats.csv
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/bench_kernel_stats.csv")))
for r in rows[:22]:
    name = r["Name"].replace("cct::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{name:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):6.2f}")
try:
    d = json.load(open("gpurun_out/bench_traced.json"))
    print("bench under tracer:", d["value"], d["ms_per_step"], d["stages"]["ms"])
except Exception as e:
    print("no bench json", e)
PY
rm -rf gpurun_out/prof_b
