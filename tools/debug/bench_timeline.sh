#!/bin/bash
# run on the GPU box: kernel trace of the bench loop, then where the encode stream's time goes in one step
# (kernels back to back, or gaps between them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_t
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python bench.py --steps 12 --warmup 3 --no-slot-comparison --no-cpu-baseline "$@" > gpurun_out/bench_traced.json 2> gpurun_out/bench_traced.err
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("cct::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
# steps are delimited by stream_kernel launches
starts = [i for i, r in enumerate(rows) if "stream_kernel" in r["Kernel_Name"]]
print("stream launches", len(starts))
enc = ("stream_kernel", "dfl_")
for a, b in list(zip(starts, starts[1:]))[6:9]:
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"--- step of {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.0f} us")
    prev_end = t0
    for r in rows[a:b]:
        if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]: continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {short(r['Kernel_Name']):28s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap_before {(s - prev_end) / 1e3:6.1f}")
        prev_end = max(prev_end, e)
PY
rm -rf gpurun_out/prof_t
