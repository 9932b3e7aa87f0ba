#!/bin/bash
# run on the GPU box: kernel trace of serial encode calls; start / duration of every kernel of one DEFLATE pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_t
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -- python tools/prof_codec.py --what enc --reps 4 > /dev/null 2>&1
f=$(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("cct::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
starts = [i for i, r in enumerate(rows) if "stream_kernel" in r["Kernel_Name"]]
a = starts[-2]; b = starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {short(r['Kernel_Name']):28s} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
PY
rm -rf gpurun_out/prof_t
