#!/bin/bash
# run on the GPU box: SQ counters of the stage (i) kernels (one pass, 8 SQ slots), summarised per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="$@"
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_sq -- python tools/prof_encode.py --paths 1 --reps 5 $ARGS > gpurun_out/pmc_sq.log 2>&1
f=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
    k = m.group(1) if m else row["Kernel_Name"][:40]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k].add(row["Dispatch_Id"])
for k, c in acc.items():
    n = len(cnt[k])
    print(k, "dispatches", n)
    for name, v in sorted(c.items()):
        print(f"   {name:22s} {v / n:16.0f}")
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"   parked {c['SQ_WAIT_ANY'] / wc:.3f}  issue-stall {c['SQ_WAIT_INST_ANY'] / wc:.3f}  issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.3f}")
PY
cp "$f" gpurun_out/pmc_sq_counter_collection.csv
rm -rf gpurun_out/pmc_sq
