#!/bin/bash
# run on the GPU box: SQ counters of every kernel of serial encode_batch / decode_batch calls, summarised per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_c -- python tools/prof_codec.py --reps 3 "$@" > gpurun_out/pmc_codec.log 2>&1
f=$(find gpurun_out/pmc_c -name "*counter_collection.csv" | head -1)
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
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    w = c.get("SQ_WAVES", 0) or 1
    print(f"{k:28s} disp {n:3d} waves {w/n:9.0f} wave_cycles/wave {wc/w:10.0f} parked {c['SQ_WAIT_ANY']/wc:.3f} issue-stall {c['SQ_WAIT_INST_ANY']/wc:.3f} issuing {c['SQ_ACTIVE_INST_ANY']/wc:.3f} valu/wave {c['SQ_INSTS_VALU']/w:9.0f} salu/wave {c['SQ_INSTS_SALU']/w:9.0f} lds/wave {c['SQ_INSTS_LDS']/w:8.0f}")
PY
rm -rf gpurun_out/pmc_c
