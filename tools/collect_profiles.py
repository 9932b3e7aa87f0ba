#!/usr/bin/env python3
"""Copy the round's measurements from gpurun_out/round_end/ (tools/gpu_round_end.sh) into profiles/ under the round's prefix and
derive the PMC summary bench.py reads (HBM traffic of the transform+pack stage, stamped with the hash of its source).

    python tools/collect_profiles.py r02
"""
import csv
import collections
import hashlib
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round_end")
DST = os.path.join(ROOT, "profiles")
PIPE_SRC = os.path.join(ROOT, "2023-compact-image-compression_amd", "csrc", "encode_stream.hip")
KERNELS = ("stream_kernel",)


def per_kernel(path, counter):
    """counter value per dispatch, summed over the dimensions rocprofv3 reports, averaged over dispatches"""
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"(stream_kernel)", row["Kernel_Name"])
            if not m:
                continue
            tot[m.group(1)] += float(row["Counter_Value"])
            disp[m.group(1)].add(row["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}, {k: len(disp[k]) for k in tot}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    keep = ["bench.json", "bench_under_rocprof.json", "bench_kernel_stats.csv", "bench_two_slots.json",
            "bench_two_slots_under_rocprof.json", "bench_two_slots_kernel_stats.csv", "bench_no_overlap.json", "bench_no_pipeline_scheduling.json",
            "bench_config4.json", "bench_config5.json", "pmc_FETCH_SIZE_counter_collection.csv",
            "pmc_WRITE_SIZE_counter_collection.csv", "pmc_sq_counter_collection.csv", "prof_encode.log", "prof_codec.log", "codec_serial_kernel_stats.csv",
            "pmc_codec_FETCH_SIZE_counter_collection.csv", "pmc_codec_WRITE_SIZE_counter_collection.csv", "pmc_codec_sq_counter_collection.csv",
            "pmc_codec.json", "stream_stamps.log"]
    for name in keep:
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, f"{tag}_{name}"))
    fetch, nf = per_kernel(os.path.join(SRC, "pmc_FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE")
    write, nw = per_kernel(os.path.join(SRC, "pmc_WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE")
    with open(PIPE_SRC, "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    per = {}
    total = 0
    for k in KERNELS:
        # counters are KB; on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane coalesced reads
        # (MI355X_MICROARCH.md, HBM section): x2 on the read side, WRITE_SIZE as it is
        fb, wb = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
        per[k] = {"FETCH_SIZE_KB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                  "fetch_bytes_corrected": int(fb), "write_bytes": int(wb), "dispatches": [nf.get(k, 0), nw.get(k, 0)]}
        total += fb + wb
    out = {"kernels": per, "traffic_bytes_per_launch": int(total), "source_sha1": sha,
           "workload": "256 x 512x512 uint16 (bench batch), tools/prof_encode.py --paths 1",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; counters are KB, summed over "
                     "the reported dimensions, averaged over the dispatches; x2 on the read side (gfx950, 16-byte-per-lane reads); "
                     "one 'launch' = the one kernel of the stage (stream_kernel)",
           "algorithmic_read_bytes": 256 * 512 * 512 * 2}
    with open(os.path.join(DST, f"{tag}_pmc_encode.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    codec_summary(tag)


def codec_summary(tag):
    """Per kernel of the serial encode / decode calls (tools/prof_codec.py): FETCH_SIZE / WRITE_SIZE (KB, as counted: the x2
    of the gfx950 note applies to 16-byte-per-lane streaming reads only and is NOT applied here) and the SQ counters,
    averaged over the dispatches."""
    def table(name, counters):
        path = os.path.join(SRC, name)
        if not os.path.exists(path):
            return {}
        acc, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] not in counters:
                    continue
                m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
                k = m.group(1) if m else row["Kernel_Name"][:48]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add(row["Dispatch_Id"])
        return {k: dict({c: v / len(disp[k]) for c, v in acc[k].items()}, dispatches=len(disp[k])) for k in acc}
    fetch = table("pmc_codec_FETCH_SIZE_counter_collection.csv", ("FETCH_SIZE",))
    write = table("pmc_codec_WRITE_SIZE_counter_collection.csv", ("WRITE_SIZE",))
    sq = table("pmc_codec_sq_counter_collection.csv", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                        "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"))
    if not (fetch or write or sq):
        return
    out = {}
    for k in sorted(set(fetch) | set(write) | set(sq)):
        row = {}
        if k in fetch:
            row["FETCH_SIZE_KB"] = round(fetch[k]["FETCH_SIZE"], 1)
        if k in write:
            row["WRITE_SIZE_KB"] = round(write[k]["WRITE_SIZE"], 1)
        if k in sq:
            c = sq[k]
            wc = c.get("SQ_WAVE_CYCLES", 0) or 1
            row.update({"waves": round(c.get("SQ_WAVES", 0)), "valu": round(c.get("SQ_INSTS_VALU", 0)), "salu": round(c.get("SQ_INSTS_SALU", 0)),
                        "lds": round(c.get("SQ_INSTS_LDS", 0)), "parked": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                        "issue_stall": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3), "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                        "dispatches": c["dispatches"]})
        out[k] = row
    doc = {"workload": "256 x 512x512 uint16 (bench batch 0), serial cct_encode_batch / cct_decode_batch calls, tools/prof_codec.py --reps 3",
           "method": "rocprofv3 --kernel-trace --pmc <counter(s)> in separate passes for FETCH_SIZE, WRITE_SIZE and the SQ set; values per dispatch, "
                     "summed over the reported dimensions, averaged over the dispatches of a kernel; FETCH_SIZE / WRITE_SIZE in KB as counted",
           "kernels": out}
    with open(os.path.join(DST, f"{tag}_pmc_codec.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{tag}_pmc_codec.json: {len(out)} kernels")


if __name__ == "__main__":
    main()
