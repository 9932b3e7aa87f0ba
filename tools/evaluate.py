#!/usr/bin/env python3
"""Corpus size comparison: the counterpart of the reference's scripts/evaluate.py:52-136 without pydicom.

    python tools/evaluate.py DIRECTORY [--results FILE.csv] [--batch 256]

Every slice under DIRECTORY (.npy, .u16/.raw, .u16.zz, 16-bit .png) gets one CSV row `File,Raw,ZIP,PNG,RLE,JP2,CCT` as
in results/encoder-comparisons.csv: Raw = bytes of the pixel array, ZIP = zlib.compress at the default level
(evaluate.py:69-71), PNG = 16-bit PNG of value << 4 (lib/png.py:25-31, written with Pillow), CCT = len(Encoder.encode())
(evaluate.py:86-89).  RLE (pydicom's DICOM RLE) and JP2 (an external opj_compress.exe) need software that is not part
of this environment: those columns hold NA.  The reference fans the slices over a process pool
(evaluate.py:107-119); here slices of one shape go to the GPU in batches through cct_hip.encode_batch, and the CPU
columns are computed by a thread pool meanwhile.
"""
import argparse
import io
import json
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.dirname(os.path.abspath(__file__))]
from _inputs import list_inputs, load_slice  # noqa: E402

FILE, RAW, ZIP, PNG, RLE, JP2, CCT = "File", "Raw", "ZIP", "PNG", "RLE", "JP2", "CCT"  # evaluate.py:29-35


def png_size(image):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray((image.astype(np.uint32) << 4).astype(np.uint16)).save(buf, format="PNG")
    return buf.tell()


def cpu_columns(image):
    return {RAW: image.nbytes, ZIP: len(zlib.compress(image.tobytes())), PNG: png_size(image), RLE: "NA", JP2: "NA"}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--results", default=os.path.join(ROOT, "gpurun_out", "evaluation.csv"))
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args(argv)
    import cct_hip
    with open(os.path.join(ROOT, "2023-compact-image-compression_amd", "config.json")) as f:
        config = json.load(f)
    config["verbose"] = False  # evaluate.py:101
    paths = list_inputs(args.directory)
    if not paths:
        print(f"no slices under {args.directory}")
        return 1
    rows = {}
    groups = {}
    for uid, path in enumerate(paths):
        img = load_slice(path)
        name = f"({uid:04})-{os.path.basename(path)}"  # evaluate.py:55
        rows[name] = {FILE: name}
        groups.setdefault((img.shape, img.dtype.str), []).append((name, img))
    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as pool:
        futures = {name: pool.submit(cpu_columns, img) for items in groups.values() for name, img in items}
        for items in groups.values():
            for i in range(0, len(items), args.batch):
                chunk = items[i:i + args.batch]
                files = cct_hip.encode_batch(np.stack([img for _, img in chunk]), config)
                for (name, _), f in zip(chunk, files):
                    rows[name][CCT] = len(f)
        for name, fut in futures.items():
            rows[name].update(fut.result())
    outputs = sorted(rows.values(), key=lambda r: r[FILE])  # evaluate.py:130
    cols = [FILE, RAW, ZIP, PNG, RLE, JP2, CCT]
    os.makedirs(os.path.dirname(os.path.abspath(args.results)), exist_ok=True)
    with open(args.results, "w") as fout:  # evaluate.py:133-136
        fout.write(",".join(cols))
        for line in outputs:
            fout.write("\n" + ",".join(str(line[c]) for c in cols))
    try:
        from tabulate import tabulate
        print(tabulate([[r[c] for c in cols] for r in outputs], headers=cols, tablefmt="simple_outline"))
    except ImportError:
        for r in outputs:
            print(r)
    raw, cct = sum(r[RAW] for r in outputs), sum(r[CCT] for r in outputs)
    print(f"{len(outputs)} slices, raw {raw} B, CCT {cct} B, ratio {raw / cct:.6f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
