#!/usr/bin/env python3
"""Round-trip demo: the counterpart of the reference's scripts/demo.py:38-106 without pydicom.

    python tools/demo.py INPUT [--workdir DIR] [--shape W,H]

Encodes one slice to `<workdir>/testing.<extension>` through codec.core.Encoder, reads the file back, decodes it
through codec.core.Decoder into the preview file get_filename() names (scripts/demo.py:16-25: "decoded-<name>.<format>"),
and prints what the reference prints: process times, error count, MSE, RMSE and the two SHA-1 digests.  Exit code 1
if the reconstruction is not exact.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2023-compact-image-compression_amd"), os.path.dirname(os.path.abspath(__file__))]
from _inputs import load_slice  # noqa: E402


def get_filename(path, is_encoding, config):
    """scripts/demo.py:16-25: <dir>/encoded-<name>.<extension> or <dir>/decoded-<name>.<decode_format>."""
    directory, filename = os.path.split(path)
    name, _ = filename.split(".", 1)
    transfer_type = "encoded" if is_encoding else "decoded"
    filetype = config["extension"] if is_encoding else config["decoder"]["decode_format"]
    return f"{directory}/{transfer_type}-{name}.{filetype}"


def MSE(A, B):  # scripts/demo.py:27-31 (the reference subtracts the uint16 arrays as they are)
    deviation = A - B
    return np.mean(np.square(deviation))


def RMSE(A, B):  # scripts/demo.py:33-36
    return np.sqrt(MSE(A, B))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "demo"))
    ap.add_argument("--shape", default=None, help="W,H for raw inputs that are not square")
    args = ap.parse_args(argv)
    from codec.core import Decoder, Encoder
    with open(os.path.join(ROOT, "2023-compact-image-compression_amd", "config.json")) as f:
        config = json.load(f)
    shape = tuple(int(x) for x in args.shape.split(",")) if args.shape else None
    image = load_slice(args.input, shape)
    os.makedirs(args.workdir, exist_ok=True)

    print("\n==================== [ENCODING] ====================")
    encoded_path = os.path.join(args.workdir, f"testing.{config['extension']}")
    start = time.process_time()
    Encoder(config, image, encoded_path).encode()
    print(f"\nEncoding Elapsed Time: {time.process_time() - start:.2f} sec")
    print(f'\n"{os.path.basename(args.input)}" encoded to "{encoded_path}"')

    print("\n==================== [DECODING] ====================\n")
    with open(encoded_path, "rb") as f:
        file_bytes = f.read()
    decoded_path = get_filename(encoded_path, False, config)
    start = time.process_time()
    output = Decoder(config, file_bytes, decoded_path).decode()
    print(f"Decoding Elapsed Time: {time.process_time() - start:.2f} sec")
    print(f'\n"{encoded_path}" preview decoded to "{decoded_path}"\n')

    error = int(np.count_nonzero(image - output))  # scripts/demo.py:85-86
    print(f"Total Error: {error}")
    print(f"Mean-Squared-Error: {MSE(image, output)}")
    print(f"Root-Mean-Squared-Error: {RMSE(image, output)}\n")
    original_hash = hashlib.sha1(image.tobytes()).hexdigest()
    recovered_hash = hashlib.sha1(output.tobytes()).hexdigest()
    print(f"SHA1 Original Hash:  {original_hash}")
    print(f"SHA1 Recovered Hash: {recovered_hash}")
    print(f"\n{len(file_bytes)} bytes, ratio {image.nbytes / len(file_bytes):.4f}")
    return 0 if (error == 0 and original_hash == recovered_hash) else 1


if __name__ == "__main__":
    sys.exit(main())
