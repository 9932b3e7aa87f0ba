#!/usr/bin/env python3
"""bench.py -- CompaCT encode+decode throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|4|5]
    (N > 1: launched by the driver as python -m torch.distributed.run ... bench.py --gpus N ...)

Workloads (BASELINE.json `configs`):
  --config 2 (default, the headline)  256 device-resident synthetic 12-bit 512x512 CT slices per GPU and step (phantoms that
             use the 12-bit container as the real corpus does, values up to ~2200; --phantom 11bit gives rounds 1-2's
             phantoms, all below 2048): encoded to
             byte-exact .cct files (transform+pack kernels, device DEFLATE bit-identical to zlib level 9, one packed D2H
             into a page-locked archive) and decoded back to rasters in HBM (archive H2D, device INFLATE, token/scatter
             kernel).  Three distinct batches rotate so the working set (3 x 134 MB) exceeds the 256 MiB Infinity Cache.
  --config 4  the same pipeline on 512 slices of 1024x1024 per step (1.07 GB of pixels: HBM for certain).
  --config 5  decode only: the archives of config 2 are encoded once outside the timed region; a step decodes one.
Slices shard across GPUs with no data-path collective (weak scaling: the same number of slices per GPU per step); the
only exchange is the all-gather of the per-slice compressed sizes over RCCL, reached through the library's C ABI
(cct_comm_init / cct_allgather_u32; the communicator id travels from rank 0 through a file): no PyTorch in this script.

One JSON line on stdout (rank 0): metric/value as the driver contract says, plus
  roofline      the transform+pack stage (stream_kernel of encode_stream.hip; decode-only: INFLATE + decode kernel):
                algorithmic HBM bytes per launch / mean stage time from HIP events recorded on the library's own stream
                around every launch of the timed region, against 8 TB/s; `traffic` from profiles/ only while the kernel
                source still has the hash the counters were collected with;
  rooflines     the same arithmetic for the other device stages (DEFLATE, INFLATE, decode kernel);
  cpu_baseline  the CPU oracle (C port of the reference algorithm): one core, and all host cores through a process
                pool (the reference's own fan-out, scripts/evaluate.py:107), CPU model and core count stated;
  stages        per-stage times, single-slice latency through codec.core (BASELINE configs[0] shape of call).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "2023-compact-image-compression_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

BS = 16
N_ROT = 3
HBM_PEAK_GBS = 8000.0
PIPE_SRC = os.path.join(PKG, "csrc", "encode_stream.hip")
PMC_JSON = os.path.join(ROOT, "profiles", "r03_pmc_encode.json")
WORKLOADS = {2: (512, 256), 4: (1024, 512), 5: (512, 256)}  # config -> (edge, slices per GPU and step)


def _usable_cpus():
    """CPUs this process may really use: affinity mask, capped by a cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _phantom_job(args):
    """ct_phantom(seed, edge, depth12), through a cache on local disk: the driver runs N = 1, 2, 4, 8 back to back and
    rank r of a wider run needs the seeds a narrower one has already made."""
    from cct_hip.synth import ct_phantom
    seed, edge, depth12 = args
    cache = os.environ.get("CCT_PHANTOM_CACHE", os.path.join("/tmp", f"cct_phantoms_{os.getuid()}"))
    path = os.path.join(cache, f"{edge}_{int(depth12)}_{seed}.npy") if cache != "0" else None
    if path:
        try:
            a = np.load(path)
            if a.shape == (edge, edge) and a.dtype == np.uint16:
                return a
        except (OSError, ValueError):
            pass
    a = ct_phantom(seed, edge, depth12)
    if path:
        try:
            os.makedirs(cache, exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp.npy"
            np.save(tmp, a)
            os.replace(tmp, path)
        except OSError:
            pass
    return a


def make_batches(rank, n_slices, edge=512, depth12=False):
    """Batch 0 = ct_phantom(seed) for distinct seeds (rank-disjoint; 1024x1024: 32 distinct phantoms, tiled, which keeps
    the generation time of 512 slices bounded); batches 1, 2 are its left-right / up-down mirrors: distinct bytes in
    HBM, same statistics.  Uses a process pool: call it BEFORE anything initialises the GPU (fork)."""
    from concurrent.futures import ProcessPoolExecutor
    distinct = n_slices if edge <= 512 else min(n_slices, 32)
    seeds = [(rank * n_slices + i, edge, depth12) for i in range(distinct)]
    # at least four workers per rank also when eight ranks share the host (two each took 4-5 s before the first GPU call)
    workers = max(min(4, _usable_cpus()), min(16, _usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    if workers > 1:
        with ProcessPoolExecutor(workers) as ex:
            imgs = list(ex.map(_phantom_job, seeds, chunksize=4))
    else:
        imgs = [_phantom_job(s) for s in seeds]
    b0 = np.stack([imgs[i % distinct] for i in range(n_slices)])
    return [b0, np.ascontiguousarray(b0[:, :, ::-1]), np.ascontiguousarray(b0[:, ::-1, :])][:N_ROT]


def _oracle_job(img):
    from oracle import oracle
    t0 = time.perf_counter()
    f = oracle.encode(img)
    t1 = time.perf_counter()
    oracle.decode(f)
    return t1 - t0, time.perf_counter() - t1


def cpu_baseline(batch, budget_s=10.0):
    """The CPU restatement (oracle/compact_oracle.c) on a bounded sample of the workload: one core, then every usable
    core through a process pool as scripts/evaluate.py:107 fans slices out.  Call before the GPU is initialised."""
    from concurrent.futures import ProcessPoolExecutor
    px = batch.shape[1] * batch.shape[2]
    t0 = time.perf_counter()
    n1, te, td = 0, 0.0, 0.0
    for img in batch:
        e, d = _oracle_job(img)
        te += e; td += d; n1 += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt1 = time.perf_counter() - t0
    cores = _usable_cpus()
    n_all = min(len(batch), max(cores, int(n1 * cores * 0.8)))
    t0 = time.perf_counter()
    with ProcessPoolExecutor(cores) as ex:
        list(ex.map(_oracle_job, batch[:n_all], chunksize=max(1, n_all // (cores * 4))))
    dt_all = time.perf_counter() - t0
    return {"value": round(n1 * px / dt1 / 1e6, 3), "unit": "MPixels/s", "cores": 1, "kind": "port",
            "sample": f"first {n1} slices of batch 0, encode+decode each, oracle/compact_oracle.c single thread",
            "encode_MPix_s": round(n1 * px / te / 1e6, 3), "decode_MPix_s": round(n1 * px / td / 1e6, 3),
            "all_cores": {"value": round(n_all * px / dt_all / 1e6, 3), "unit": "MPixels/s", "cores": cores,
                          "sample": f"first {n_all} slices of batch 0 over a pool of {cores} processes (incl. pool start)"},
            "host_cpu_count": os.cpu_count(), "cpu_model": _cpu_model(),
            "reference_python_survey_container": {"encode_MPix_s": 0.105, "decode_MPix_s": 0.157,
                                                  "note": "reference src/codec timed in the survey container (SURVEY 6), 1 core"}}


def pmc_traffic():
    """HBM traffic of the transform+pack stage from the committed --pmc run, only while the kernel source is unchanged."""
    try:
        with open(PMC_JSON) as f:
            pmc = json.load(f)
        with open(PIPE_SRC, "rb") as f:
            sha = hashlib.sha1(f.read()).hexdigest()
        if pmc.get("source_sha1") == sha:
            return pmc["traffic_bytes_per_launch"], os.path.relpath(PMC_JSON, ROOT) + " (FETCH_SIZE x2 + WRITE_SIZE, same launch shape)"
        return None, "stale: " + os.path.relpath(PMC_JSON, ROOT) + " was collected with another encode_stream.hip"
    except (OSError, KeyError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--slices", type=int, default=None, help="slices per GPU per step (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phantom", default="11bit", choices=("12bit", "11bit"),
                    help="synthetic slices in the 12-bit container: 11bit = the phantoms of rounds 1-2, every value below 2048 (the "
                         "default: the headline stays comparable across rounds); 12bit = ct_phantom(depth12=True), bone at 1900-2150 and "
                         "values above 2047 like the real corpus.  The other kind is timed after the timed region (stages.other_phantom)")
    ap.add_argument("--no-overlap", action="store_true", help="finish decode of step k before encoding step k+1")
    ap.add_argument("--no-slot-comparison", action="store_true",
                    help="skip the short run with the other --encode-slots setting after the timed region (use under a tracer)")
    ap.add_argument("--no-pipeline-scheduling", action="store_true",
                    help="library options decode_yields = 0 and queue_ahead = 0: decode kernels start at once, an encode call keeps its "
                         "slot until the host has the sizes (rounds 1-2; for the A/B in profiles/)")
    ap.add_argument("--encode-slots", type=int, default=1, choices=(1, 2),
                    help="encode batches on the device at a time (library option encode_slots; 2 is worth -8 ... +8 %% end to end depending on "
                         "the box, but then every kernel's duration includes another batch's "
                         "DEFLATE pass next to it)")
    args = ap.parse_args()
    edge, n = WORKLOADS[args.config]
    if args.slices:
        n = args.slices
    if args.steps is None:
        args.steps = 50 if edge == 512 else 8
    W = H = edge
    npx = n * W * H
    decode_only = args.config == 5

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- host-side work that forks: before any GPU / RCCL initialisation
    batches = make_batches(rank, n, edge, args.phantom == "12bit")
    other_batches = None  # the other kind of phantom, for a short run after the timed region (single GPU, headline config only)
    if args.config == 2 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_slot_comparison:
        other_batches = make_batches(rank, n, edge, args.phantom != "12bit")
    phantom_note = ("12-bit container; phantoms with bone at 1900-2150 + texture, values above 2047 as in the real corpus, Q7-safe by construction"
                    if args.phantom == "12bit" else "12-bit container; the phantoms of rounds 1-2, every value below 2048")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batches[0])

    import cct_hip
    from cct_hip import _ffi, parallel
    L = _ffi.lib()
    _ffi.check(L.cct_init(local_rank))
    multi = world > 1
    if multi:  # one process per GPU; RCCL through the library's C ABI (no PyTorch in the data path or around it)
        parallel.comm_init(rank, world)

    def gather_sizes(local_sizes):
        return parallel.gather_sizes_rccl(local_sizes) if multi else np.asarray(local_sizes, dtype=np.uint32).copy()

    info = cct_hip.device_info()
    # host team: the library sizes it from the CPUs this process may use (cgroup quota aware); ranks of one node share them
    zt = C.c_int(0)
    _ffi.check(L.cct_get_option(b"zlib_threads", C.byref(zt)))
    zthreads = max(1, zt.value // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    if os.environ.get("CCT_HOST_THREADS"):
        zthreads = int(os.environ["CCT_HOST_THREADS"])
    _ffi.check(L.cct_set_option(b"zlib_threads", zthreads))
    _ffi.check(L.cct_set_option(b"encode_slots", args.encode_slots))
    if args.no_pipeline_scheduling:
        _ffi.check(L.cct_set_option(b"decode_yields", 0))
        _ffi.check(L.cct_set_option(b"queue_ahead", 0))
    for env, key in (("CCT_WG_THREADS", b"wg_threads"), ("CCT_DEFLATE_GRAPH", b"deflate_graph"),
                     ("CCT_DEVICE_INFLATE", b"device_inflate"), ("CCT_TILE_PATH", b"tile_path"),
                     ("CCT_ENCODE_SLOTS", b"encode_slots"), ("CCT_INFLATE_LANES", b"inflate_lanes"),
                     ("CCT_DEFLATE_FORK", b"deflate_fork")):
        if os.environ.get(env):
            _ffi.check(L.cct_set_option(key, int(os.environ[env])))
    dev_deflate, dev_inflate = C.c_int(0), C.c_int(0)
    _ffi.check(L.cct_get_option(b"device_deflate", C.byref(dev_deflate)))
    _ffi.check(L.cct_get_option(b"device_inflate", C.byref(dev_inflate)))
    cfg = cct_hip.default_config()
    cfg["verbose"] = False
    flags, bs, eof, magic, ch, bpc = cct_hip.codec_params(cfg, np.uint16)

    d_imgs = [cct_hip.DeviceBuffer.from_numpy(b) for b in batches]
    # Three buffer sets: while step k is decoded (device INFLATE + decode kernel on the decode stream) step k+1 is
    # already being encoded (transform+pack + device DEFLATE).  Host threads drive the C ABI; ctypes drops the GIL.
    out_stride = L.cct_file_bound(W, H, bs)
    NSET = 3
    d_back = [cct_hip.DeviceBuffer(batches[0].nbytes) for _ in range(NSET)]
    arch_cap = n * out_stride
    h_arch_pin = [cct_hip.PinnedArray(arch_cap) for _ in range(NSET)]   # page-locked: D2H / H2D without staging
    h_arch = [p.array for p in h_arch_pin]                                # .cct files back to back (archive layout)
    h_offs = [np.zeros(n + 1, dtype=np.uint64) for _ in range(NSET)]
    h_sizes = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    h_psizes = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    h_status = [np.zeros(n, dtype=np.uint32) for _ in range(NSET)]
    acc = {"enc_kernel": 0.0, "d2h": 0.0, "deflate": 0.0, "inflate": 0.0, "dec_kernel": 0.0, "enc": 0.0, "dec": 0.0,
           "gather": 0.0, "payload_bytes": 0, "file_bytes": 0, "n_enc": 0, "n_dec": 0}
    from concurrent.futures import ThreadPoolExecutor
    # two encode threads: the library gives the device lock back while a batch's files are still on the wire, so
    # the next batch's kernels start during that copy
    # decode only: two decode threads, so that the archive upload of one batch runs under the kernels of the other (two decode slots)
    pool_enc, pool_dec = ThreadPoolExecutor(2 if not args.no_overlap else 1), ThreadPoolExecutor(2 if (args.config == 5 and not args.no_overlap) else 1)
    overlap = not args.no_overlap

    def enc_step(i, k, record):
        t0 = time.perf_counter()
        tm = (C.c_float * 6)()
        _ffi.check(L.cct_encode_batch_packed(d_imgs[i % len(d_imgs)].ptr, 1, n, W, H, bs, flags, eof, magic, ch, bpc,
                                             h_arch[k].ctypes.data, arch_cap, h_offs[k].ctypes.data, h_sizes[k].ctypes.data,
                                             h_status[k].ctypes.data, h_psizes[k].ctypes.data, None))
        L.cct_last_timings(tm)
        if record:
            acc["enc_kernel"] += tm[0]; acc["d2h"] += tm[1]; acc["deflate"] += tm[2]
            acc["enc"] += (time.perf_counter() - t0) * 1e3
            acc["payload_bytes"] += int(h_psizes[k].sum()); acc["file_bytes"] += int(h_sizes[k].sum())
            acc["n_enc"] += 1

    def dec_step(k, enc_future, record):
        if enc_future is not None:
            enc_future.result()
        t0 = time.perf_counter()
        tm = (C.c_float * 6)()
        st = np.zeros(n, dtype=np.uint32)
        _ffi.check(L.cct_decode_batch(h_arch[k].ctypes.data, h_offs[k].ctypes.data, n, bs, magic, d_back[k].ptr, 1,
                                      n * W * H, st.ctypes.data))
        L.cct_last_timings(tm)
        if record:
            acc["inflate"] += tm[3]; acc["dec_kernel"] += tm[4]
            acc["dec"] += (time.perf_counter() - t0) * 1e3
            acc["n_dec"] += 1

    in_flight = []
    state = {"sizes": None}

    def run_steps(first, count, record):
        prev = None  # (future, set) of the previous encode
        for i in range(first, first + count):
            k = i % NSET
            while len(in_flight) >= (NSET if overlap else 1):
                in_flight.pop(0).result()       # buffer set k is free again
            if decode_only:
                in_flight.append(pool_dec.submit(dec_step, k, None, record))
                continue
            e = pool_enc.submit(enc_step, i, k, record)
            in_flight.append(pool_dec.submit(dec_step, k, e, record))
            if not overlap:
                prev = (e, k)
            if prev is not None:                # at most two encodes in flight
                prev[0].result()
                t1 = time.perf_counter()
                state["sizes"] = gather_sizes(h_sizes[prev[1]])  # RCCL all-gather of compressed sizes
                if record:
                    acc["gather"] += (time.perf_counter() - t1) * 1e3
            prev = None if not overlap else (e, k)
        if prev is not None:
            prev[0].result()
            state["sizes"] = gather_sizes(h_sizes[prev[1]])
        while in_flight:
            in_flight.pop(0).result()

    def barrier():
        _ffi.check(L.cct_sync())
        if multi:
            parallel.barrier()

    if decode_only:  # the archives are produced once, outside the timed region
        for k in range(NSET):
            enc_step(k, k, True)
        state["sizes"] = gather_sizes(h_sizes[0])
        acc["n_enc"] = max(1, acc["n_enc"])
    run_steps(0, args.warmup, False)
    barrier()
    t_start = time.perf_counter()
    run_steps(args.warmup, args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t_start
    if multi:
        elapsed = parallel.allreduce_max_float(elapsed)

    # ---- verification outside the timed region: exact round trip + oracle bytes on a sample
    last_i = args.warmup + args.steps - 1
    kset = last_i % NSET
    last = kset % len(batches) if decode_only else last_i % len(batches)
    back = d_back[kset].download(np.uint16, npx).reshape(n, W, H)
    verified = bool(np.array_equal(back, batches[last]))
    from oracle import oracle
    for j in (0, n // 2, n - 1):
        verified &= oracle.encode(batches[last][j]) == h_arch[kset][int(h_offs[kset][j]):int(h_offs[kset][j + 1])].tobytes()
    all_sizes = state["sizes"]

    # ---- the same loop with the other encode-slot setting, outside the timed region (20 steps): what the setting is worth
    other = None
    if not decode_only and overlap and not multi and not args.no_slot_comparison:
        other_slots = 2 if args.encode_slots == 1 else 1
        _ffi.check(L.cct_set_option(b"encode_slots", other_slots))
        ko = min(20, args.steps)
        run_steps(0, args.warmup, False)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.warmup, ko, False)
        barrier()
        other = {"encode_slots": other_slots, "MPixels_s": round(npx * ko / (time.perf_counter() - t0) / 1e6, 1), "steps": ko}
        _ffi.check(L.cct_set_option(b"encode_slots", args.encode_slots))

    # ---- the same loop on the other kind of phantom (20 steps, outside the timed region): what the workload choice is worth
    other_ph = None
    if other_batches is not None and not decode_only and overlap:
        for d, b in zip(d_imgs, other_batches):
            d.upload(b)
        ko = min(20, args.steps)
        run_steps(0, args.warmup, False)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.warmup, ko, False)
        barrier()
        other_ph = {"phantom": "12bit" if args.phantom == "11bit" else "11bit", "MPixels_s": round(npx * ko / (time.perf_counter() - t0) / 1e6, 1),
                    "steps": ko, "max_pixel_value": int(max(int(b.max()) for b in other_batches))}
        for d, b in zip(d_imgs, batches):
            d.upload(b)

    # ---- the transform+pack stage with nothing else on the device (outside the timed region): in the timed region two encode
    # batches and a decode share the chip, so the stage's events there measure its kernels next to another batch's DEFLATE
    alone_ms = None
    if not decode_only:
        from cct_hip.batch import payload_stride
        stride = payload_stride(W, H, bs)
        d_pay, d_sz, d_st = cct_hip.DeviceBuffer(n * stride), cct_hip.DeviceBuffer(4 * n), cct_hip.DeviceBuffer(4 * n)
        params = cct_hip.codec_params(cfg, np.uint16)
        e0, e1 = cct_hip.Event(), cct_hip.Event()
        ts = []
        for it in range(13):
            e0.record()
            cct_hip.encode_payload_dev(d_imgs[it % len(d_imgs)], n, W, H, params, d_pay, d_sz, d_st)
            e1.record()
            ts.append(e1.elapsed_ms_since(e0))
        alone_ms = sorted(ts[3:])[len(ts[3:]) // 2]

    # ---- single-slice latency through the reference's class surface (BASELINE configs[0] shape of call)
    single = None
    if rank == 0:
        from codec.core import Decoder, Encoder
        img = batches[0][0]
        te, td = [], []
        for _ in range(7):
            t0 = time.perf_counter(); f = Encoder(cfg, img).encode(); t1 = time.perf_counter()
            Decoder(cfg, f).decode(); t2 = time.perf_counter()
            te.append((t1 - t0) * 1e3); td.append((t2 - t1) * 1e3)
        single = {"encode_ms": round(sorted(te)[3], 3), "decode_ms": round(sorted(td)[3], 3),
                  "note": f"codec.core.Encoder.encode() / Decoder.decode() of one {W}x{H} slice, host arrays in and out, median of 7"}

    if rank == 0:
        K = max(1, args.steps)
        ne, nd = max(1, acc["n_enc"]), max(1, acc["n_dec"])
        ms_step = elapsed * 1e3 / K
        value = world * npx * K / elapsed / 1e6
        enc_kernel_ms = acc["enc_kernel"] / ne
        payload_per_launch = acc["payload_bytes"] / ne
        file_per_launch = acc["file_bytes"] / ne
        traffic, traffic_src = pmc_traffic() if (edge == 512 and n == 256) else (None, None)

        def roof(kernel, alg_bytes, ms, extra=None):
            gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            r = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(alg_bytes), "avg_ms": round(ms, 4)}
            if extra:
                r.update(extra)
            return r

        pack = roof("transform+pack stage: stream_kernel (image -> token payload, one pass), events around its launch",
                    2.0 * npx, enc_kernel_ms,
                    {"traffic": traffic, "traffic_source": traffic_src,
                     "read_plus_write_GBs": round((2.0 * npx + payload_per_launch) / (enc_kernel_ms * 1e-3) / 1e9, 1),
                     "avg_kernel_ms": round(enc_kernel_ms, 4),
                     "alone": {"avg_ms": round(alone_ms, 4), "frac": round(2.0 * npx / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "the same launches with nothing else on the device (median of 10, after the timed region); "
                                       "avg_ms / frac above are measured live, while a decode (and, with --encode-slots 2, another "
                                       "batch's DEFLATE pass) shares the chip"} if alone_ms else None})
        deflate_ms, inflate_ms, dec_ms = acc["deflate"] / ne, acc["inflate"] / nd, acc["dec_kernel"] / nd
        others = {
            "deflate": roof("device DEFLATE pass (sort, match, lazy parse, trees, emit: ~25 kernels as one graph)",
                            payload_per_launch + file_per_launch, deflate_ms),
            "inflate": roof("inflate_kernel (device INFLATE)", file_per_launch + payload_per_launch, inflate_ms),
            "decode": roof("decode_kernel (tokens -> raster)", payload_per_launch + 2.0 * npx, dec_ms),
        }
        # per-kernel HBM bytes of these stages from the committed PMC passes (serial calls on this workload, profiles/): a record,
        # not a live measurement -- FETCH_SIZE / WRITE_SIZE in KB as counted, per dispatch
        if args.config == 2 and n == 256:
            try:
                with open(os.path.join(ROOT, "profiles", "r03_pmc_codec.json")) as f:
                    pk = json.load(f)["kernels"]
                def kb(names):
                    return {k: {"FETCH_SIZE_KB": v.get("FETCH_SIZE_KB"), "WRITE_SIZE_KB": v.get("WRITE_SIZE_KB")}
                            for k, v in pk.items() if k.startswith(names)}
                others["deflate"]["pmc_per_kernel"] = dict(kb(("dfl_",)), source="profiles/r03_pmc_codec.json")
                others["inflate"]["pmc_per_kernel"] = dict(kb(("inflate_",)), source="profiles/r03_pmc_codec.json")
                others["decode"]["pmc_per_kernel"] = dict(kb(("decode_",)), source="profiles/r03_pmc_codec.json")
            except (OSError, KeyError, ValueError):
                pass
        if decode_only:
            main_roof = roof("decode stages: inflate_kernel + decode_kernel (archive -> rasters)", file_per_launch + 2.0 * npx,
                             inflate_ms + dec_ms, {"traffic": None, "traffic_source": None})
        else:
            main_roof = pack
        what = "decode only" if decode_only else "encode to .cct + decode back"
        out = {
            "metric": f"MPixels/s encode+decode, 12-bit {W}x{H} CT batch, bytes-exact" if not decode_only
            else f"MPixels/s decode, 12-bit {W}x{H} CT batch, bytes-exact",
            "value": round(value, 2), "unit": "MPixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: batch of {n} synthetic {W}x{H} uint16 CT slices per GPU "
                                   f"({phantom_note}), {what}, {N_ROT} rotating device-resident batches",
                       "phantom": args.phantom, "max_pixel_value": int(max(int(b.max()) for b in batches)),
                       "slices_per_gpu": n, "width": W, "height": H, "block_size": bs,
                       "flags": "fractal+segmentation+deflate(level 9)", "sharding": f"per-slice, {world} GPU(s)",
                       "encode_slots": args.encode_slots, "pipeline_scheduling": not args.no_pipeline_scheduling},
            "roofline": main_roof,
            "hbm_read_roofline_end_to_end": {  # north_star: the whole job against the HBM-read roofline (2 B per pixel per step)
                "achieved": round(2.0 * npx * world / (ms_step * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "frac": round(2.0 * npx / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "rooflines": others if decode_only else dict(others, transform_pack=pack),
            "stages": {
                "encode_transform_pack_MPix_s": round(npx / (enc_kernel_ms * 1e-3) / 1e6, 1),
                "decode_tokens_scatter_MPix_s": round(npx / (dec_ms * 1e-3) / 1e6, 1),
                "encode_end_to_end_MPix_s": round(npx / (acc["enc"] / ne * 1e-3) / 1e6, 2),
                "decode_end_to_end_MPix_s": round(npx / (acc["dec"] / nd * 1e-3) / 1e6, 2),
                "ms": {"enc_kernel": round(enc_kernel_ms, 3), "d2h": round(acc["d2h"] / ne, 3), "deflate": round(deflate_ms, 3),
                       "inflate": round(inflate_ms, 3), "dec_kernel": round(dec_ms, 3), "enc": round(acc["enc"] / ne, 3),
                       "dec": round(acc["dec"] / nd, 3), "gather": round(acc["gather"] / K, 3)},
                "deflate": "device (deflate_kernels.hip, byte-identical to zlib 1.2.11 level 9)" if dev_deflate.value
                else "host libz thread team", "inflate": "device (inflate_kernels.hip, speculative lane-parallel decode)" if dev_inflate.value
                else "host libz thread team",
                "other_encode_slot_setting": other,
                "other_phantom": other_ph,
                "note": "enc/dec = wall time of the C calls; with overlap two encode calls are in flight, so enc includes "
                        "the wait for the device lock",
                "host_threads": zthreads, "host_cpus": os.cpu_count() or 1,
                "compression_ratio": round(2.0 * npx / max(1.0, file_per_launch), 4),
                "single_slice_latency": single},
            "device": info["name"], "verified": verified,
            "sizes_gathered": int(np.asarray(all_sizes).size) if all_sizes is not None else 0,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if multi:
        parallel.barrier()
        _ffi.check(L.cct_comm_destroy())
    if not verified:
        sys.exit(3)


if __name__ == "__main__":
    main()
