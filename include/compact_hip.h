/*
 * compact_hip.h -- C ABI of libcompact_hip.so, the MI355X (gfx950) implementation of the
 * CompaCT per-slice encode/decode hot path.
 *
 * The reference (taaha-khan/2023-CompaCT-Image-Compression) is pure Python and has no
 * FFI of its own: its contract is the class surface codec.core.Encoder / Decoder
 * (src/codec/core.py:170-365, 367-543).  The entry points below are what a ctypes
 * binding for that surface needs; each one names the reference code it replaces.
 * The Python mirror that binds them lives in 2023-compact-image-compression_amd/codec/
 * and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * CCT_E_* code and never throws; buffers are caller-allocated; "d_" pointers are device
 * (HBM) addresses obtained from cct_dev_alloc (or any hipMalloc), "h_" pointers are host
 * addresses.  Threading: calls may come from several host threads.  An encode batch call
 * (cct_encode_batch, cct_encode_batch_packed) takes an internal encode slot -- a HIP stream
 * with its own workspaces; by default there is one, so a second call waits (option
 * "encode_slots" = 2: two batches side by side on the device, which pays on some boxes only,
 * see DESIGN.md); decode calls (cct_decode_batch,
 * cct_zlib_decompress_batch) likewise take one of two decode slots ("decode_slots"), next to
 * the encodes; the size gather (cct_allgather_u32) has its own stream as well.  Everything else
 * shares the main stream (= encode slot 0) under one mutex.  The streams are NOT ordered
 * against each other: every host-facing call is complete when it returns (cct_dev_memset
 * and the h2d/d2h copies included); only cct_encode_payload_dev / cct_decode_payload_dev
 * leave work queued (main / decode stream) -- call cct_sync() before another call reads or
 * overwrites their buffers.  Rare runtime work -- capturing and instantiating a slot's graphs, growing a workspace,
 * creating a stream -- is done with no other call of the library in flight (the other calls finish first; a
 * call that arrives meanwhile waits a few milliseconds): this happens on the first batches of a shape only.
 * The library keeps up to eight HIP streams busy; the runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware
 * queues (default 4) and two streams on one queue run one after the other, so the first device call sets that variable to 8
 * unless the caller has set it (it is read when the HIP runtime initialises: set it yourself if HIP is up before this library).
 * The library initialises HIP lazily on the first device call.  Processes that fork
 * workers (scripts/evaluate.py:107) must fork BEFORE that call: each child then binds the
 * GPU itself.  A child forked after its parent initialised the GPU gets CCT_E_DEVICE from
 * every device call (ROCm cannot share a runtime across fork).
 *
 * There is NO CPU fallback: every function that computes needs a gfx950 device and fails
 * with CCT_E_DEVICE when none is usable.
 */
#ifndef COMPACT_HIP_H
#define COMPACT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCT_ABI_VERSION 1

/* error codes */
#define CCT_OK 0
#define CCT_E_MAGIC 1     /* header magic mismatch          -> ValueError   (core.py:388-389) */
#define CCT_E_ZLIB 2      /* DEFLATE stream invalid          -> zlib.error   (core.py:421)     */
#define CCT_E_OVERFLOW 3  /* pixel left [0,65535] in decode  -> OverflowError(core.py:506,516) */
#define CCT_E_STREAM 4    /* truncated / malformed token stream (TypeError/IndexError there)  */
#define CCT_E_SHAPE 5     /* width*height % block_size != 0  -> ValueError   (core.py:245,429) */
#define CCT_E_CAP 6       /* caller buffer too small */
#define CCT_E_NOMEM 7
#define CCT_E_DEVICE 8    /* no usable gfx950 device / HIP runtime error */
#define CCT_E_ARG 9       /* unsupported argument (block_size not in {4,8,16,32,64}, n<0, ...) */
#define CCT_E_MIXED 10    /* decode batch whose members differ in shape or flags */

/* encoder flags: config['encoder']['transforms'] + deflate_compression (core.py:207-209) */
#define CCT_FLAG_FRACTAL 1u       /* transforms.fractal      (core.py:234) */
#define CCT_FLAG_SEGMENTATION 2u  /* transforms.segmentation (core.py:251) */
#define CCT_FLAG_DEFLATE 4u       /* deflate_compression     (core.py:337) */
#define CCT_FLAG_SIGNED_SEG 8u    /* caller's array dtype is int16: segmentation sees signed
                                     values (core.py:254 image.flatten().tolist()) */

/* per-slice status bits written by the device kernels (0 = clean) */
#define CCT_ST_Q7 1u         /* encode: a traversal delta outside [-2047,2048] was emitted; the
                                stream is reference-identical but not decodable (SURVEY App.A Q7) */
#define CCT_ST_CAP 2u        /* encode: payload stride too small */
#define CCT_ST_OVERFLOW 4u   /* decode: CCT_E_OVERFLOW condition */
#define CCT_ST_STREAM 8u     /* decode: CCT_E_STREAM condition */
#define CCT_ST_ZLIB 16u      /* decode: CCT_E_ZLIB condition (device INFLATE) */

typedef struct cct_header {  /* the 13-byte .cct header (core.py:193-210 / 385-402) */
	int32_t width;             /* image.shape[0] */
	int32_t height;            /* image.shape[1] */
	int32_t channels;
	int32_t bytes_per_channel;
	int32_t fractal, segmentation, deflate;
} cct_header;

typedef struct cct_slice_stats {  /* Encoder.info / partition statistics of one slice */
	uint32_t n_short;      /* info['delta'], core.py:317 */
	uint32_t n_full;       /* info['full'],  core.py:322 */
	uint32_t n_jump;       /* len(BLOCK_JUMPS), cluster.py:166 */
	uint32_t n_difficult;  /* len(block_deltas), cluster.py:51-59 */
} cct_slice_stats;

/* ---- library / device ------------------------------------------------------------- */
int cct_version(void);                 /* returns CCT_ABI_VERSION */
const char *cct_last_error(void);      /* thread-local text of the last failure */
int cct_init(int device);              /* bind HIP device (default: LOCAL_RANK or 0); idempotent */
int cct_shutdown(void);
int cct_device_info(char *name, size_t name_cap, int *compute_units, uint64_t *hbm_bytes);

int cct_dev_alloc(void **d_ptr, size_t bytes);
int cct_dev_free(void *d_ptr);
/* page-locked host memory: archives / file buffers allocated here are copied to and from the device without a
 * staging pass (the batch entry points detect it); plain malloc'ed buffers keep working */
int cct_host_alloc(void **h_ptr, size_t bytes);
int cct_host_free(void *h_ptr);
int cct_h2d(void *d_dst, const void *h_src, size_t bytes);
int cct_d2h(void *h_dst, const void *d_src, size_t bytes);
int cct_dev_memset(void *d_dst, int value, size_t bytes);
int cct_sync(void);
/* elapsed-time markers on the library's stream (bench.py: HIP events around the kernels) */
int cct_event_create(void **ev);
int cct_event_record(void *ev);
int cct_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms); /* synchronises on ev_stop */
int cct_event_destroy(void *ev);

/* ---- traversal -------------------------------------------------------------------- */
/* Replaces GeneralizedHilbertCurve(width, height, get_index=True).generate_all()
 * (src/codec/curve.py:45-138; call sites core.py:234-237, 423-425).  Host code; shape-only. */
int cct_curve_table(int width, int height, int32_t *h_out);

/* ---- sizes ------------------------------------------------------------------------ */
/* bytes one slice's token stream (+EOF) can need; also the stride between slices in d_payload
 * (a multiple of 256). */
size_t cct_payload_stride(int width, int height, int block_size);
/* bytes one .cct file can need (13-byte header + zlib compressBound of the payload) */
size_t cct_file_bound(int width, int height, int block_size);

/* ---- encode ----------------------------------------------------------------------- */
/* Stage (i), the HBM-bound part: traversal -> segmentation/mesh -> delta -> tag-byte pack
 * (+EOF) for n device-resident slices.  Replaces core.py:234-330 + cluster.py:20-199.
 * d_images: n*width*height uint16 (C order, shape (width,height) each).
 * d_payload: n * payload_stride bytes; d_payload_sizes[n]; d_status[n] (CCT_ST_* bits). */
int cct_encode_payload_dev(const uint16_t *d_images, int n, int width, int height,
                           int block_size, uint32_t flags, int eof_byte /* -1 = none */,
                           uint8_t *d_payload, size_t payload_stride,
                           uint32_t *d_payload_sizes, uint32_t *d_status,
                           cct_slice_stats *d_stats /* may be NULL */,
                           uint8_t *d_roles /* may be NULL; n*NB bytes: the block partition
                              (cluster.py:49-199) as one role per traversal block: 0 = emitted alone,
                              1..63 = leader of a meshed pair (BLOCK_JUMPS[b] - b), 0xFF = partner */);

/* Stages (i)+(ii): whole .cct files (header + DEFLATE(level 9) or raw payload) into host
 * memory.  Replaces Encoder.encode (core.py:212-365) for a batch.  Slice i's file is
 * h_out[i*out_stride .. +h_out_sizes[i]).  images_on_device selects d_/h_ meaning of `images`. */
int cct_encode_batch(const uint16_t *images, int images_on_device, int n, int width, int height,
                     int block_size, uint32_t flags, int eof_byte, const char magic[4],
                     int channels, int bytes_per_channel,
                     uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes,
                     uint32_t *h_status /* CCT_ST_* bits per slice */,
                     uint32_t *h_payload_sizes /* may be NULL */,
                     cct_slice_stats *h_stats /* may be NULL */);

/* Same as cct_encode_batch with the files written back to back ("archive" layout, what a corpus
 * run such as scripts/evaluate.py would store): file i = h_archive[h_offsets[i] .. h_offsets[i+1]),
 * h_offsets has n+1 entries; this is exactly the input layout of cct_decode_batch.  Needs
 * CCT_FLAG_DEFLATE and the device DEFLATE path. */
int cct_encode_batch_packed(const uint16_t *images, int images_on_device, int n, int width, int height,
                            int block_size, uint32_t flags, int eof_byte, const char magic[4],
                            int channels, int bytes_per_channel,
                            uint8_t *h_archive, size_t archive_cap, uint64_t *h_offsets,
                            uint32_t *h_out_sizes, uint32_t *h_status,
                            uint32_t *h_payload_sizes /* may be NULL */,
                            cct_slice_stats *h_stats /* may be NULL */);

/* DEFLATE stage alone, on the device: n byte strings (h_in[h_offsets[i] .. h_offsets[i+1])) ->
 * n zlib streams byte-identical to zlib 1.2.11 compress2(level 9), i.e. CPython's
 * zlib.compress(data, level=9) that the reference calls at core.py:340.  Stream i lands at
 * h_out + i*out_stride; out_stride >= compressBound(longest input rounded up to 256) + 64.
 * Option "device_deflate" (default 1) selects this implementation inside cct_encode_batch;
 * 0 runs libz on the host thread team instead. */
int cct_zlib_compress_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n,
                            uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes);

/* INFLATE stage alone, on the device: n zlib streams (h_in[h_offsets[i] .. h_offsets[i+1])) -> the bytes
 * zlib.decompress returns for each (what the reference calls at core.py:421).  Output i lands at
 * h_out + i*out_stride (out_stride a multiple of 16); h_status[i] = CCT_OK, CCT_E_ZLIB (anything libz
 * rejects: bad header, invalid code, distance too far back, truncated stream, Adler-32 mismatch) or
 * CCT_E_CAP (the stream inflates to more than out_stride bytes).  Returns the first non-OK status.
 * Option "device_inflate" (default 1) selects this implementation inside cct_decode_batch. */
int cct_zlib_decompress_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n,
                              uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status);

/* ---- decode ----------------------------------------------------------------------- */
/* Replaces Decoder.read_header (core.py:385-402). */
int cct_read_header(const uint8_t *h_file, size_t len, const char magic[4], cct_header *out);

/* Token stream -> raster for n device-resident payloads of one shape.  Replaces
 * core.py:423-520.  d_payload_sizes[i] counts the trailing EOF byte, which is ignored
 * exactly like ByteReader.padding_len (core.py:136-142). */
int cct_decode_payload_dev(const uint8_t *d_payload, size_t payload_stride,
                           const uint32_t *d_payload_sizes, int n, int width, int height,
                           int block_size, int fractal,
                           uint16_t *d_images, uint32_t *d_status);

/* Whole files -> rasters (Decoder.decode with out_path=None, core.py:404-543) for n files of
 * identical shape/flags laid out back to back: file i = h_files[h_offsets[i] .. h_offsets[i+1]).
 * Output: n*width*height uint16 at `images` (device or host). */
int cct_decode_batch(const uint8_t *h_files, const uint64_t *h_offsets, int n, int block_size,
                     const char magic[4], uint16_t *images, int images_on_device,
                     size_t images_cap_px, uint32_t *h_status /* CCT_E_* code per file */);

/* ---- multi-GPU: the path's only exchange step ----------------------------------------- */
/* Slices shard over GPUs with no data-path collective (scripts/evaluate.py:107-119 treats them as independent units too);
 * what ranks exchange is the per-slice compressed size, so that every rank knows every file's offset in the archive.
 * One process per GPU.  RCCL (librccl.so, loaded on first use) carries the all-gather over xGMI.  Bootstrap: rank 0 calls
 * cct_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the launcher has (bench.py: a file
 * next to the rendezvous port); then every rank calls cct_comm_init.  No PyTorch anywhere. */
#define CCT_COMM_ID_BYTES 128
int cct_comm_unique_id(void *id128);                             /* ncclGetUniqueId */
int cct_comm_init(const void *id128, int rank, int world);        /* ncclCommInitRank on the library's device */
int cct_comm_info(int *rank, int *world);                         /* -1, 0 when no communicator exists */
/* every rank passes its n_local values and the same max_local >= every rank's n_local; h_all receives world * max_local
 * values, rank r's at h_all[r * max_local ..] (the caller trims by the counts it gathers the same way).  Without a
 * communicator: a plain copy (single-process use). */
int cct_allgather_u32(const uint32_t *h_local, int n_local, int max_local, uint32_t *h_all);
int cct_comm_destroy(void);

/* ---- PackBits utility -------------------------------------------------------------- */
/* Replaces PackBits(apply_delta_transform).encode / .decode of src/codec/packbits.py:74-163 (dead code in the reference:
 * nothing imports it, the .cct path never runs it) for n byte strings h_in[h_offsets[i] .. h_offsets[i+1]).
 * Output i lands at h_out + i*out_stride.  Encode needs out_stride >= cct_packbits_bound(longest input); decode reports
 * CCT_E_CAP per string when out_stride is too small and CCT_E_STREAM for a run header without its byte. */
size_t cct_packbits_bound(size_t n_bytes);
int cct_packbits_encode_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, int delta_transform,
                              uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes);
int cct_packbits_decode_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, int delta_transform,
                              uint8_t *h_out, size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status);

/* ---- tuning / introspection (bench.py) --------------------------------------------- */
/* Stage times of the CALLING THREAD's most recent cct_encode_batch / cct_decode_batch, milliseconds (kept per
 * thread: an encode and a decode driven from two threads do not overwrite each other; takes no lock):
 * [0] encode kernel (HIP events on the library stream), [1] packed files device -> host, [2] DEFLATE (HIP events
 * on the device path, host wall on the libz path), [3] INFLATE (likewise), [4] decode kernel (HIP events),
 * [5] reserved. */
int cct_last_timings(float *out6);
/* Options: "encode_slots" / "decode_slots" (1 or 2 batches on the device at a time), "device_deflate" / "device_inflate"
 * (0: that stage on the host thread team of "zlib_threads" threads), "inflate_lanes" (lanes per stream of the INFLATE kernel:
 * 256, 512, or 0 = 512 unless an encode call is in flight when the decode starts; "last_inflate_lanes" reads back the choice),
 * "tile_path", "stream_tpg", "deflate_graph", "deflate_compact_records", "wg_threads", "pipe_tpw", "pipe_timing" (kernel choice and
 * tuning), "decode_yields" / "queue_ahead" (scheduling of pipelined calls; 0 switches them off). */
int cct_set_option(const char *key, int value);
int cct_get_option(const char *key, int *value);

#ifdef __cplusplus
}
#endif
#endif /* COMPACT_HIP_H */
