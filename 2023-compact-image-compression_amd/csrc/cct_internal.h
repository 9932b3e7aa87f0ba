// Internal declarations shared by the host side (api.cpp) and the HIP kernels.
// Not part of the C ABI (see include/compact_hip.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace cct {

// ---- encode kernel geometry ----------------------------------------------------------
constexpr int ENC_CH = 8192;          // pixels per chunk staged in LDS (multiple of every block size)
constexpr int ENC_RING = 4;           // chunks resident in the LDS ring (power of two)
constexpr int ENC_LIST_CAP = 1536;    // difficult-block records kept in LDS; the rest spill to HBM
constexpr int ENC_STG_BYTES = 2 * ENC_CH + ENC_CH / 8 + 64;  // token staging for one chunk + carry
constexpr int ENC_MAX_LDS_ROLE = 48 * 1024;  // role[] lives in LDS when NB <= this

constexpr uint8_t ROLE_PARTNER = 0xFF; // block consumed as the second half of a meshed pair

struct EncArgs {
	const uint16_t *images;   // n * N pixels
	const int32_t *lut;       // traversal order O[N] (device) or nullptr for the identity
	int N, NB;
	int eof;                  // -1: none
	uint32_t flags;           // CCT_FLAG_*
	uint8_t *payload; size_t stride;
	uint32_t *sizes; uint32_t *status;
	uint8_t *ws_role;         // n * NB bytes when role[] does not fit LDS, else nullptr
	uint32_t *ws_lidx;        // n * NB spill: difficult-block index
	uint64_t *ws_lmask;       // n * NB spill: candidate masks
	uint8_t *ws_lcur;         // n * NB spill: cur counts
	uint32_t *stats;          // optional n * 4: short, full, jump tokens, difficult blocks
	uint8_t *roles_out;       // optional n * NB: final role of every block (BLOCK_JUMPS)
	uint32_t dbg_skip;        // tuning only (option "debug_skip"): phases to skip, results then invalid
};

// tile-staged fast path (block_size 16, traversal made of aligned 64x64 tiles)
constexpr int TILE_MAX_TILES = 1024;
constexpr int TILE_MAX_ORIENT = 4;
struct TileEncArgs {
	EncArgs e;
	const uint32_t *tile_org;    // n_tiles: raster index of each tile's top-left pixel
	const uint8_t *tile_orient;  // n_tiles: which pattern table the tile uses
	const uint16_t *patterns;    // n_orient * 4096: LDS byte offset of every traversal position of a tile
	int n_orient, n_tiles, row_pitch;
};
size_t enc_tiles_lds_bytes(int NB, bool *role_in_lds);
hipError_t launch_encode_tiles(const TileEncArgs &ta, int n, hipStream_t s);

// staged pipeline (encode_pipe.hip): analyse -> resolve -> pack, tile-parallel, same applicability as the tile path
// up to 1024x1024 (larger tiled shapes stay on encode_tiles_kernel)
constexpr int PIPE_MAX_NB = 65536;          // blocks per slice (the resolve kernel keeps role[] in LDS)
constexpr int PIPE_DEFAULT_MAX_NB = 16384;  // largest slice for which the pipeline is the default choice (see api.cpp)
constexpr int PIPE_PAIR_REC = 80;           // bytes per meshed-pair record: jump byte + up to 64 token bytes, padded to 16
constexpr uint32_t CCT_ST_INTERNAL = 0x80000000u;  // the kernels disagree about a size: a bug, never a data property
struct PipeTiles {             // small per-shape tables carried IN the kernel arguments: one scalar load, no pointer to chase
	uint32_t orgo[256];          // raster index of each tile's top-left pixel (< 2^24) | tile orientation << 24: ONE scalar load gives
	                             // both (a byte array indexed by the tile became a vector load the next table lookup had to wait for)
	uint32_t last[TILE_MAX_ORIENT];  // raster offset inside the tile of the tile's last traversal position
	uint32_t mid[TILE_MAX_ORIENT];   // the same for position 2047 (the last pixel of the first half tile)
	uint32_t qorg[TILE_MAX_ORIENT];  // raster offset inside the tile of the 32x32-pixel quadrant its first 64 traversal blocks cover
	uint32_t geom[TILE_MAX_ORIENT * 2];  // region of half h of orientation o: bit 0 = vertical split (32x64 pixels), bits 8.. = first
	                                     // block row (horizontal split) or first block-pair column (vertical split)
};
struct PipeArgs {
	EncArgs e;                   // e.lut must be the traversal table
	PipeTiles tiles;
	const uint32_t *ptab;        // n_orient * 128 entries of 4 dwords, one per lane of a tile workgroup (lane = block row * 8 + block pair):
	                             //   [0] traversal block index | orientation << 8 of the left block, the same << 16 for the right block
	                             //   [1], [2] raster offset inside the tile of the pixel that precedes the left / right block in
	                             //   traversal order (0xFFFFFFFF: the block opens the tile)
	const uint32_t *ptab2;       // n_orient * 2 * 64 entries of 4 dwords, one per lane of a half-tile wave (half = 128 traversal blocks):
	                             //   [0..2] as ptab, [3] raster offset inside the tile of the lane's 8x4-pixel region
	const uint32_t *btab;        // n_orient * 256: traversal block of a tile -> raster offset of its top-left pixel | orientation << 24
	const uint32_t *otab;        // 4 * 16 dwords per block orientation: eight v_perm selectors, quadrant choice bits
	const uint32_t *ttab;        // 16 * 4 dwords: token byte selectors and length for the 16 two-byte masks of a 4-pixel group
	int n_orient, n_tiles, row_pitch;
	uint8_t *ssz;                // n * NB: token bytes of every block emitted alone after its traversal predecessor | 0x80 if difficult
	uint64_t *mask;              // n * NB: candidate fit masks, valid for difficult blocks
	uint32_t *tflag;             // n * n_tiles: tile has difficult blocks; then
	uint32_t *tcount;            // n (contiguous with tflag, one memset): tiles listed per slice
	uint32_t *tlist;             // n * n_tiles: the listed tiles of every slice (work of the mask kernel)
	uint8_t *roles;              // n * NB: the block partition
	uint32_t *spec;              // n * NB: leaders: pair record << 8 | group bytes; blocks after a meshed block: predecessor pixel
	uint32_t *toff;              // n * (2 * n_tiles + 1): payload offset of every half tile's first token, then the token total
	uint8_t *pairrec;            // n * (NB / 2) * PIPE_PAIR_REC
	uint32_t *spill_idx;         // n * NB: ordered difficult-block list beyond the LDS capacity of the resolve kernel
};
struct PipeTune {   // tuning runs only
	int tpw;           // tiles per analyse workgroup (0: default)
	float *times_us;   // if set: the four kernels are timed with events (synchronises): [analyse, masks, resolve, pack]
};
hipError_t launch_encode_pipe(const PipeArgs &pa, int n, hipStream_t s, const PipeTune *tune = nullptr);

// single streaming pass (encode_stream.hip): same applicability as the pipeline; one kernel, every pixel read once
constexpr int STREAM_TPG = 4;       // default tiles per group = waves per workgroup (1, 2 or 4: option "stream_tpg")
struct StreamArgs {
	EncArgs e;
	PipeTiles tiles;
	const uint32_t *ptab;        // as PipeArgs::ptab
	const uint32_t *htab;        // n_orient * 32 entries of 2 dwords: the 32 block pairs (8 rows x 4) of the quadrant a tile's first 64
	                             //   traversal blocks cover (origin: tiles.qorg): [0] as ptab[0], [1] raster offset of the pair inside the tile
	const uint32_t *otab, *ttab; // as PipeArgs (ttab entry: selectors, token bytes, kept bits of the low bytes)
	int n_tiles, row_pitch, gps, tpg; // gps: groups per slice, tpg: tiles per group
	int dbg;                     // tuning runs only (CCT_STREAM_DBG): 1 no carry wait, 2 no look-back (results then invalid), 4 wrong group guess (results valid)
	uint64_t *hand;              // n * gps * 4 hand-off words, then
	uint32_t *ticket;            // n group tickets (one allocation: zeroed by one memset before every launch)
	uint64_t *spill_mask;        // n * NB: candidate masks beyond the LDS list of a tile
	uint16_t *spill_idx;         // n * NB: their blocks
	uint8_t *pairrec;            // n * (NB / 2) * PIPE_PAIR_REC: meshed pairs beyond one per lane
};
inline size_t stream_ws_bytes(int n, int gps) { return ((size_t)n * gps * 32 + (size_t)n * 4 + 15) & ~(size_t)15; }
hipError_t launch_encode_stream(const StreamArgs &sa, int n, hipStream_t s);

size_t enc_lds_bytes(int NB, bool *role_in_lds);
hipError_t launch_encode(const EncArgs &a, int n, int block_size, int threads, hipStream_t s);

// ---- decode kernel geometry ----------------------------------------------------------
constexpr int DEC_SEG = 16;           // payload bytes parsed per lane per step
constexpr int DEC_JLIST_CAP = 2048;   // jump records kept in LDS; the rest spill to HBM

struct DecArgs {
	const uint8_t *payload; size_t stride; const uint32_t *sizes;
	const int32_t *lut;       // traversal order or nullptr
	int N, NB;
	uint16_t *images; uint32_t *status;
	uint8_t *ws_role;         // n * NB bytes (always in HBM for decode)
	uint32_t *ws_slot;        // n * NB: stream slot -> leader block | kind << 30
	uint32_t *ws_jord;        // n * (NB/2+1) spill: jump ordinals
	uint8_t *ws_jval;         // n * (NB/2+1) spill: jump distances
	// traversal made of aligned 64x64 tiles (same tables as encode_tiles_kernel): position -> raster offset from LDS
	const uint32_t *tile_org; const uint8_t *tile_orient; const uint16_t *patterns;
	int n_tiles, n_orient, width;   // n_tiles == 0: use lut
	// pass A leaves every lane's view of every parsing step here (pixel ordinal | entry state << 31, value before the
	// segment) so that pass B does not repeat the workgroup scans: n * pcache_steps * threads entries
	uint2 *ws_pcache; int pcache_steps;
};

hipError_t launch_decode(const DecArgs &a, int n, int block_size, int threads, hipStream_t s);

// ---- device DEFLATE (zlib 1.2.11 level 9 restatement, deflate_kernels.hip) ----------------
struct BlockMeta {          // one DEFLATE block of a slice
	uint64_t bit_off;         // absolute bit offset of the 3 header bits inside the slice's output
	uint32_t type;            // 0 stored, 1 static trees, 2 dynamic trees
	uint32_t last;
	uint32_t first_sym, nsym;
	uint32_t in_begin, stored_len;
	uint32_t hdr_nbits, body_bits;
};
struct BlockTables {
	uint16_t lcode[286]; uint8_t llen[286];
	uint16_t dcode[30]; uint8_t dlen[30];
	uint32_t hdr_bits[160];
};
struct DeflateArgs {
	const uint8_t *in; size_t in_stride; const uint32_t *in_sizes;  // token payloads (device)
	uint64_t *rec_in, *rec_out;                                      // n * in_stride each: sort records (deflate_kernels.hip, "Sort records"), between / after the sort passes
	uint32_t pos_mask;                                               // position bits of a record's lower word: 2^22 - 1 (compact records, in_stride < 4 MiB) or all 32
	uint32_t *seg_begin, *seg_end;                                   // n
	void *mr;                                                        // n * in_stride * 8 bytes: match records, valid where they carry the tag *gen
	uint32_t *gen;                                                   // device counter of the passes run on this mr buffer, 1 .. 16383 (deflate_kernels.hip MatchRec)
	uint32_t *heavy_list, *sym, *run_ends;                           // n * in_stride each
	uint32_t *run_counts; int run_chunks;                            // n * 2 * run_chunks right behind sort_hist: run ends / starts per chunk of 1784 positions (run_chunks = chunks per slice)
	uint32_t *sort_hist;                                             // n * 384: per slice the histograms of hash & 255 and of hash >> 8 (dfl_run_len_kernel -> sort passes)
	uint16_t *run_len;                                               // n * in_stride: equal bytes ahead (<= 258) | has_prev << 15
	uint32_t *rec32, *exit_pos, *exit_cnt;                           // n * in_stride each
	uint32_t *blk_entry, *blk_symbase;                               // n * in_stride / 64
	uint32_t *total_syms, *postloop_lit, *n_blocks, *adler, *heavy_count, *deep_count, *run_end_count;  // n
	uint32_t *blk_end;                                               // n * max_blocks
	BlockMeta *block_meta; BlockTables *block_tables;                // n * max_blocks
	int max_blocks;
	uint8_t *out; size_t out_stride; uint32_t *out_sizes;            // whole .cct files (header + zlib stream)
	uint8_t header13[16];
};
hipError_t deflate_init_tables();
size_t deflate_sort_temp_bytes(size_t total, int n);
hipError_t launch_pack(const uint8_t *src, size_t stride, const uint32_t *sizes, int n, uint64_t *offsets, uint8_t *dst,
                       int exact, hipStream_t st);
hipError_t launch_deflate(const DeflateArgs &a, int n, void *sort_temp, size_t sort_temp_bytes, hipStream_t st, hipStream_t side = nullptr,
                          const hipEvent_t *fork_join_events = nullptr);  // side + four events (no timing): independent kernels side by side

// ---- gate between the decode and the encode stream (sched_kernels.hip) ---------------------------
hipError_t launch_gate_bump(uint32_t *word, hipStream_t st);
hipError_t launch_gate_wait(const uint32_t *gate, uint32_t want_pass, uint32_t grace_us, uint32_t timeout_us, hipStream_t st);

// ---- device INFLATE (inflate_kernels.hip) ----------------------------------------------------
struct InflateArgs {
	const uint8_t *in; uint64_t in_total;   // archive bytes on the device (padded to 16), total size
	const uint64_t *offsets; int skip;      // stream i = in[offsets[i] + skip .. offsets[i+1])
	uint8_t *out; size_t out_stride;        // inflated payloads
	uint32_t *out_sizes, *status;           // status: CCT_ST_ZLIB / CCT_ST_STREAM bits
};
// lanes: 256 or 512 per stream (512: faster alone, slower next to an encode batch; inflate_kernels.hip)
hipError_t launch_inflate(const InflateArgs &a, int n, hipStream_t st, int lanes);

// ---- PackBits utility (packbits_kernels.hip) -------------------------------------------------
hipError_t launch_packbits_encode(const uint8_t *d_in, const uint64_t *d_offsets, int n, int delta, uint32_t *d_ws, uint8_t *d_out,
                                  size_t out_stride, uint32_t *d_out_sizes, hipStream_t st);
hipError_t launch_packbits_decode(const uint8_t *d_in, const uint64_t *d_offsets, int n, int delta, uint8_t *d_out, size_t out_stride,
                                  uint32_t *d_out_sizes, uint32_t *d_status, hipStream_t st);

}  // namespace cct
