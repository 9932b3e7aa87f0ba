// Device DEFLATE, byte-identical to zlib 1.2.11 `deflate(level 9, wbits 15, memLevel 8, default
// strategy, one-shot Z_FINISH)` -- the stream CPython's zlib.compress(data, level=9) produces for
// the reference (src/codec/core.py:340).  zlib is a third-party dependency of the reference and is
// not vendored there; the algorithm is restated here in data-parallel form (CPU model of the same
// restatement, pinned against libz: oracle/deflate_model.c).
//
//   sort      stable two-pass LSD radix sort of (hash, position) per slice, the 15-bit rolling hash of 3 bytes
//             (deflate.c UPDATE_HASH) computed on the fly: a bucket in position order IS zlib's hash chain
//             (head/prev), walked backwards
//   runs      ordered list of the runs of >= 3 equal bytes and, per position, the equal bytes ahead
//   match     what longest_match() returns over the first 4096 / 1024 chain entries (max_chain, and
//             max_chain>>2 once prev_length >= good_match), with the NIL / MAX_DIST / lookahead rules of
//             deflate.c: one lane per position for short chains, one wave per position for long ones, and
//             the run list instead of the chain for strings that start a run (their bucket holds every
//             position of every run of that byte)
//   parse     deflate_slow's lazy evaluation as a walk over "decision positions": per position the
//             deferral chain is resolved locally, 64-position blocks are summarised by pointer
//             doubling (entry -> exit, symbol count), the block-to-block hop chain is walked
//             speculatively by 256 lanes per slice
//   symbols   visited positions emit literals / (length, distance) pairs in stream order
//   trees     per 16383-symbol block: histograms, then build_tree / gen_bitlen / gen_codes /
//             scan_tree / build_bl_tree exactly as trees.c (the heap is replayed: ties are decided by heap
//             position), stored / static / dynamic choice
//   emit      code bits of every symbol at its prefix-summed bit offset; zlib header, Adler-32
// The launches of one pass are captured into a HIP graph by the caller (api.cpp).
#include <hip/hip_runtime.h>
#include <cstring>
#include <algorithm>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

constexpr int MIN_MATCH = 3, MAX_MATCH = 258, WSIZE = 32768;
constexpr int MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1;
constexpr int MAX_DIST = WSIZE - MIN_LOOKAHEAD;  // 32506
constexpr int TOO_FAR = 4096;
constexpr int BLOCK_SYMS = 16383;                // lit_bufsize - 1 (memLevel 8)
constexpr int L_CODES = 286, D_CODES = 30, BL_CODES = 19, HEAP_SIZE = 2 * L_CODES + 1;
constexpr int END_BLOCK = 256, MAX_BITS = 15, MAX_BL_BITS = 7;

__constant__ uint8_t c_extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint8_t c_extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t c_extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
__constant__ uint8_t c_bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
// tr_static_init tables, filled by the host once (cct::deflate_init_tables)
__constant__ uint8_t c_length_code[256];
__constant__ uint16_t c_base_length[29];
__constant__ uint8_t c_dist_code[512];
__constant__ uint16_t c_base_dist[30];
__constant__ uint16_t c_static_lcode[288];
__constant__ uint8_t c_static_llen[288];
__constant__ uint16_t c_static_dcode[30];

__device__ __forceinline__ int d_code(uint32_t dist) { return dist < 256 ? c_dist_code[dist] : c_dist_code[256 + (dist >> 7)]; }

struct MatchRec { uint16_t len4096, len1024, dist4096, dist1024; };
// Match records are only stored where there is something to say (a match of >= MIN_MATCH bytes): a record counts when it
// carries the tag of the current pass -- 14 bits in the spare upper bits of the two length fields (lengths are <= 258), taken
// from a device counter that dfl_offsets_kernel advances at the start of every pass (1 .. 16383; the host clears the buffer
// before the counter would come round, api.cpp deflate_locked).  Writing a "no match" record for every position instead cost
// 0.17 ms per batch (555 MB of stores), as much as the scattered stores it had saved in the match kernel.
constexpr uint32_t GEN_MAX = 16383;
__device__ __forceinline__ uint2 pack_match(uint32_t len4096, uint32_t len1024, uint32_t dist4096, uint32_t dist1024, uint32_t gen)
{
	return make_uint2((len4096 | ((gen & 127u) << 9)) | ((len1024 | ((gen >> 7) << 9)) << 16), dist4096 | (dist1024 << 16));
}
// The record of position p as the parse reads it: the stored one if it carries this pass's tag; else "no match" -- or, deep
// inside a run (in[p-1] == in[p] and at least max_len equal bytes ahead), the chain head p-1 at the cap, which nobody stores:
// rw = run-length word of p (dfl_run_len_kernel).
__device__ __forceinline__ MatchRec checked_match(MatchRec r, uint32_t gen, uint32_t rw, uint32_t p, uint32_t L)
{
	const uint32_t tag = (uint32_t)(r.len4096 >> 9) | ((uint32_t)(r.len1024 >> 9) << 7);
	const uint32_t lookahead = L - min(p, L);
	const uint32_t max_len = lookahead < (uint32_t)MAX_MATCH ? lookahead : (uint32_t)MAX_MATCH;
	const bool maximal = (rw >> 15) && (rw & 0x7FFFu) >= max_len && p + 2 < L;
	MatchRec o;
	const bool ok = tag == gen;
	o.len4096 = ok ? (uint16_t)(r.len4096 & 511u) : (uint16_t)(maximal ? max_len : 0u);
	o.len1024 = ok ? (uint16_t)(r.len1024 & 511u) : (uint16_t)(maximal ? max_len : 0u);
	o.dist4096 = ok ? r.dist4096 : (uint16_t)(maximal ? 1u : 0u);
	o.dist1024 = ok ? r.dist1024 : (uint16_t)(maximal ? 1u : 0u);
	return o;
}

// ------------------------------------------------------------------ 1. hash + sort by (hash, position)
// The chain of a string = the earlier strings with the same 15-bit hash, most recent first (deflate.c
// INSERT_STRING / prev[]).  Sorting the positions of a slice by (hash, position) lays every chain out as a
// contiguous run.  Stable LSD radix sort in two passes (hash & 255, then hash >> 8); ONE workgroup of 1024 lanes
// per slice walks its slice tile by tile with running digit offsets in LDS, so there is no cross-workgroup
// scan: pass A hashes the input on the fly (value = position = index) and also counts pass B's digits.
// Per tile: wave-level match-any gives each lane its rank among the equal digits of its wave, the 16 per-wave
// counts are prefix-summed per digit, and lanes scatter to offset[digit] + wave prefix + rank.  All waves of a
// slice write the same 256 (128) output streams tile after tile, which keeps the partially written lines in L2.
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would serialise the
// scattered stores of a tile with its barriers
__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
	return v;
}

template <bool FIRST>
__global__ void __launch_bounds__(1024) dfl_sort_pass_kernel(DeflateArgs a)
{
	constexpr int BITS = FIRST ? 8 : 7, NB = 1 << BITS;
	__shared__ uint32_t offs[256];
	__shared__ uint32_t wcnt[16][NB];
	const int s = blockIdx.x;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const size_t base = (size_t)s * a.in_stride;
	const uint32_t npos = L >= MIN_MATCH ? L - 2 : 0;
	// one 8-byte record per string: position | hash << 32.  The scatter below is bound by store transactions (every lane
	// of a store instruction hits another output stream), so key and value travel in one store, and the match kernels
	// get both with one scattered load
	const uint64_t *rec_src = a.rec_in + base;
	uint64_t *rec_dst = (FIRST ? a.rec_in : a.rec_out) + base;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	if (tid < 256) offs[tid] = 0;
	__syncthreads();
	// digit histogram of this pass: counted by dfl_run_len_kernel, which reads the input in position order with the whole chip
	// (here one workgroup per slice would wait for its loads, and pass A paid an LDS atomic per element for pass B's histogram)
	if (tid < NB) offs[tid] = a.sort_hist[(size_t)s * 384 + (FIRST ? 0 : 256) + tid];
	__syncthreads();
	if (wave == 0) {  // exclusive scan of up to 256 bins: 4 per lane
		uint32_t v[4], sum = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) { v[k] = offs[lane * 4 + k]; sum += v[k]; }
		uint32_t inc = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
		uint32_t run = inc - sum;
#pragma unroll
		for (int k = 0; k < 4; k++) { offs[lane * 4 + k] = run; run += v[k]; }
	}
	__syncthreads();
	// A tile = 4096 elements: wave w owns elements [t0 + 256 w, t0 + 256 w + 256) and ranks them in four rounds of
	// 64 (running per-digit counts in its own LDS row), so the cross-wave prefix and its two barriers are paid
	// once per 4096 elements.  Loads of the next tile are issued before the current one is ranked.
#ifndef CCT_SORT_E
#define CCT_SORT_E 4  // pass B alone, per batch: 2: 366, 4: 359, 8: 330-343 us, 12 spills (1318 us) -- but with 8 the kernel needs 116
                      // VGPRs, a CU no longer holds its 16 waves next to an INFLATE workgroup, and in the pipelined bench the DEFLATE pass
                      // took 4.67 instead of 3.5 ms (13.3 K instead of 17.4 K MPixels/s): 4 (72 VGPRs)
#endif
	constexpr int E = CCT_SORT_E;
	constexpr int ND = E / 4;  // pass A: input dwords per lane (a wave's tile = 64 E positions = 16 E dwords)
	// Loads of the next tile are issued before the current one is ranked and taken after it.  Two things the compiler did with
	// the obvious code: inside a conditional it waits for every load right where it is issued (s_waitcnt vmcnt(0) before the
	// join), which made the "prefetch" four full memory round trips in a row per tile; and it moved the first use of the
	// loaded values past this tile's scattered stores, where waiting for the loads means waiting for the stores too (one
	// in-order counter).  So: the index is clamped instead of tested, the raw values pass through an empty asm right before
	// the stores (that is where the wait lands), and hash / position are derived after it.
	const uint32_t last = npos ? npos - 1 : 0;
	// Pass A reads the input as aligned dwords: the 64 E positions of a wave's tile need bytes [B, B + 64 E + 4), lane l loads the dwords
	// at B + 256 d + 4 l and everybody the one after the tile; a position takes its two dwords from the lanes that hold them (ds_bpermute) and
	// shifts its five bytes out.  (One byte-granular load per position and field was tried first: the pass went 0.44 -> 0.59 ms,
	// the address unit handles such loads lane by lane.)  Bytes past the end of the slice are whatever the buffer holds: the hash
	// of a valid position (p < L - 2) never sees them, and the match kernel does not use the compact fields of the last positions.
	static_assert(E % 4 == 0 && ND + 1 <= E, "pass A: whole dwords per lane, kept in ra[0 .. ND]");
#ifdef CCT_SORT_PROBE  // tuning builds only (results invalid)
	const bool compact = false;
#else
	const bool compact = a.pos_mask != 0xFFFFFFFFu;
#endif
	const uint32_t dmax = (uint32_t)a.in_stride - 4u;
	// tile_base = first element of this wave's 256; ra/rb = the values as loaded: anything computed here is computed (and waited for) early
	auto fetch = [&](uint32_t tile_base, uint32_t (&ra)[E], uint32_t (&rb)[E]) {
		if (FIRST) {
#pragma unroll
			for (int d = 0; d < ND; d++) ra[d] = *reinterpret_cast<const uint32_t *>(in + min(tile_base + 256u * d + 4u * (uint32_t)lane, dmax));
			ra[ND] = *reinterpret_cast<const uint32_t *>(in + min(tile_base + 256u * ND, dmax));  // the dword after the tile, the same in every lane
#pragma unroll
			for (int e = 0; e < E; e++) rb[e] = compact ? a.run_len[base + min(tile_base + (uint32_t)(e * 64 + lane), last)] : 0u;  // see finish
		} else {
#pragma unroll
			for (int e = 0; e < E; e++) {
				const uint64_t r = rec_src[min(tile_base + (uint32_t)(e * 64 + lane), last)];
				ra[e] = (uint32_t)r; rb[e] = (uint32_t)(r >> 32);
			}
		}
	};
	// h = upper record word (hash in bits 0..14), p = lower record word (position in the bits of a.pos_mask): see "Sort records"
	auto finish = [&](uint32_t tile_base, const uint32_t (&ra)[E], const uint32_t (&rb)[E], uint32_t (&h)[E], uint32_t (&p)[E]) {
#pragma unroll
		for (int e = 0; e < E; e++) {
			const uint32_t idx = tile_base + (uint32_t)(e * 64 + lane);
			if (FIRST) {
				const int la = (e & 3) * 16 + (lane >> 2);  // lane holding the dword of byte idx, in register e / 4
				const uint32_t da = (uint32_t)__builtin_amdgcn_ds_bpermute(la << 2, (int)ra[e >> 2]);
				uint32_t db = (uint32_t)__builtin_amdgcn_ds_bpermute(((la + 1) & 63) << 2, (int)ra[e >> 2]);
				if ((e & 3) == 3) {  // the last lanes take the next dword from the next register's lane 0 (or from the dword after the tile)
					const uint32_t nxt = (e >> 2) + 1 < ND ? (uint32_t)__builtin_amdgcn_readfirstlane((int)ra[(e >> 2) + 1 < ND ? (e >> 2) + 1 : 0]) : ra[ND];
					if (la == 63) db = nxt;
				}
				const uint32_t sh = (uint32_t)lane & 3u;
				const uint32_t w = __builtin_amdgcn_alignbyte(db, da, sh);  // bytes idx .. idx + 3
				const uint32_t b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u;
				h[e] = ((b0 << 10) ^ (b1 << 5) ^ b2) & 0x7FFFu; p[e] = idx;
				if (compact) {
					// bytes 3 and 4 -- or, for a string that starts with three equal bytes (those go to the run matcher, nobody
					// compares their bytes 3 and 4), the run-length word of the position, which saves the match kernel a scattered load
					const uint32_t b4 = (db >> (8u * sh)) & 255u;
					const uint32_t f16 = (b0 == b1 && b1 == b2) ? rb[e] : ((w >> 24) | (b4 << 8));
					h[e] |= (f16 << 15) | ((b0 >> 7) << 31); p[e] |= (b1 << 22) | (((b0 >> 5) & 3u) << 30);
				}
			}
			else { h[e] = rb[e]; p[e] = ra[e]; }
			if (idx >= npos) { h[e] = 0; p[e] = idx; }
		}
	};
	uint32_t hn[E], pn[E];
	{
		uint32_t ra[E], rb[E];
		fetch((uint32_t)(wave * 64 * E), ra, rb);
		finish((uint32_t)(wave * 64 * E), ra, rb, hn, pn);
	}
	for (uint32_t t0 = 0; t0 < npos; t0 += 1024 * E) {
		uint32_t h[E], p[E], rk[E], ra[E], rb[E];
		const uint32_t idx0 = t0 + (uint32_t)(wave * 64 * E + lane);
#pragma unroll
		for (int e = 0; e < E; e++) { h[e] = hn[e]; p[e] = pn[e]; }
		const uint32_t next_base = t0 + (uint32_t)(1024 * E + wave * 64 * E);
		fetch(next_base, ra, rb);
		for (int k = lane; k < NB; k += 64) wcnt[wave][k] = 0;
#pragma unroll
		for (int e = 0; e < E; e++) {
			const bool valid = idx0 + e * 64 < npos;
			const uint32_t d = FIRST ? (h[e] & 255u) : ((h[e] & 0x7FFFu) >> 8);
			uint64_t same = __ballot(valid);  // lanes of this round with the same digit
#pragma unroll
			for (int b = 0; b < BITS; b++) {
				const bool bit = (d >> b) & 1u;
				const uint64_t bal = __ballot(bit);
				same &= bit ? bal : ~bal;
			}
			const uint32_t rank = (uint32_t)__popcll(same & lt_mask);
			uint32_t before = 0;
			if (valid) {
				before = wcnt[wave][d];  // equal digits of this wave's earlier rounds
				if (rank == 0) wcnt[wave][d] = before + (uint32_t)__popcll(same);
			}
			rk[e] = before + rank;
		}
		lds_barrier();
		if (tid < NB) {  // running offset of digit tid, handed out wave by wave
			uint32_t c[16], run = offs[tid];
#pragma unroll
			for (int w = 0; w < 16; w++) c[w] = wcnt[w][tid];
#pragma unroll
			for (int w = 0; w < 16; w++) { wcnt[w][tid] = run; run += c[w]; }
			offs[tid] = run;
		}
		lds_barrier();
#pragma unroll
		for (int e = 0; e < E; e++) asm volatile("" : "+v"(ra[e]), "+v"(rb[e]) :: "memory");
		finish(next_base, ra, rb, hn, pn);
#pragma unroll
		for (int e = 0; e < E; e++) {
			if (idx0 + e * 64 < npos) {
				const uint32_t d = FIRST ? (h[e] & 255u) : ((h[e] & 0x7FFFu) >> 8);
				const uint32_t dst = wcnt[wave][d] + rk[e];
				rec_dst[dst] = (uint64_t)p[e] | ((uint64_t)h[e] << 32);
			}
		}
	}
}

// ------------------------------------------------------------------ 2. longest_match for every position
// The chain of sorted index i = sorted entries i-1, i-2, ... of the same hash (deflate.c:1236-1386).
// Pass A: one lane per position walks at most LIGHT_STEPS candidates; positions whose chain is
// longer ("heavy": long runs of one byte put tens of thousands of strings in one bucket) are queued.
// Pass B: one WAVE per heavy position, 64 candidates per step.
constexpr int LIGHT_STEPS = 12;

// Block -> (slice, part) for the kernels that touch one slice's arrays at random (hash order): match records
// are scattered 8-byte stores and the string loads are scattered too, so they only stay on chip if the whole
// working set of a slice (~3 MB) sits in ONE L2 while it is being processed.  Blocks are dealt round-robin
// over the 8 XCDs (observed behaviour, used for speed only): blocks lin and lin + 8 share an L2, so the blocks
// with lin % 8 == c walk through slices c, c + 8, c + 16, ... one after the other.  gridDim.y is n rounded up to 8.
__device__ __forceinline__ bool xcd_slice(int n, int &s, uint32_t &part, uint32_t &nparts)
{
	const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y;
	const uint32_t c = lin & 7u, k = lin >> 3;
	nparts = gridDim.x;
	s = (int)(8u * (k / nparts) + c);
	part = k % nparts;
	return s < n;
}

__device__ __forceinline__ uint32_t nil_candidate(uint32_t p, uint32_t lookahead)
{
	// slide_hash NIL quirk at the very end of the input (see oracle/deflate_model.c)
	return (lookahead < (uint32_t)MIN_LOOKAHEAD && p >= 32506u + 32768u && (p - 32506u) % 32768u == 0) ? p - 32506u : 0xFFFFFFFFu;
}

// ------------------------------------------------------------------ 1b. run length ahead of every position
// run_len[p] = number of bytes equal to in[p] starting at p, capped at MAX_MATCH (and by the end of the input),
// with bit 15 = (p >= 2 && in[p-1] == in[p]).  The match kernels visit positions in hash order, where
// scanning the bytes would be a chain of uncoalesced loads; here the input is read once, in order.
// A workgroup scans 2048 positions (8 per lane) and publishes the first RUNLEN_OUT of them: for those the
// scanned tail (>= 264 bytes) decides every length below the cap.
constexpr int RUNLEN_OUT = 2048 - 264;

__global__ void __launch_bounds__(256) dfl_run_len_kernel(DeflateArgs a)
{
	__shared__ uint32_t wtot[4];  // per wave: has_change << 31 | distance from the wave's first position to its first change
	__shared__ uint32_t hist[384];  // digits of the sort: hash & 255 of every string of this workgroup's positions, then hash >> 8
	for (int t = threadIdx.x; t < 384; t += 256) hist[t] = 0;
	__syncthreads();
	const int s = blockIdx.y;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	uint16_t *rl = a.run_len + (size_t)s * a.in_stride;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t nchunks = (L + RUNLEN_OUT - 1) / RUNLEN_OUT;
	for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
		const uint32_t c0 = c * RUNLEN_OUT;
		const uint32_t g = c0 + threadIdx.x * 8;  // first of this lane's 8 positions
		uint8_t b[12];                            // in[g-1 .. g+10]
		if (g + 12 <= L && g >= 1) {
			uint64_t w;
			__builtin_memcpy(&w, in + g, 8);
#pragma unroll
			for (int k = 0; k < 8; k++) b[k + 1] = (uint8_t)(w >> (8 * k));
			b[0] = in[g - 1];
			uint32_t w2; __builtin_memcpy(&w2, in + g + 8, 4);
			b[9] = (uint8_t)w2; b[10] = (uint8_t)(w2 >> 8); b[11] = (uint8_t)(w2 >> 16);
		} else {
#pragma unroll
			for (int k = 0; k < 12; k++) { const int64_t y = (int64_t)g - 1 + k; b[k] = (y >= 0 && y < (int64_t)L) ? in[y] : 0; }
		}
		// the sort's digit histograms (UPDATE_HASH of the three bytes from every published position that starts a string).
		// A quarter of a CT payload is runs of one byte: a lane whose strings all hash alike adds them in one go, and so does
		// a wave (every lane adding 1 to the same two counters made this kernel 0.31 instead of 0.09 ms)
		{
			uint32_t hk[8], cnt = 0, h0 = 0;
			bool same = true;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const bool v = threadIdx.x * 8 + k < (uint32_t)RUNLEN_OUT && g + k + 2 < L;
				hk[k] = v ? (((uint32_t)b[k + 1] << 10) ^ ((uint32_t)b[k + 2] << 5) ^ (uint32_t)b[k + 3]) & 0x7FFFu : 0xFFFFFFFFu;
				if (v) { if (!cnt) h0 = hk[k]; else same &= hk[k] == h0; cnt++; }
			}
			const bool wave_same = __all(same && cnt == 8 && h0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)h0));
			if (wave_same) {
				if (lane == 0) { atomicAdd(&hist[h0 & 255u], 512u); atomicAdd(&hist[256u + (h0 >> 8)], 512u); }
			} else if (same) {
				if (cnt) { atomicAdd(&hist[h0 & 255u], cnt); atomicAdd(&hist[256u + (h0 >> 8)], cnt); }
			} else {
#pragma unroll
				for (int k = 0; k < 8; k++)
					if (hk[k] != 0xFFFFFFFFu) { atomicAdd(&hist[hk[k] & 255u], 1u); atomicAdd(&hist[256u + (hk[k] >> 8)], 1u); }
			}
		}
		uint32_t chg = 0;  // bit k: the run containing position g+k ends at g+k
#pragma unroll
		for (int k = 0; k < 8; k++) chg |= (uint32_t)(b[k + 1] != b[k + 2] || g + k + 1 >= L) << k;
		// suffix scan over lanes of (has_change, distance to first change): spans combine as
		//   (cA, nA) . (cB, nB) = (cA | cB, cA ? nA : nA + nB)       with n = 8 * span for a change-free span
		uint32_t hc = chg != 0, n = hc ? (uint32_t)__ffs((int)chg) - 1u : 8u;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t hb = (uint32_t)__shfl_down((int)hc, d, 64), nb = (uint32_t)__shfl_down((int)n, d, 64);
			if (lane + d < 64 && !hc) { n += nb; hc = hb; }
		}
		__syncthreads();  // wtot of the previous chunk has been read by everyone
		if (lane == 0) wtot[wave] = (hc << 31) | n;
		__syncthreads();
		uint32_t xh = 0, xn = 0;  // the waves after this one
		for (int w = 3; w > wave; w--) {
			const uint32_t t = wtot[w];
			if (t >> 31) { xh = 1; xn = t & 0x7FFFFFFFu; } else xn += t & 0x7FFFFFFFu;
		}
		if (!xh) xn += 1024;  // nothing scanned beyond: only reached by positions that are not published
		const uint32_t nfull = hc ? n : n + xn;  // distance from g to the first change at or after g
		uint32_t nnext = (uint32_t)__shfl_down((int)nfull, 1, 64);  // the same for g + 8
		if (lane == 63) nnext = xn;
		uint16_t out[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t rest = chg >> k;
			uint32_t r = rest ? (uint32_t)__ffs((int)rest) : (8u - k) + nnext + 1u;
			if (r > (uint32_t)MAX_MATCH) r = MAX_MATCH;
			out[k] = (uint16_t)(r | ((g + k >= 2 && b[k] == b[k + 1]) ? 0x8000u : 0u));
		}
		if (threadIdx.x * 8 < (uint32_t)RUNLEN_OUT && g < L) {
			if (g + 8 <= L && threadIdx.x * 8 + 8 <= (uint32_t)RUNLEN_OUT) __builtin_memcpy(rl + g, out, 16);
			else {
#pragma unroll
				for (int k = 0; k < 8; k++) if (g + k < L && threadIdx.x * 8 + k < (uint32_t)RUNLEN_OUT) rl[g + k] = out[k];
			}
		}
		// entries this chunk contributes to the lists of run ends and run starts (dfl_run_lists_kernel): position p ends a run of
		// >= 3 three bytes on (exactly three equal bytes from p, the fourth inside the input) and starts one when >= 3 equal bytes
		// follow and the byte before differs
		{
			// from the "changes after this byte" bits of in[g-1 .. g+9] (bit i: in[g-1+i] != in[g+i], or the right byte is past the
			// end, or -- bit 0 -- there is no byte before position 0): position p = g+k, pair p <-> bit k+1
			uint32_t cx = g == 0 ? 1u : 0u;
#pragma unroll
			for (int i = 0; i < 11; i++) cx |= (uint32_t)(b[i] != b[i + 1] || g + (uint32_t)i >= L) << i;
			const uint32_t t8 = threadIdx.x * 8;
			const uint32_t npub = t8 < (uint32_t)RUNLEN_OUT ? min(min(8u, (uint32_t)RUNLEN_OUT - t8), g < L ? L - g : 0u) : 0u;  // published positions of the lane
			const uint32_t pubm = (1u << npub) - 1u;
			const uint32_t ne3 = g + 3 < L ? min(8u, L - 3u - g) : 0u;                 // positions with p + 3 < L
			const uint32_t run3 = ~(cx >> 1) & ~(cx >> 2);                             // three equal bytes from p
			const uint32_t me = run3 & (cx >> 3) & pubm & ((1u << ne3) - 1u);          // ... and the fourth differs, inside the input
			const uint32_t ms = run3 & cx & pubm;                                      // ... and the byte before differs
			uint32_t ce = 0, cs = 0;  // wave totals; run boundaries are rare (inside a long run there is none): most waves skip this
			if (__any((me | ms) != 0)) { ce = wave_sum((uint32_t)__popc(me)); cs = wave_sum((uint32_t)__popc(ms)); }
			if (lane == 0 && (ce | cs)) {  // (counters zeroed with the sort histograms: no barrier, no store where there is no run)
				uint32_t *cnt = a.run_counts + (size_t)s * 2 * a.run_chunks;
				if (ce) atomicAdd(&cnt[2 * c], ce);
				if (cs) atomicAdd(&cnt[2 * c + 1], cs);
			}
		}
	}
	__syncthreads();
	for (int t = threadIdx.x; t < 384; t += 256)
		if (hist[t]) atomicAdd(&a.sort_hist[(size_t)s * 384 + t], hist[t]);
}

// length of the common prefix of x and y, continuing from len, capped at cap; 8 bytes per step
__device__ __forceinline__ int common_prefix(const uint8_t *x, const uint8_t *y, int len, int cap)
{
	while (len + 8 <= cap) {
		uint64_t u, v;
		__builtin_memcpy(&u, x + len, 8);
		__builtin_memcpy(&v, y + len, 8);
		const uint64_t d = u ^ v;
		if (d) return len + ((__ffsll((long long)d) - 1) >> 3);
		len += 8;
	}
	while (len < cap && x[len] == y[len]) len++;
	return len;
}

// Sort records.  Wide (any slice size): position | hash << 32.  Compact (slices below 4 MiB, a.pos_mask = 2^22 - 1): the record
// also identifies the first five bytes b0..b4 of its string,
//   lower word: position (22 bits) | b1 << 22 | (b0 >> 5 & 3) << 30        upper word: hash (15 bits) | b3 << 15 | b4 << 23 | (b0 >> 7) << 31
// Two strings of one bucket start with the same three bytes exactly when b1 and the top three bits of b0 agree (the hash then
// pins b2 and the rest of b0), so the match kernel decides every candidate whose match is shorter than five bytes -- nearly all
// of them on token payloads -- from the records alone, which are read in sorted order (coalesced); before, every position paid
// a scattered load for its own string and one per candidate.
__device__ __forceinline__ uint32_t rec_hash(uint64_t r) { return (uint32_t)(r >> 32) & 0x7FFFu; }
__device__ __forceinline__ uint32_t rec_pos(uint64_t r, uint32_t pos_mask) { return (uint32_t)r & pos_mask; }
constexpr uint32_t COMPACT_POS_BITS = 22, COMPACT_POS_MASK = (1u << COMPACT_POS_BITS) - 1u;

// longest_match over at most LIGHT_STEPS chain entries, strings read from the input (wide records, and the last seven positions
// of a slice with compact ones).  kind: 0 = best/best_q hold the result, 1 = heavy (longer chain), 2 = starts a run (run_r).
__device__ __forceinline__ void light_match_from_memory(const uint8_t *in, const uint64_t *recs, const uint16_t *rl, uint32_t pos_mask,
                                                        uint32_t L, uint32_t i, uint32_t p, uint32_t h,
                                                        int &kind, int &best, uint32_t &best_q, uint32_t &run_r)
{
	const uint32_t lookahead = L - p;
	const int max_len = lookahead < (uint32_t)MAX_MATCH ? (int)lookahead : MAX_MATCH;
	const uint32_t nil_q = nil_candidate(p, lookahead);
	int count = 0;
	const uint8_t *sp = in + p;
	// strings that START a run of three equal bytes share one bucket with every other position of
	// every run of that byte (tens of thousands of entries); dfl_match_run_kernel evaluates them from
	// the list of run ends instead of walking the chain
	// the first 8 bytes of the string decide most things; in hash order every access to the slice is its own
	// L2 transaction, so the run-length word is only fetched for the strings that need it
	const bool wide = p + 8 <= L;  // then max_len >= 8 and sp[0..7] is inside the input
	uint64_t ow = 0;
	if (wide) __builtin_memcpy(&ow, sp, 8);
	const bool in_run = wide ? (((ow >> 8) ^ ow) & 0xFFFFu) == 0 : (sp[1] == sp[0] && sp[2] == sp[0]);
	if (in_run) {
		const uint32_t rw = rl[p];
		run_r = rw & 0x7FFFu;  // >= 3, <= max_len by construction
		if ((rw >> 15) && (int)run_r >= max_len) { best = max_len; best_q = p - 1; }  // chain head p-1 is already maximal
		else kind = 2;
		return;
	}
	for (int64_t j = (int64_t)i - 1; j >= 0; j--) {
		const uint64_t rj = recs[j];
		if (rec_hash(rj) != h) break;
		if (count == LIGHT_STEPS) { kind = 1; break; }           // heavy: finish cooperatively
		const uint32_t q = rec_pos(rj, pos_mask);
		const uint32_t dist = p - q;
		if (q == 0 || q == nil_q) break;                         // NIL ends the chain
		if (count == 0 ? dist > (uint32_t)MAX_DIST : dist >= (uint32_t)MAX_DIST) break;
		const uint8_t *mp = in + q;
		if (best < max_len) {                                    // only a longer match can replace the best
			int len = 0;
			if (wide) {  // one 8-byte load decides most candidates (the loads of mp are the uncoalesced ones)
				if (best < 8 || mp[best] == sp[best]) {
					uint64_t cw;
					__builtin_memcpy(&cw, mp, 8);
					const uint64_t d = cw ^ ow;
					len = d ? (__ffsll((long long)d) - 1) >> 3 : common_prefix(mp, sp, 8, max_len);
				}
			} else if (mp[best] == sp[best]) len = common_prefix(mp, sp, 0, max_len);
			if (len > best) { best = len; best_q = q; }
		}
		count++;
		if (best >= max_len) break;                              // len >= nice_match
	}
}

// v of the lane below; lane 0 takes `fill` (DPP wave_shr:1 -- every lane of the wave must be enabled where this is called)
__device__ __forceinline__ uint32_t wave_shift_up(uint32_t v, uint32_t fill)
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}

template <bool CMP>
__global__ void dfl_match_kernel(DeflateArgs a, int n)
{
	int s; uint32_t part, nparts;
	if (!xcd_slice(n, s, part, nparts)) return;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const size_t base = (size_t)s * a.in_stride;
	const uint32_t npos = L >= MIN_MATCH ? L - 2 : 0;
	const uint64_t *recs = a.rec_out + base;  // sorted records, see "Sort records"
	uint2 *mr = reinterpret_cast<uint2 *>(a.mr) + base;
	uint32_t *heavy = a.heavy_list + base;
	const uint16_t *rl = a.run_len + base;
	const int lane = threadIdx.x & 63;
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	const uint32_t gen = *a.gen;
	for (uint32_t i0 = part * blockDim.x; i0 < npos; i0 += nparts * blockDim.x) {  // wave-uniform trip count
		const uint32_t i = i0 + threadIdx.x;
		const bool valid = i < npos;
		int kind = 0;  // 0: record written, 1: queue for the cooperative heavy pass, 2: queue for the run pass
		int best = 0;
		uint32_t best_q = 0, run_r = 0;
		const uint64_t ri = recs[min(i, npos - 1)];
		const uint32_t p = rec_pos(ri, CMP ? COMPACT_POS_MASK : 0xFFFFFFFFu);
		const uint32_t h = rec_hash(ri);
		bool from_memory = valid;
		bool found = false;  // something to store (see MatchRec)
		if (CMP) {
			// The chain of sorted index i is i-1, i-2, ...: the records of the lanes below, then of the 64 indices before the wave.
			// Both sit in registers (one coalesced load each) and move up one lane per step, so a step waits for nothing; with
			// one load per step a wave paid a memory round trip for every entry of its longest chain.
			const uint64_t rb = recs[min(i >= 64 ? i - 64 : 0u, npos - 1)];  // lane l: index (wave's first) + l - 64, unused where that is negative
			const uint32_t lo_i = (uint32_t)ri, hi_i = (uint32_t)(ri >> 32), lo_b = (uint32_t)rb, hi_b = (uint32_t)(rb >> 32);
			const uint32_t lookahead = L - p;
			const int max_len = lookahead < (uint32_t)MAX_MATCH ? (int)lookahead : MAX_MATCH;
			const bool fast = valid && p + 8 <= L;  // max_len >= 8; the compact fields of the record are those of the string
			from_memory = valid && !fast;
			const uint32_t nil_q = nil_candidate(p, lookahead);
			const uint32_t b1 = (lo_i >> 22) & 255u;
			const bool in_run = h == (((b1 << 10) ^ (b1 << 5) ^ b1) & 0x7FFFu) && (((lo_i >> 30) | ((hi_i >> 31) << 2)) == (b1 >> 5));
			if (fast && in_run) {  // the record carries the run-length word in place of bytes 3 and 4
				const uint32_t rw = (hi_i >> 15) & 0xFFFFu;
				run_r = rw & 0x7FFFu;
				if (!((rw >> 15) && (int)run_r >= max_len)) kind = 2;  // else: chain head p-1 at the cap, which the parse knows from the run-length word (checked_match)
			}
			bool act = fast && !in_run;
			// NIL (position 0, and the slide_hash quirk, which sits at distance MAX_DIST exactly) and the window as one bound on the
			// candidate's position: the chain head may be MAX_DIST away, later entries must be closer
			const uint32_t qmin_head = (p > (uint32_t)MAX_DIST ? p - (uint32_t)MAX_DIST : 1u) + (nil_q != 0xFFFFFFFFu ? 1u : 0u);  // (nil_q, when there is one, is p - MAX_DIST)
			const uint32_t qmin_rest = p >= (uint32_t)MAX_DIST ? p - (uint32_t)MAX_DIST + 1u : 1u;
			const uint32_t steps_i = min(i, (uint32_t)LIGHT_STEPS + 1u);  // chain entries that exist at all
			uint32_t clo = lo_i, chi = hi_i;
#pragma unroll
			for (int k = 1; k <= LIGHT_STEPS + 1; k++) {
				// candidate i - k: shift the records up one lane, lane 0 takes index (wave's first) - k from the second register
				clo = wave_shift_up(clo, (uint32_t)__builtin_amdgcn_readlane((int)lo_b, 64 - k));
				chi = wave_shift_up(chi, (uint32_t)__builtin_amdgcn_readlane((int)hi_b, 64 - k));
				if (!__any(act)) break;
				const uint32_t x = hi_i ^ chi;
				const uint32_t q = clo & COMPACT_POS_MASK;
				// the chain ends: before the first record, at another hash, at NIL / outside the window
				if ((uint32_t)k > steps_i || (x & 0x7FFFu) != 0 || q < (k == 1 ? qmin_head : qmin_rest)) act = false;
				if (act) {
					if (k == LIGHT_STEPS + 1) { kind = 1; act = false; }  // a 13th entry: heavy
					else {
						// three bytes in common (less never counts): then x holds the differences of bytes 3 and 4 only
						if (best < max_len && (((lo_i ^ clo) >> 22) | (x >> 31)) == 0) {
							int len;
							if (x & 0x007F8000u) len = 3;
							else if (x) len = 4;
							else if (best < 5 || in[q + best] == in[p + best]) len = common_prefix(in + q, in + p, 5, max_len);
							else len = 0;
							if (len > best) { best = len; best_q = q; found = true; }
						}
						if (best >= max_len) act = false;  // len >= nice_match
					}
				}
			}
		}
		if (from_memory) {
			light_match_from_memory(in, recs, rl, CMP ? COMPACT_POS_MASK : 0xFFFFFFFFu, L, i, p, h, kind, best, best_q, run_r);
			found = kind == 0 && best >= MIN_MATCH;
		}
		if (found && kind == 0) {  // shorter than MIN_MATCH never counts (match_of); queued positions are written by their kernels
			mr[p] = pack_match((uint32_t)best, (uint32_t)best, p - best_q, p - best_q, gen);
		}
		// wave-aggregated appends: one atomic per wave and list
		const uint64_t bh = __ballot(kind == 1), br = __ballot(kind == 2);
		uint32_t base_h = 0, base_r = 0;
		if (lane == 0) {
			if (bh) base_h = atomicAdd(&a.heavy_count[s], (uint32_t)__popcll(bh));
			if (br) base_r = atomicAdd(&a.deep_count[s], (uint32_t)__popcll(br));
		}
		base_h = (uint32_t)__shfl((int)base_h, 0, 64); base_r = (uint32_t)__shfl((int)base_r, 0, 64);
		if (kind == 1) heavy[base_h + (uint32_t)__popcll(bh & lt_mask)] = i;
		if (kind == 2) *(heavy + a.in_stride - 1 - (base_r + (uint32_t)__popcll(br & lt_mask))) = i;  // grows downwards from the end
	}
}

// Wave-cooperative longest_match for one position whose chain is long: 64 candidates per step.
// Returns the packed MatchRec (lo = len4096 | len1024 << 16, hi = dist4096 | dist1024 << 16), wave-uniform.
__device__ __forceinline__ void coop_longest_match(const uint8_t *in, const uint64_t *recs, uint32_t L,
                                                   uint32_t i, uint32_t p, uint32_t pos_mask, int lane, uint32_t &lo, uint32_t &hi)
{
	const uint32_t h = rec_hash(recs[i]);
	const uint32_t lookahead = L - p;
	const int max_len = lookahead < (uint32_t)MAX_MATCH ? (int)lookahead : MAX_MATCH;
	const uint32_t nil_q = nil_candidate(p, lookahead);
	const uint8_t *sp = in + p;
	int best = 0;
	uint32_t best_q = 0;
	int len1024 = -1;
	uint32_t q1024 = 0;
	auto rec_of_round = [&](int r) { const int64_t j = (int64_t)i - 1 - (int64_t)(r * 64 + lane); return recs[j >= 0 ? j : 0]; };
	uint64_t rj_next = rec_of_round(0);
	for (int r = 0; r < 64; r++) {  // 64 x 64 = max_chain_length 4096 candidates
		const int64_t j = (int64_t)i - 1 - (int64_t)(r * 64 + lane);
		const uint64_t rj = j >= 0 ? rj_next : ~0ull;
		rj_next = rec_of_round(r + 1);  // (requested before this round's strings are compared)
		const bool in_chain = j >= 0 && rec_hash(rj) == h;
		const uint32_t q = in_chain ? rec_pos(rj, pos_mask) : 0u;
		const uint32_t dist = p - q;
		const bool term = !in_chain || q == 0 || q == nil_q ||
		                  ((r == 0 && lane == 0) ? dist > (uint32_t)MAX_DIST : dist >= (uint32_t)MAX_DIST);
		const uint64_t tmask = __ballot(term);
		const int nvalid = tmask ? (__ffsll((long long)tmask) - 1) : 64;
		int len = 0;
		if (lane < nvalid && best < max_len) {
			const uint8_t *mp = in + q;
			if (mp[best] == sp[best]) len = common_prefix(mp, sp, 0, max_len);
		}
		// longest length in this step, earliest candidate (lowest lane) that reaches it
		uint64_t cand = __ballot(len > best);
		if (cand) {
			for (int b = 8; b >= 0; b--) {
				const uint64_t mb = __ballot((len >> b) & 1) & cand;
				if (mb) cand = mb;
			}
			const int win = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
			best = __builtin_amdgcn_readlane(len, win);
			best_q = (uint32_t)__builtin_amdgcn_readlane((int)q, win);
		}
		if (r == 15 && nvalid == 64) { len1024 = best; q1024 = best_q; }  // after exactly 1024 candidates
		if (best >= max_len || nvalid < 64) break;
	}
	if (len1024 < 0) { len1024 = best; q1024 = best_q; }
	lo = (uint32_t)best | ((uint32_t)len1024 << 16);
	hi = (best ? p - best_q : 0u) | ((len1024 ? p - q1024 : 0u) << 16);
}

__global__ void __launch_bounds__(256) dfl_match_heavy_kernel(DeflateArgs a, int n)
{
	int s; uint32_t part, nparts;
	if (!xcd_slice(n, s, part, nparts)) return;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const size_t base = (size_t)s * a.in_stride;
	const uint64_t *recs = a.rec_out + base;  // sorted (hash, position) records: position | hash << 32
	uint2 *mr = reinterpret_cast<uint2 *>(a.mr) + base;
	const uint32_t *heavy = a.heavy_list + base;
	const uint32_t nheavy = a.heavy_count[s];
	const int lane = threadIdx.x & 63;
	const uint32_t gen = *a.gen;
	for (uint32_t e = part * (blockDim.x >> 6) + (threadIdx.x >> 6); e < nheavy; e += nparts * (blockDim.x >> 6)) {
		const uint32_t i = heavy[e];
		const uint32_t p = rec_pos(recs[i], a.pos_mask);
		uint32_t lo, hi;
		coop_longest_match(in, recs, L, i, p, a.pos_mask, lane, lo, hi);
		if (lane == 0) mr[p] = pack_match(lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16, gen);
	}
}

// ------------------------------------------------------------------ 2c. positions deep inside a run of one byte
// For p with in[p-3..p+2] all equal to b and r further b's ahead (r < max_len), the chain head is p-1
// and yields exactly r.  Only a candidate q that is r bytes before the END of an earlier run of b,
// followed by the same byte c = in[p+r], can be longer.  So instead of 4096 chain steps the lane scans
// the (much shorter) list of run ends backwards.  Chain-length limits translate into position limits
// through the sorted order: the first K chain entries are the sorted indices i-1 .. i-K.
// Ordered lists of the ends AND starts of runs of >= 3 equal bytes, per slice, from the run-length words: position p puts
// p + 3 on the list of ends when exactly three equal bytes start at p and the fourth byte lies inside the input, and p on the
// list of starts when >= 3 equal bytes start at p and the byte before differs.  dfl_run_len_kernel has counted the entries of
// every chunk of RUNLEN_OUT positions; a workgroup sums the counts of the chunks before its own and writes its entries at that
// offset (a workgroup prefix sum places them, so both lists come out in position order); dfl_run_info_kernel pairs the k-th
// start with the k-th end (run length without scanning).  Until round 3 one workgroup per slice walked its 8192-byte
// chunks in a row, computing the flags byte by byte (0.18 ms per batch).
// Buffer (in_stride words of its own, at most L/4 runs): ends at [0, 1/4), starts at [1/4, 1/2), length|byte<<16 at [1/2, 3/4),
// rank of every eighth position among the ends at [3/4, 1) (in_stride / 8 words).
__global__ void __launch_bounds__(256) dfl_run_lists_kernel(DeflateArgs a)
{
	__shared__ uint32_t wsum_e[4], wsum_s[4], base_es[2];
	const int s = blockIdx.y;
	const uint32_t L = a.in_sizes[s];
	const uint32_t c = blockIdx.x;
	const uint32_t c0 = c * RUNLEN_OUT;
	if (c0 >= L) return;
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const uint16_t *rl = a.run_len + (size_t)s * a.in_stride;
	uint32_t *re = a.run_ends + (size_t)s * a.in_stride;
	uint32_t *rs = re + (a.in_stride >> 2);
	const uint32_t *cnt = a.run_counts + (size_t)s * 2 * a.run_chunks;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t g = c0 + (uint32_t)tid * 8;
	// the lane's eight run-length words: one aligned 16-byte load (g is a multiple of 8; in_stride a multiple of 256)
	uint4 wv = make_uint4(0, 0, 0, 0);
	if (tid * 8 < RUNLEN_OUT && g < L) wv = *reinterpret_cast<const uint4 *>(rl + g);
	const bool first_two_equal = L >= 2 && in[0] == in[1];  // the "same as before" bit of a word is only kept from position 2 on
	if (wave == 0) {  // entries of the chunks before this one
		uint32_t be = 0, bs = 0;
		for (uint32_t q = lane; q < c; q += 64) { be += cnt[2 * q]; bs += cnt[2 * q + 1]; }
		be = wave_sum(be); bs = wave_sum(bs);
		if (lane == 0) { base_es[0] = be; base_es[1] = bs; }
	}
	const uint32_t w32[4] = {wv.x, wv.y, wv.z, wv.w};
	uint32_t me = 0, ms = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const uint32_t w = (w32[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, r = w & 0x7FFFu, p = g + (uint32_t)k;
		const bool pub = (uint32_t)tid * 8 + k < (uint32_t)RUNLEN_OUT && p < L;
		const bool prev_eq = p >= 2 ? (w >> 15) != 0 : (p == 1 && first_two_equal);
		me |= (uint32_t)(pub && r == 3u && p + 3 < L) << k;
		ms |= (uint32_t)(pub && r >= 3u && !prev_eq) << k;
	}
	const uint32_t ne = (uint32_t)__popc(me), ns = (uint32_t)__popc(ms);
	uint32_t inc_e = ne, inc_s = ns;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t oe = __shfl_up(inc_e, d, 64), os = __shfl_up(inc_s, d, 64);
		if (lane >= d) { inc_e += oe; inc_s += os; }
	}
	if (lane == 63) { wsum_e[wave] = inc_e; wsum_s[wave] = inc_s; }
	__syncthreads();
	uint32_t ie = base_es[0] + inc_e - ne, is = base_es[1] + inc_s - ns, tot_e = base_es[0];
	for (int w2 = 0; w2 < 4; w2++) {
		const uint32_t te = wsum_e[w2], ts = wsum_s[w2];
		if (w2 < wave) { ie += te; is += ts; }
		tot_e += te;
	}
	// rank table for the run matcher: ends put on the list by the positions before g (last quarter of the slice's buffer): the
	// "last run end <= p" of a position is then one lookup and a step or two instead of a binary search of twelve dependent loads
	if ((uint32_t)tid * 8 < (uint32_t)RUNLEN_OUT && g < L) re[3 * (a.in_stride >> 2) + (g >> 3)] = ie;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		if ((me >> k) & 1u) re[ie++] = g + k + 3;
		if ((ms >> k) & 1u) rs[is++] = g + k;
	}
	if (tid == 0 && c0 + RUNLEN_OUT >= L) a.run_end_count[s] = tot_e;  // the slice's last chunk
}

__global__ void __launch_bounds__(256) dfl_run_info_kernel(DeflateArgs a)
{
	const int s = blockIdx.y;
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const uint32_t *re = a.run_ends + (size_t)s * a.in_stride;
	const uint32_t *rs = re + (a.in_stride >> 2);
	uint32_t *rl = a.run_ends + (size_t)s * a.in_stride + (a.in_stride >> 1);
	const uint32_t nre = a.run_end_count[s];
	for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nre; t += gridDim.x * blockDim.x) {
		const uint32_t x = re[t], len = x - rs[t];
		rl[t] = (len < 511u ? len : 511u) | ((uint32_t)in[x - 1] << 16);
	}
}

// Positions p whose string starts with three equal bytes b (r = number of b's from p, capped).
// Every candidate of that bucket that can reach length 3 is itself a position of some run of b:
//   * p-1 (when in[p-1] == b) is the chain head and yields exactly r;
//   * a position q of an EARLIER run [s_j, e_j) yields min(e_j - q, r), plus the common prefix of what
//     follows both runs when e_j - q == r; descending q inside a run, the first position reaching the run's
//     maximum is q = e_j - r (run long enough) or the lowest position still inside the chain limits.
// So the lane scans run ends backwards, one O(1) step per run.  The limits of deflate.c become position
// limits: the first K chain entries are the sorted indices i-1 .. i-K (K = 4096, and 1024 for the
// good_match variant); distance < MAX_DIST; position 0 is NIL.
__global__ void __launch_bounds__(256) dfl_match_run_kernel(DeflateArgs a, int n)
{
	int s; uint32_t part, nparts;
	if (!xcd_slice(n, s, part, nparts)) return;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const size_t base = (size_t)s * a.in_stride;
	const uint64_t *recs = a.rec_out + base;  // sorted (hash, position) records: position | hash << 32
	uint2 *mr = reinterpret_cast<uint2 *>(a.mr) + base;
	const uint32_t *deep = a.heavy_list + base + a.in_stride - 1;  // grows downwards from the end
	const uint32_t ndeep = a.deep_count[s];
	const uint32_t *re = a.run_ends + base;
	const uint32_t *rl = re + (a.in_stride >> 1);
	const uint32_t nre = a.run_end_count[s];
	const uint32_t gen = *a.gen;
	for (uint32_t e = part * blockDim.x + threadIdx.x; e < ndeep; e += nparts * blockDim.x) {
		const uint32_t i = *(deep - e);
		// the four records a position can need depend on its sorted index only: requested together (clamped, not tested)
		const uint64_t ri = recs[i];
		const uint64_t r_head = recs[i >= 1 ? i - 1 : 0], r_1024 = recs[i >= 1024 ? i - 1024 : 0], r_4096 = recs[i >= 4096 ? i - 4096 : 0];
		const uint32_t p = rec_pos(ri, a.pos_mask);
		const uint32_t h = rec_hash(ri);
		const uint32_t lookahead = L - p;
		const uint32_t max_len = lookahead < (uint32_t)MAX_MATCH ? lookahead : (uint32_t)MAX_MATCH;
		const uint8_t b = in[p], b_prev = in[p >= 1 ? p - 1 : 0];
		const uint32_t r = a.run_len[base + p] & 0x7FFFu;  // run length from p, capped at max_len
		const bool has_prev = p >= 2 && b_prev == b;  // position 0 is NIL
		uint32_t best4 = has_prev ? r : 0u, q4 = p - 1, best1 = best4, q1 = p - 1;
		bool scan = true;
		if (!has_prev) {  // chain head rules of deflate_slow / longest_match
			const uint64_t rh = i >= 1 ? r_head : ~0ull;
			const bool have_head = i >= 1 && rec_hash(rh) == h;
			const uint32_t hq = have_head ? rec_pos(rh, a.pos_mask) : 0u;
			if (!have_head || hq == 0 || hq == nil_candidate(p, lookahead) || p - hq > (uint32_t)MAX_DIST) scan = false;
			else if (p - hq == (uint32_t)MAX_DIST) {  // only the head itself may sit at distance MAX_DIST
				uint32_t len = 0;
				len = (uint32_t)common_prefix(in + hq, in + p, 0, (int)max_len);
				best4 = best1 = len; q4 = q1 = hq;
				scan = false;
			}
		}
		if (scan && best4 < max_len) {
			const bool ext_ok = r < max_len;  // bytes after the run can only matter below the cap
			const uint8_t c = ext_ok ? in[p + r] : 0;
			const uint32_t qw = p >= (uint32_t)MAX_DIST ? p - (uint32_t)MAX_DIST + 1 : 1u;  // dist < MAX_DIST, q != NIL
			uint32_t qmin4 = qw, qmin1 = qw;
			if (i >= 4096 && rec_hash(r_4096) == h) qmin4 = max(qmin4, rec_pos(r_4096, a.pos_mask));
			if (i >= 1024 && rec_hash(r_1024) == h) qmin1 = max(qmin1, rec_pos(r_1024, a.pos_mask));
			// number of run ends <= p (none lies strictly inside p's own run): an end x comes from position x - 3, so these are the
			// entries of the positions below p - 2 -- the rank table gives those below the 8-aligned part, the rest is a step or two
			uint32_t lo = 0;
			if (p >= 3) {
				lo = re[3 * (a.in_stride >> 2) + ((p - 2) >> 3)];
				while (lo < nre && re[lo] <= p) lo++;
			}
			// (the entry of the next step is requested before this one is evaluated: the loop is a chain of dependent lookups otherwise)
			uint32_t x_n = lo ? re[lo - 1] : 0u, lw_n = lo ? rl[lo - 1] : 0u;
			for (int64_t t = (int64_t)lo - 1; t >= 0; t--) {
				const uint32_t x = x_n;
				const uint32_t lw = lw_n;          // min(run length, 511) | run byte << 16
				{ const int64_t tn = t > 0 ? t - 1 : 0; x_n = re[tn]; lw_n = rl[tn]; }
				if (x < qmin4 + 3) break;          // even q = x-3 is outside the first 4096 entries / the window
				if ((lw >> 16) != b) continue;     // a run of another byte
				const uint32_t m = min(min(lw & 0xFFFFu, r), x - qmin4);  // run length counted down to qmin4, capped at r
				if (m >= 3) {
					uint32_t q, len;
					if (m == r) {
						q = x - r; len = r;
						if (ext_ok && in[x] == c) len = (uint32_t)common_prefix(in + q, in + p, (int)r + 1, (int)max_len);
					} else { q = x - m; len = m; }
					if (len > best4) { best4 = len; q4 = q; }
					// the same run seen through the 1024-entry limit
					if (x >= qmin1 + 3) {
						const uint32_t m1 = min(m, x - qmin1);
						if (m1 == m) { if (len > best1) { best1 = len; q1 = q; } }
						else if (m1 >= 3 && m1 > best1) { best1 = m1; q1 = x - m1; }
					}
					if (best4 >= max_len) break;
				}
				if (x - m <= qmin4) break;  // the run was cut by the limit: older runs are outside
			}
		}
		mr[p] = pack_match(best4, best1, best4 ? p - q4 : 0u, best1 ? p - q1 : 0u, gen);
	}
}

// ------------------------------------------------------------------ 3a. decision records + block summaries
// rec32: bits 0..7 k (deferred literals), 8..16 match length (0 = none), 17..31 distance
__device__ __forceinline__ void match_of(const MatchRec &r, uint32_t p, uint32_t npos, int prev_len, int &len, int &dist)
{
	// deflate_slow: match_length after longest_match + TOO_FAR rule, given prev_length (deflate.c:1863-1880); r = record of p
	len = 2; dist = 0;
	if (p >= npos || prev_len >= MAX_MATCH) return;
	const int l = prev_len >= 32 ? r.len1024 : r.len4096;
	const int d = prev_len >= 32 ? r.dist1024 : r.dist4096;
	if (l > prev_len && l >= MIN_MATCH) { len = l; dist = d; }
	if (len == MIN_MATCH && dist > TOO_FAR) len = 2;
}
__device__ __forceinline__ void match_at(const MatchRec *mr, const uint16_t *rl, uint32_t gen, uint32_t p, uint32_t npos, int prev_len, int &len, int &dist)
{
	len = 2; dist = 0;
	if (p >= npos || prev_len >= MAX_MATCH) return;
	match_of(checked_match(mr[p], gen, rl[p], p, npos + 2), p, npos, prev_len, len, dist);
}

__global__ void __launch_bounds__(256) dfl_rec_kernel(DeflateArgs a)
{
	const int s = blockIdx.y;
	const uint32_t L = a.in_sizes[s];
	const size_t base = (size_t)s * a.in_stride;
	const uint32_t npos = L >= MIN_MATCH ? L - 2 : 0;
	const MatchRec *mr = reinterpret_cast<const MatchRec *>(a.mr) + base;
	const uint16_t *rl = a.run_len + base;
	const uint32_t nblk64 = (L + 63) / 64;
	const int lane = threadIdx.x & 63;
	const uint32_t gen = *a.gen;
	// the lane's own record and its right neighbour's (what the first deferral test reads) are requested together and without a
	// branch (a load inside a conditional is waited for on the spot, which made them two round trips in a row), one turn ahead
	const uint32_t lastp = npos ? npos - 1 : 0;
	const uint32_t wb_step = gridDim.x * (blockDim.x >> 6);
	struct Req { MatchRec m0, m1; uint32_t w0, w1; };
	auto request = [&](uint32_t wb) {
		const uint32_t p = wb * 64 + lane, pa = min(p, lastp), pb = min(p + 1, lastp);
		return Req{mr[pa], mr[pb], rl[pa], rl[pb]};
	};
	const uint32_t wb0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	Req nx = request(wb0);
	for (uint32_t wb = wb0; wb < nblk64; wb += wb_step) {
		const uint32_t p = wb * 64 + lane;
		const uint32_t blk_end = wb * 64 + 64;
		uint32_t rec = 0, nxt = blk_end, cnt = 0;
		const uint32_t pa = min(p, lastp), pb = min(p + 1, lastp);
		const Req cq = nx;
		nx = request(wb + wb_step);
		const MatchRec r0 = checked_match(cq.m0, gen, cq.w0, pa, L), r1 = checked_match(cq.m1, gen, cq.w1, pb, L);
		if (p < L) {
			int len, dist;
			match_of(r0, p, npos, 2, len, dist);
			if (len < MIN_MATCH) { rec = 0; nxt = p + 1; cnt = 1; }  // literal in[p]
			else {
				uint32_t k = 0;
				for (;;) {  // lazy evaluation: defer while the next position has a longer match
					int l2, d2;
					if (k == 0) match_of(r1, p + 1, npos, len, l2, d2);
					else match_at(mr, rl, gen, p + k + 1, npos, len, l2, d2);
					if (l2 > len) { len = l2; dist = d2; k++; } else break;
				}
				rec = k | ((uint32_t)len << 8) | ((uint32_t)dist << 17);
				nxt = p + k + (uint32_t)len;
				cnt = k + 1;
			}
		}
		a.rec32[base + p] = rec;  // in_stride >= L rounded up to 256, so p < stride
		// entry -> exit summary of this 64-position block by pointer doubling
#pragma unroll
		for (int r = 0; r < 6; r++) {
			const bool inside = nxt < blk_end;
			const int j = inside ? (int)(nxt - wb * 64) : lane;
			const uint32_t n2 = __shfl(nxt, j), c2 = __shfl(cnt, j);
			if (inside) { nxt = n2; cnt += c2; }
		}
		a.exit_pos[base + p] = (nxt - blk_end) | (cnt << 16);  // both < 1024: one word per position (entry -> exit offset past the block, symbols)
	}
}

// ------------------------------------------------------------------ 3b. block-to-block walk (one workgroup per slice)
// The parse enters 64-position block b at exit_pos[previous entry]; that hop chain is serial (~4500 hops per
// slice), but its only state is the position, and chains that start at different positions merge as soon as
// they share one decision position.  So the slice is cut into WALK_T segments: every lane hops through its own
// segment from a speculative start (the segment boundary), then restarts from where its predecessor really
// landed until no start moves any more (lane k is final after k passes at the latest; typically 2-3 passes
// of ~18 hops).  A prefix sum over the symbol counts gives every lane its symbol base, and a last pass
// publishes entry position and symbol base of every block the parse really enters.
#ifndef CCT_WALK_T
#define CCT_WALK_T 256
#endif
constexpr int WALK_T = CCT_WALK_T;  // lanes (= segments) per slice; measured per batch: 128: 85, 256: 77, 512: 89, 1024: 163 us (more segments, more restart passes)
__global__ void __launch_bounds__(WALK_T) dfl_walk_kernel(DeflateArgs a, int n)
{
	__shared__ uint32_t s_land[WALK_T];
	__shared__ uint32_t s_wsum[WALK_T / 64];
	const int s = blockIdx.x;
	(void)n;
	const uint32_t L = a.in_sizes[s];
	const size_t base = (size_t)s * a.in_stride;
	const size_t bbase = (size_t)s * (a.in_stride / 64);
	const uint32_t *exit_rec = a.exit_pos + base;  // (offset of the exit past the entry's block) | symbols << 16
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t seg = (((L + WALK_T - 1) / WALK_T) + 63) & ~63u;
	const uint32_t seg_end = min(L, (uint32_t)(tid + 1) * seg);
	uint32_t start = min(L, (uint32_t)tid * seg), land = start, cnt = 0;
	auto hop_through = [&]() {
		uint32_t cur = start, c = 0;
		while (cur < seg_end) { const uint32_t e = exit_rec[cur]; c += e >> 16; cur = (cur & ~63u) + 64u + (e & 0xFFFFu); }
		land = cur; cnt = c;
	};
	hop_through();
	for (;;) {
		__syncthreads();
		s_land[tid] = land;
		__syncthreads();
		const uint32_t pl = tid ? s_land[tid - 1] : 0u;
		const bool moved = tid > 0 && pl != start;
		if (!__syncthreads_or(moved)) break;
		if (moved) { start = pl; hop_through(); }
	}
	// symbols before this lane's segment
	uint32_t inc = cnt;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
	if (lane == 63) s_wsum[wave] = inc;
	__syncthreads();
	uint32_t syms = inc - cnt;
	for (int w = 0; w < wave; w++) syms += s_wsum[w];
	if (tid == WALK_T - 1) a.total_syms[s] = syms + cnt;
	for (uint32_t cur = start; cur < seg_end;) {
		const uint32_t b = cur >> 6;
		a.blk_entry[bbase + b] = cur;
		a.blk_symbase[bbase + b] = syms;
		const uint32_t e = exit_rec[cur];
		syms += e >> 16;
		cur = (cur & ~63u) + 64u + (e & 0xFFFFu);
	}
}

// ------------------------------------------------------------------ 3c. symbols in stream order
// sym32: bits 0..7 lc (literal or length-3), bits 16..31 distance (0 = literal)
__global__ void __launch_bounds__(256) dfl_symbols_kernel(DeflateArgs a)
{
	const int s = blockIdx.y;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	const size_t base = (size_t)s * a.in_stride;
	const size_t bbase = (size_t)s * (a.in_stride / 64);
	const uint32_t nblk64 = (L + 63) / 64;
	const int lane = threadIdx.x & 63;
	uint32_t *sym = a.sym + base;
	uint32_t *bend = a.blk_end + (size_t)s * a.max_blocks;
	// everything a block needs is requested without a branch (a load inside a conditional is waited for on the spot: entry,
	// record, symbol base and the literal byte used to be four round trips in a row) -- and one turn ahead
	const uint32_t wb_step = gridDim.x * (blockDim.x >> 6), wb_last = nblk64 ? nblk64 - 1 : 0;
	struct Req { uint32_t entry, symbase, rec; uint8_t lit0; };
	auto request = [&](uint32_t wb) {
		const uint32_t wbc = min(wb, wb_last), pc = min(wbc * 64 + lane, L - 1);  // L > 0 where the loop runs
		return Req{a.blk_entry[bbase + wbc], a.blk_symbase[bbase + wbc], a.rec32[base + pc], in[pc]};
	};
	const uint32_t wb0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	Req nx = nblk64 ? request(wb0) : Req{0xFFFFFFFFu, 0, 0, 0};
	for (uint32_t wb = wb0; wb < nblk64; wb += wb_step) {
		const uint32_t p = wb * 64 + lane;
		const Req cur_req = nx;
		nx = request(wb + wb_step);
		const uint32_t entry = cur_req.entry, symbase = cur_req.symbase;
		uint32_t rec = cur_req.rec;
		const uint8_t lit0 = cur_req.lit0;
		if (entry == 0xFFFFFFFFu) continue;  // block jumped over by a match
		if (p >= L) rec = 0;
		const uint32_t k = rec & 0xFFu, len = (rec >> 8) & 0x1FFu, dist = rec >> 17;
		const uint32_t nxt = len ? p + k + len : p + 1;
		const uint32_t cnt = len ? k + 1 : 1;
		// which lanes are decision positions: a literal position hands over to its neighbour, so the chain from the
		// entry covers whole stretches of lanes up to the next position that starts a match, and only the matches
		// need a hop (wave-uniform loop on the scalar unit)
		const uint64_t jump = __ballot(len != 0);
		const uint32_t nvalid = min(64u, L - wb * 64);
		uint64_t visited = 0;
		uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(entry - wb * 64));
		while (cur < nvalid) {
			const uint64_t rest = jump >> cur;
			if (!rest) { visited |= ~0ull << cur; break; }
			const int d = __ffsll((long long)rest) - 1;  // lanes cur .. cur + d are visited, the last one jumps
			visited |= (d >= 63 ? ~0ull : ((2ull << d) - 1ull)) << cur;
			cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur + d) - wb * 64;
		}
		if (nvalid < 64) visited &= (1ull << nvalid) - 1ull;
		const bool mine = (visited >> lane) & 1ull;
		// exclusive prefix of symbol counts over the visited lanes
		uint32_t v = mine ? cnt : 0u, inc = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d);
			if (lane >= d) inc += t;
		}
		uint32_t off = symbase + inc - v;
		if (mine) {
			if (p == L - 1) a.postloop_lit[s] = 1;  // the literal tallied after deflate_slow's main loop
			for (uint32_t t = 0; t < (len ? k : 1u); t++, off++) {  // literals
				sym[off] = t ? in[p + t] : lit0;
				if (off % BLOCK_SYMS == BLOCK_SYMS - 1) bend[off / BLOCK_SYMS] = p + t + 1;
			}
			if (len) {
				sym[off] = (len - MIN_MATCH) | (dist << 16);
				if (off % BLOCK_SYMS == BLOCK_SYMS - 1) bend[off / BLOCK_SYMS] = p + k + len;
			}
		}
	}
}

// ------------------------------------------------------------------ 4. Huffman trees per block (trees.c)
// One WAVE per (slice, block).  trees.c's heap decides ties by heap position, so the heap itself is replayed
// by lane 0; everything around it (histogram, leaf list, bit-length statistics, code assignment) is done by
// all 64 lanes, and the tables the serial parts index are staged in LDS (a __constant__ lookup with a
// per-lane index is a global load: ~10x the latency of an LDS read when nothing hides it).
// The working set decides how many blocks a CU works on at once (the serial heap replay is pure LDS latency, so
// resident waves are what hides it): 7.3 KB lets 13 waves share the ~96 KB a CU hands out, which is one wave for every
// block of a 256-slice batch (13 blocks per slice) in a single round.  Hence the aliasing below: the histogram
// counters live in the heap array, the length/distance code tables of the histogram phase in dad[], codes are kept for
// leaves only, the header bits go straight to HBM and the static code lengths are a formula.
struct TreeScratch {  // one block's working set, in LDS
	uint16_t freq[HEAP_SIZE], dad[HEAP_SIZE], len[HEAP_SIZE], code[L_CODES + 2];      // literal/length tree
	uint16_t dfreq[2 * D_CODES + 1], ddad[2 * D_CODES + 1], dlen[2 * D_CODES + 1], dcode[D_CODES + 2];
	uint16_t bfreq[2 * BL_CODES + 1], bdad[2 * BL_CODES + 1], blen[2 * BL_CODES + 1], bcode[BL_CODES + 1];
	// heap entries carry their own sort key: freq << 15 | depth << 10 | node, so that trees.c's smaller(n, m)
	// is (e_n >> 10) <= (e_m >> 10) and one 64-bit LDS read fetches both children (freq <= 16384: 15 bits;
	// depth <= 21 for that total weight: 5 bits; node < 573: 10 bits)
	alignas(16) uint32_t heap[HEAP_SIZE + 3];
	uint32_t bl_count[MAX_BITS + 1];
	uint16_t next_code[MAX_BITS + 1];
	int heap_len, heap_max, max_code, overflow, lmax, dmax;
	uint32_t opt_len, static_len, dyn_body_bits;
	// dynamic-block header (14 + 3*19 + up to 316 * 14 bits), assembled by the whole wave with LDS atomics
	uint32_t hdr_bits[160];
	uint32_t hdr_nbits;
	int ntok, btype, max_blindex;
	// staged tables
	uint8_t extra_l[29], extra_d[30], extra_bl[19], bl_order[19];
	// histogram phase only (see above)
	__device__ __forceinline__ uint32_t *hist_l() { return heap; }
	__device__ __forceinline__ uint32_t *hist_d() { return heap + L_CODES + 2; }
	__device__ __forceinline__ uint8_t *length_code() { return reinterpret_cast<uint8_t *>(dad); }
	__device__ __forceinline__ uint8_t *dist_code() { return reinterpret_cast<uint8_t *>(dad) + 256; }
	// header phase: the run-length tokens of the two code-length sequences (scan_tree records them, the wave emits them);
	// dad[] of the literal/length tree is free by then.  token = symbol 0..18 | extra-bits value << 5
	__device__ __forceinline__ uint16_t *tok() { return dad; }
};
static_assert(sizeof(TreeScratch) <= 8192, "13 tree waves per CU: keep the working set around 8 KB");
static_assert(HEAP_SIZE >= L_CODES + D_CODES, "token list fits dad[]");
static_assert((HEAP_SIZE + 1) * 4 >= (L_CODES + 2 + D_CODES) * 4 && HEAP_SIZE * 2 >= 768, "aliases fit");
// tr_static_init's literal/length code lengths (trees.c:255-258)
__device__ __forceinline__ uint32_t static_llen(int n) { return n <= 143 ? 8u : n <= 255 ? 9u : n <= 279 ? 7u : 8u; }

// LDS-typed pointers: a view chosen at run time (one copy of the tree code serves all three alphabets) still
// compiles to ds_* instructions
typedef __attribute__((address_space(3))) uint16_t lds_u16;
struct TreeView { lds_u16 *freq, *dad, *len, *code; };

// trees.c pqdownheap.  Every level is an LDS round trip on the critical path of the tree kernel, so two levels are
// fetched at once: the pair of children and, below them, the four grandchildren (contiguous, 16-byte aligned).
// The replay runs on one lane, but nothing in it differs between lanes: every loaded value is passed through
// v_readfirstlane so that the compiler keeps the state in SGPRs and branches with s_cbranch instead of juggling the
// exec mask around every `if` (that overhead, not the LDS latency, was most of the cost of a level).
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ void pqdownheap(TreeScratch &S, int k, int hl)
{
	const uint32_t v = uni(S.heap[k]);
	const uint32_t vk = v >> 10;
	int j = k << 1;
	while (j <= hl) {
		uint2 pr = *reinterpret_cast<const uint2 *>(&S.heap[j]);  // j is even: children j, j+1 in one read
		uint4 gc = *reinterpret_cast<const uint4 *>(&S.heap[min(2 * j, (HEAP_SIZE & ~3))]);  // only used when 2j <= hl
		pr.x = uni(pr.x); pr.y = uni(pr.y);
		gc.x = uni(gc.x); gc.y = uni(gc.y); gc.z = uni(gc.z); gc.w = uni(gc.w);
		uint32_t e = pr.x;
		int c = j;
		if (j < hl && (pr.y >> 10) <= (pr.x >> 10)) { c = j + 1; e = pr.y; }
		if (vk <= (e >> 10)) break;
		S.heap[k] = e;
		k = c;
		j = c << 1;
		if (j > hl) break;
		const bool left = (c & 1) == 0;
		const uint32_t x2 = left ? gc.x : gc.z, y2 = left ? gc.y : gc.w;
		e = x2;
		c = j;
		if (j < hl && (y2 >> 10) <= (x2 >> 10)) { c = j + 1; e = y2; }
		if (vk <= (e >> 10)) break;
		S.heap[k] = e;
		k = c;
		j = c << 1;
	}
	S.heap[k] = v;
}

// pqdownheap(1) for a value the caller holds in a register: heap[1] is a hole, v sinks from there.  Returns what ends up
// at heap[1], so that the caller never reads the root back.
// All blocks of a batch replay their heaps at the same time, 13 waves to a CU, and what they compete for is instruction
// issue: a CU has ONE scalar unit, and lane-masked control flow (every `if`, every early exit) is mostly scalar
// instructions.  So this runs a fixed number of levels without a branch: once v has found its place the remaining
// levels still execute, reading harmlessly and storing to the unused heap[0].
__device__ __forceinline__ uint32_t sift_root(TreeScratch &S, uint32_t v, int hl, int levels)
{
	// smaller(a, b) of trees.c is (a >> 10) <= (b >> 10) on these entries, i.e. a <= (b | 1023): one OR instead of two shifts
	if (levels == 0) { S.heap[1] = v; return v; }  // (uniform) a heap of one entry
	// first level, hole at the root: what lands there is the value to return
	const uint2 p0 = *reinterpret_cast<const uint2 *>(&S.heap[2]);
	uint32_t right = (2 < hl && p0.y <= (p0.x | 1023u)) ? 1u : 0u;
	uint32_t e = right ? p0.y : p0.x;
	uint32_t moving = v > (e | 1023u) ? 1u : 0u;  // hl >= 2 here
	const uint32_t rootv = moving ? e : v;
	S.heap[1] = rootv;
	uint32_t k = 2u + right;
	// levels 1 .. levels-2: k lies above the last full level (depth < levels - 1), so both children exist (2k + 1 < 2^levels <= hl)
	// and nothing needs a bound -- a third fewer instructions per level than the general step below, and these are most levels
	for (int l = 1; l < levels - 1; l++) {
		const uint32_t j = k << 1;
		const uint2 pr = *reinterpret_cast<const uint2 *>(&S.heap[j]);
		right = pr.y <= (pr.x | 1023u) ? 1u : 0u;
		e = right ? pr.y : pr.x;
		const uint32_t go = (moving && v > (e | 1023u)) ? 1u : 0u;
		S.heap[moving ? k : 0u] = go ? e : v;  // the hole at k receives the smaller child, or v (which then stays there)
		k = go ? j + right : k;
		moving = go;
	}
	if (levels >= 2) {  // (uniform) the last level: children may lie past the end of the heap
		const uint32_t j = k << 1;
		const uint2 pr = *reinterpret_cast<const uint2 *>(&S.heap[min(j, (uint32_t)(HEAP_SIZE - 1))]);
		right = (j < (uint32_t)hl && pr.y <= (pr.x | 1023u)) ? 1u : 0u;
		e = right ? pr.y : pr.x;
		const uint32_t go = (moving && j <= (uint32_t)hl && v > (e | 1023u)) ? 1u : 0u;
		S.heap[moving ? k : 0u] = go ? e : v;
		k = go ? j + right : k;
		moving = go;
	}
	S.heap[moving ? k : 0u] = v;
	return rootv;
}


// build_tree + gen_bitlen + gen_codes (trees.c:486-700) in five phases.  The heap replay and the overflow repair are
// serial (one lane per block: dfl_tree_kernel runs them for several blocks at once, one block per lane); the phases
// around them use all 64 lanes on one block at a time.
// kind: 0 literal/length, 1 distance, 2 bit-length
__device__ __forceinline__ TreeView view_of(TreeScratch &S, int kind)
{
	if (kind == 0) return TreeView{(lds_u16 *)S.freq, (lds_u16 *)S.dad, (lds_u16 *)S.len, (lds_u16 *)S.code};
	if (kind == 1) return TreeView{(lds_u16 *)S.dfreq, (lds_u16 *)S.ddad, (lds_u16 *)S.dlen, (lds_u16 *)S.dcode};
	return TreeView{(lds_u16 *)S.bfreq, (lds_u16 *)S.bdad, (lds_u16 *)S.blen, (lds_u16 *)S.bcode};
}
__device__ __forceinline__ int elems_of(int kind) { return kind == 0 ? L_CODES : kind == 1 ? D_CODES : BL_CODES; }

// (whole wave) leaves with a non-zero frequency, in symbol order: the initial heap array
__device__ void tree_leaves(TreeScratch &S, int kind)
{
	const TreeView t = view_of(S, kind);
	const int lane = threadIdx.x, elems = elems_of(kind);
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	int max_code = -1;
	uint32_t nleaf = 0;
	for (int n0 = 0; n0 < elems; n0 += 64) {
		const int n = n0 + lane;
		const uint32_t f = n < elems ? t.freq[n] : 0u;
		const uint64_t bal = __ballot(f != 0);
		if (f != 0) S.heap[1 + nleaf + (uint32_t)__popcll(bal & lt_mask)] = (f << 15) | (uint32_t)n;  // depth 0
		else if (n < elems) t.len[n] = 0;
		if (bal) max_code = n0 + 63 - __clzll((long long)bal);
		nleaf += (uint32_t)__popcll(bal);
	}
	if (lane == 0) { S.heap_len = (int)nleaf; S.max_code = max_code; }
}

// (one lane) the heap replay proper (trees.c:625-668)
__device__ void tree_heap(TreeScratch &S, int kind)
{
	const TreeView t = view_of(S, kind);
	int n, m, node, max_code = (int)uni((uint32_t)S.max_code);
	int heap_len = (int)uni((uint32_t)S.heap_len), heap_max = HEAP_SIZE;  // in registers: every LDS access of this loop is on the critical path
	while (heap_len < 2) {
		node = max_code < 2 ? ++max_code : 0;
		t.freq[node] = 1; S.opt_len--;
		S.heap[++heap_len] = (1u << 15) | (uint32_t)node;
		if (kind == 0) S.static_len -= static_llen(node);
		else if (kind == 1) S.static_len -= 5;
	}
	for (n = heap_len / 2; n >= 1; n--) pqdownheap(S, n, heap_len);
	node = elems_of(kind);
	uint32_t top = S.heap[1];
	do {
		const uint32_t en = top;
		const uint32_t lastv = S.heap[heap_len];
		heap_len--;
		const int levels = 31 - __clz(heap_len | 1);  // levels below the root: floor(log2(heap_len))
		const uint32_t em = sift_root(S, lastv, heap_len, levels);
		n = (int)(en & 1023u); m = (int)(em & 1023u);
		S.heap[--heap_max] = en; S.heap[--heap_max] = em;
		const uint32_t f = (en >> 15) + (em >> 15);
		const uint32_t dn = (en >> 10) & 31u, dm = (em >> 10) & 31u;
		const uint32_t d = (dn >= dm ? dn : dm) + 1;
		t.freq[node] = (uint16_t)f;
		t.dad[n] = t.dad[m] = (uint16_t)node;
		top = sift_root(S, (f << 15) | (d << 10) | (uint32_t)node, heap_len, levels);
		node++;
	} while (heap_len >= 2);
	S.heap[--heap_max] = top;
	S.heap_len = heap_len; S.heap_max = heap_max;
	S.max_code = max_code;
}

// (whole wave) gen_bitlen's first loop and its statistics
__device__ void tree_depths(TreeScratch &S, int kind)
{
	const TreeView t = view_of(S, kind);
	const int lane = threadIdx.x;
	const int max_length = kind == 2 ? MAX_BL_BITS : MAX_BITS;
	const int base = kind == 0 ? 257 : 0;
	const int max_code = S.max_code, heap_max = S.heap_max;
	const uint32_t root = S.heap[heap_max];
	// trees.c:506-520: depth of every node = depth of its parent + 1, clamped to max_length; `overflow` counts the nodes
	// that had to be clamped.  Parents precede children in the sorted tail, so trees.c does it in one serial pass; here
	// all nodes of a level are done at once, level after level.
	constexpr uint16_t UNKNOWN = 0xFFFF;
	for (int h = heap_max + 1 + lane; h < HEAP_SIZE; h += 64) t.len[S.heap[h] & 1023u] = UNKNOWN;
	if (lane == 0) t.len[root & 1023u] = 0;
	if (lane <= MAX_BITS) S.bl_count[lane] = 0;
	__syncthreads();
	for (bool again = true; again;) {
		again = false;
		for (int h0 = heap_max + 1; h0 < HEAP_SIZE; h0 += 64) {
			const int h = h0 + lane;
			bool unknown = false;
			if (h < HEAP_SIZE) {
				const int n = (int)(S.heap[h] & 1023u);
				if (t.len[n] == UNKNOWN) {
					const uint32_t ld = t.len[t.dad[n]];
					if (ld == UNKNOWN) unknown = true;
					else t.len[n] = (uint16_t)min(ld + 1u, (uint32_t)max_length);
				}
			}
			if (__ballot(unknown)) again = true;
		}
		__syncthreads();
	}
	int overflow = 0;
	for (int h0 = heap_max + 1; h0 < HEAP_SIZE; h0 += 64) {
		const int h = h0 + lane;
		const bool clamped = h < HEAP_SIZE && t.len[t.dad[S.heap[h] & 1023u]] == (uint32_t)max_length;
		overflow += (int)__popcll(__ballot(clamped));
	}
	// bl_count / opt_len / static_len over the leaves (every leaf of the tree has freq != 0)
	uint32_t o = 0, st = 0;
	for (int n = lane; n <= max_code; n += 64) {
		const uint32_t f = t.freq[n];
		if (f == 0) continue;
		const uint32_t bits = t.len[n];
		atomicAdd(&S.bl_count[bits], 1u);
		uint32_t xbits = 0;
		if (n >= base) xbits = kind == 0 ? S.extra_l[n - base] : kind == 1 ? S.extra_d[n] : S.extra_bl[n];
		o += f * (bits + xbits);
		if (kind == 0) st += f * (static_llen(n) + xbits);
		else if (kind == 1) st += f * (5u + xbits);
	}
	o = wave_sum(o); st = wave_sum(st);
	if (lane == 0) { S.overflow = overflow; S.opt_len += o; S.static_len += st; }
}

// (one lane) gen_bitlen's overflow repair (rare) and the first code of every length
__device__ void tree_fix(TreeScratch &S, int kind)
{
	const TreeView t = view_of(S, kind);
	const int max_length = kind == 2 ? MAX_BL_BITS : MAX_BITS;
	const int max_code = S.max_code;
	int overflow = S.overflow;
	if (overflow > 0) {  // move leaves down until the Kraft sum fits (trees.c:540-571)
		int bits, n, m, h = HEAP_SIZE;
		do {
			bits = max_length - 1;
			while (S.bl_count[bits] == 0) bits--;
			S.bl_count[bits]--; S.bl_count[bits + 1] += 2; S.bl_count[max_length]--;
			overflow -= 2;
		} while (overflow > 0);
		for (bits = max_length; bits != 0; bits--) {
			n = (int)S.bl_count[bits];
			while (n != 0) {
				m = (int)(S.heap[--h] & 1023u);
				if (m > max_code) continue;
				if ((uint32_t)t.len[m] != (uint32_t)bits) {
					S.opt_len += ((uint32_t)bits - t.len[m]) * t.freq[m];
					t.len[m] = (uint16_t)bits;
				}
				n--;
			}
		}
	}
	uint32_t code = 0;
	for (int bits = 1; bits <= MAX_BITS; bits++) { code = (code + S.bl_count[bits - 1]) << 1; S.next_code[bits] = (uint16_t)code; }
}

// (whole wave) gen_codes: symbol n gets next_code[len] + (number of lower symbols with the same length), bit-reversed
__device__ void tree_codes(TreeScratch &S, int kind)
{
	const TreeView t = view_of(S, kind);
	const int lane = threadIdx.x, max_code = S.max_code;
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	uint32_t seen[MAX_BITS + 1];
#pragma unroll
	for (int b = 1; b <= MAX_BITS; b++) seen[b] = S.next_code[b];
	for (int n0 = 0; n0 <= max_code; n0 += 64) {
		const int n = n0 + lane;
		const int len = n <= max_code ? (int)t.len[n] : 0;
		uint32_t mine = 0;
#pragma unroll
		for (int b = 1; b <= MAX_BITS; b++) {
			const uint64_t bal = __ballot(len == b);
			if (len == b) mine = seen[b] + (uint32_t)__popcll(bal & lt_mask);
			seen[b] += (uint32_t)__popcll(bal);
		}
		if (len != 0) t.code[n] = (uint16_t)(__brev(mine) >> (32 - len));
	}
}

// scan_tree + the token side of send_tree (trees.c:703-800), by the whole wave.  trees.c walks the code lengths with a
// small state machine (count, max_count, min_count, prevlen); its state is reset at every change of value, so what it emits
// for a maximal run of R equal lengths v depends on (v, R) only:
//   v != 0:  R < 4: v, R times.   4 <= R <= 6: v, then REP_3_6(R - 1).   R >= 7: v, REP_3_6(6), then REP_3_6(6) for every
//            further six, and for the r = (R - 7) % 6 left over r times v (r < 3) or REP_3_6(r)
//   v == 0:  REPZ_11_138(138) for every 138, and for the r = R % 138 left over r times 0 (r < 3), REPZ_3_10(r) (r <= 10) or
//            REPZ_11_138(r)
// (the first chunk of a non-zero run is cut at 7 = literal + repeat of 6, later ones at 6; zero runs at 138).  So every run
// start computes its tokens on its own; a prefix sum over the runs gives the place in the token list.
// token = symbol 0..18 | extra-bits value << 5; bit-length frequencies are counted in S.hdr_bits[0..18] (free until the
// header is written) and moved to bfreq by the caller.
__device__ void scan_tree_wave(TreeScratch &S, const TreeView &t, int max_code)
{
	const int lane = threadIdx.x;
	constexpr int NCH = (L_CODES + 63) / 64;  // 5 chunks of 64 positions
	const int n_el = max_code + 1;
	uint32_t v[NCH];
	uint64_t startmask[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		const int n = c * 64 + lane;
		v[c] = n < n_el ? (uint32_t)t.len[n] : 0xFFFFu;
		const uint32_t before = n == 0 ? 0xFFFFFFFFu : (uint32_t)t.len[max(n - 1, 0)];
		startmask[c] = __ballot(n < n_el && v[c] != before);
	}
	uint16_t *tk = S.tok();
	uint32_t base = (uint32_t)S.ntok;
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		if (c * 64 >= n_el) break;  // (uniform)
		const int n = c * 64 + lane;
		const bool start = (startmask[c] >> lane) & 1ull;
		// run length: distance to the next run start (or to the end of the sequence)
		uint32_t R = 0;
		if (start) {
			int nxt = n_el;
			const uint64_t rest = lane < 63 ? startmask[c] >> (lane + 1) : 0ull;
			if (rest) nxt = n + (int)__ffsll((long long)rest);
			else {
#pragma unroll
				for (int c2 = NCH - 1; c2 > c; c2--) if (startmask[c2]) nxt = min(nxt, c2 * 64 + (int)__ffsll((long long)startmask[c2]) - 1);
				// (descending loop with min: the nearest following chunk that has a start wins)
			}
			R = (uint32_t)(nxt - n);
		}
		const uint32_t val = v[c];
		// tokens of the run
		uint32_t nlit_head = 0, nrep6 = 0, tail_lit = 0, tail_sym = 0xFFFFFFFFu, nbig = 0;
		if (start) {
			if (val != 0) {
				if (R < 4) nlit_head = R;
				else if (R <= 6) { nlit_head = 1; tail_sym = 16u | ((R - 4u) << 5); }
				else {
					nlit_head = 1; nrep6 = 1 + (R - 7u) / 6u;
					const uint32_t r = (R - 7u) % 6u;
					if (r < 3) tail_lit = r; else tail_sym = 16u | ((r - 3u) << 5);
				}
			} else {
				nbig = R / 138u;
				const uint32_t r = R % 138u;
				if (r < 3) tail_lit = r;
				else if (r <= 10) tail_sym = 17u | ((r - 3u) << 5);
				else tail_sym = 18u | ((r - 11u) << 5);
			}
		}
		const uint32_t ntok = nlit_head + nrep6 + nbig + tail_lit + (tail_sym != 0xFFFFFFFFu ? 1u : 0u);
		uint32_t inc = ntok;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
		if (start) {
			uint32_t w = base + inc - ntok;
			for (uint32_t i = 0; i < nlit_head; i++) tk[w++] = (uint16_t)val;
			for (uint32_t i = 0; i < nrep6; i++) tk[w++] = (uint16_t)(16u | (3u << 5));
			for (uint32_t i = 0; i < nbig; i++) tk[w++] = (uint16_t)(18u | (127u << 5));
			for (uint32_t i = 0; i < tail_lit; i++) tk[w++] = (uint16_t)val;
			if (tail_sym != 0xFFFFFFFFu) tk[w++] = (uint16_t)tail_sym;
			if (nlit_head + tail_lit) atomicAdd(&S.hdr_bits[val], nlit_head + tail_lit);
			if (nrep6) atomicAdd(&S.hdr_bits[16], nrep6);
			if (nbig) atomicAdd(&S.hdr_bits[18], nbig);
			if (tail_sym != 0xFFFFFFFFu) atomicAdd(&S.hdr_bits[tail_sym & 31u], 1u);
		}
		base += (uint32_t)__shfl((int)inc, 63, 64);
	}
	if (lane == 0) S.ntok = (int)base;
}

// `n` bits of `value` at bit position `pos` of the (zeroed) header image
__device__ __forceinline__ void hdr_or(TreeScratch &S, uint32_t pos, uint32_t value, uint32_t n)
{
	const uint32_t w = pos >> 5, sh = pos & 31u;
	atomicOr(&S.hdr_bits[w], value << sh);
	if (sh + n > 32u) atomicOr(&S.hdr_bits[w + 1], value >> (32u - sh));
}

// (whole wave) the header of a dynamic block (trees.c send_all_trees): HLIT, HDIST, HCLEN, the code lengths of the
// bit-length alphabet in bl_order, then the tokens scan_tree recorded, each at its prefix-summed bit offset
__device__ void send_header(TreeScratch &S)
{
	const int lane = threadIdx.x;
	const int nbl = S.max_blindex + 1;
	if (lane == 0) hdr_or(S, 0, (uint32_t)(S.lmax + 1 - 257) | ((uint32_t)(S.dmax + 1 - 1) << 5) | ((uint32_t)(nbl - 4) << 10), 14);
	if (lane < nbl) hdr_or(S, 14u + 3u * (uint32_t)lane, S.blen[S.bl_order[lane]], 3);
	uint32_t base = 14u + 3u * (uint32_t)nbl;
	const uint16_t *tk = S.tok();
	const int nt = S.ntok;
	for (int c0 = 0; c0 < nt; c0 += 64) {
		const int i = c0 + lane;
		uint32_t nbits = 0, val = 0;
		if (i < nt) {
			const uint32_t sym = tk[i] & 31u, ex = tk[i] >> 5;
			const uint32_t cl = S.blen[sym];
			nbits = cl + S.extra_bl[sym];
			val = (uint32_t)S.bcode[sym] | (ex << cl);
		}
		uint32_t inc = nbits;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
		if (nbits) hdr_or(S, base + inc - nbits, val, nbits);
		base += (uint32_t)__shfl((int)inc, 63, 64);
	}
	if (lane == 0) S.hdr_nbits = base;
}

// One wave per TREE_BLOCKS consecutive blocks of a slice: histograms + trees (trees.c _tr_flush_block).  The serial
// parts of a block (heap replay, run-length scan of the code lengths, header bits) keep one lane busy and cost the wave a
// full instruction stream each; with four blocks per wave, lanes 0..3 run them side by side and the instruction count per
// block drops accordingly.  The parallel parts take the blocks in turn.
constexpr int TREE_BLOCKS = 1;
__global__ void __launch_bounds__(64) dfl_tree_kernel(DeflateArgs a)
{
	__shared__ TreeScratch S4[TREE_BLOCKS];
	const int s = blockIdx.y, m0 = blockIdx.x * TREE_BLOCKS, lane = threadIdx.x;
	const uint32_t T = a.total_syms[s];
	const uint32_t L = a.in_sizes[s];
	// number of blocks: one flush per 16383 tallied symbols inside the loop, plus the final flush
	uint32_t nfull = T / BLOCK_SYMS;
	if (T % BLOCK_SYMS == 0 && nfull > 0 && a.postloop_lit[s]) nfull--;  // the post-loop literal never flushes
	const uint32_t nblocks = nfull + 1;
	BlockMeta *meta = a.block_meta + (size_t)s * a.max_blocks;
	if ((uint32_t)m0 >= nblocks) return;
	const int nb = (int)min((uint32_t)TREE_BLOCKS, nblocks - (uint32_t)m0);
	const uint32_t *bend = a.blk_end + (size_t)s * a.max_blocks;
	const bool owner = lane < nb;  // lane q runs the serial parts of block m0 + q
#ifdef CCT_TREE_PROF  // tuning builds only: phase times of one workgroup
	long long tp[16]; int tpi = 0;
#define TREE_STAMP() do { if (tpi < 16) tp[tpi++] = clock64(); } while (0)
#else
#define TREE_STAMP() do {} while (0)
#endif
	TREE_STAMP();

	for (int q = 0; q < nb; q++) {
		TreeScratch &S = S4[q];
		const int m = m0 + q;
		const bool last = (uint32_t)m == nblocks - 1;
		const uint32_t first = (uint32_t)m * BLOCK_SYMS;
		const uint32_t nsym = last ? T - first : (uint32_t)BLOCK_SYMS;
		const uint32_t *sym = a.sym + (size_t)s * a.in_stride + first;
		for (int i = lane; i < HEAP_SIZE; i += 64) { S.freq[i] = 0; S.len[i] = 0; }
		for (int i = lane; i < L_CODES + 2; i += 64) S.code[i] = 0;
		for (int i = lane; i < 2 * D_CODES + 1; i += 64) { S.dfreq[i] = 0; S.dlen[i] = 0; S.ddad[i] = 0; }
		for (int i = lane; i < D_CODES + 2; i += 64) S.dcode[i] = 0;
		for (int i = lane; i < 2 * BL_CODES + 1; i += 64) { S.bfreq[i] = 0; S.blen[i] = 0; S.bdad[i] = 0; }
		for (int i = lane; i < BL_CODES + 1; i += 64) S.bcode[i] = 0;
		uint8_t *length_code = S.length_code(), *dist_code = S.dist_code();
		{  // the constant tables, all loads first: one round trip instead of sixteen in a row (a per-lane index into
			// __constant__ memory is a global load, and a load feeding an LDS store in a rolled loop is waited for at once)
			uint8_t dc[8], lc4[4];
#pragma unroll
			for (int k = 0; k < 8; k++) dc[k] = c_dist_code[lane + 64 * k];
#pragma unroll
			for (int k = 0; k < 4; k++) lc4[k] = c_length_code[lane + 64 * k];
			const int l29 = min(lane, 28), l30 = min(lane, 29), l19 = min(lane, 18);
			const uint8_t xl = c_extra_lbits[l29], xd = c_extra_dbits[l30], xb = c_extra_blbits[l19], bo = c_bl_order[l19];
#pragma unroll
			for (int k = 0; k < 8; k++) dist_code[lane + 64 * k] = dc[k];
#pragma unroll
			for (int k = 0; k < 4; k++) length_code[lane + 64 * k] = lc4[k];
			if (lane < 29) S.extra_l[lane] = xl;
			if (lane < 30) S.extra_d[lane] = xd;
			if (lane < 19) { S.extra_bl[lane] = xb; S.bl_order[lane] = bo; }
		}
		uint32_t *hl = S.hist_l(), *hd = S.hist_d();
		for (int i = lane; i < L_CODES; i += 64) hl[i] = 0;
		if (lane < D_CODES) hd[lane] = 0;
		__syncthreads();
		// sixteen loads in flight (one wave per block, nothing else hides HBM latency), and the next sixteen requested before
		// these are counted: the index is clamped instead of tested so that the loads carry no branch
		{
			const uint32_t lasts = nsym ? nsym - 1 : 0;
			uint32_t nv[16];
#pragma unroll
			for (int k = 0; k < 16; k++) nv[k] = sym[min((uint32_t)(k * 64 + lane), lasts)];
			for (uint32_t i0 = 0; i0 < nsym; i0 += 64 * 16) {
				uint32_t v[16];
#pragma unroll
				for (int k = 0; k < 16; k++) v[k] = nv[k];
#pragma unroll
				for (int k = 0; k < 16; k++) nv[k] = sym[min(i0 + 1024u + (uint32_t)(k * 64 + lane), lasts)];
#pragma unroll
				for (int k = 0; k < 16; k++) {
					if (i0 + (uint32_t)(k * 64 + lane) >= nsym) continue;
					const uint32_t dist = v[k] >> 16, lc = v[k] & 0xFFu;
					if (dist == 0) atomicAdd(&hl[lc], 1u);
					else {
						const uint32_t d1 = dist - 1;
						atomicAdd(&hl[length_code[lc] + 256 + 1], 1u);
						atomicAdd(&hd[d1 < 256 ? dist_code[d1] : dist_code[256 + (d1 >> 7)]], 1u);
					}
				}
			}
		}
		__syncthreads();
		for (int i = lane; i < L_CODES; i += 64) S.freq[i] = (uint16_t)hl[i];
		if (lane < D_CODES) S.dfreq[lane] = (uint16_t)hd[lane];
		for (int i = lane; i < 160; i += 64) S.hdr_bits[i] = 0;
		if (lane == 0) { S.opt_len = 0; S.static_len = 0; S.hdr_nbits = 0; S.ntok = 0; S.btype = 0; S.max_blindex = 0; }
		__syncthreads();
		if (lane == 0) S.freq[END_BLOCK] = 1;
		for (int i = lane; i < HEAP_SIZE; i += 64) S.dad[i] = 0;  // held the code tables of the histogram
	}
	__syncthreads();
	TREE_STAMP();
#pragma unroll 1
	for (int kind = 0; kind < 3; kind++) {  // literal/length, distance, then the bit-length tree over both
		if (kind == 2) {
			if (owner) S4[lane].dyn_body_bits = S4[lane].opt_len;  // code + extra bits of all symbols and END_BLOCK
			__syncthreads();
			for (int q = 0; q < nb; q++) {
				TreeScratch &S = S4[q];
				scan_tree_wave(S, view_of(S, 0), S.lmax);
				__syncthreads();  // (the token count of the first sequence is the base of the second)
				scan_tree_wave(S, view_of(S, 1), S.dmax);
				__syncthreads();
				if (lane < BL_CODES) { S.bfreq[lane] = (uint16_t)S.hdr_bits[lane]; S.hdr_bits[lane] = 0; }
			}
			__syncthreads();
		}
		for (int q = 0; q < nb; q++) tree_leaves(S4[q], kind);
		__syncthreads();
		TREE_STAMP();
		if (owner) tree_heap(S4[lane], kind);
		__syncthreads();
		TREE_STAMP();
		for (int q = 0; q < nb; q++) tree_depths(S4[q], kind);
		__syncthreads();
		if (owner) {
			tree_fix(S4[lane], kind);
			if (kind == 0) S4[lane].lmax = S4[lane].max_code; else if (kind == 1) S4[lane].dmax = S4[lane].max_code;
		}
		__syncthreads();
		for (int q = 0; q < nb; q++) tree_codes(S4[q], kind);
		__syncthreads();
		TREE_STAMP();
	}
	if (owner) {
		TreeScratch &S = S4[lane];
		const int m = m0 + lane;
		const bool last = (uint32_t)m == nblocks - 1;
		const uint32_t first = (uint32_t)m * BLOCK_SYMS;
		const uint32_t nsym = last ? T - first : (uint32_t)BLOCK_SYMS;
		const uint32_t in_begin = m == 0 ? 0u : bend[m - 1];
		const uint32_t in_end = last ? L : bend[m];
		int max_blindex;
		for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
			if (S.blen[S.bl_order[max_blindex]] != 0) break;
		S.opt_len += 3 * ((uint32_t)max_blindex + 1) + 5 + 5 + 4;
		uint32_t opt_lenb = (S.opt_len + 3 + 7) >> 3;
		const uint32_t static_lenb = (S.static_len + 3 + 7) >> 3;
		if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
		const uint32_t stored_len = in_end - in_begin;
		// stored blocks need block_start >= 0 in window coordinates; a block that an incompressible
		// verdict could apply to spans < 32 KiB of input, so its start is always inside the window
		BlockMeta bm;
		bm.in_begin = in_begin; bm.stored_len = stored_len; bm.first_sym = first; bm.nsym = nsym; bm.last = last ? 1u : 0u;
		if (stored_len + 4 <= opt_lenb) {
			bm.type = 0; bm.hdr_nbits = 0; bm.body_bits = 0;
		} else if (static_lenb == opt_lenb) {
			bm.type = 1; bm.hdr_nbits = 0; bm.body_bits = S.static_len;
		} else {
			bm.type = 2; bm.hdr_nbits = 0;  // hdr_nbits follows below, once the wave has written the header
			bm.body_bits = S.dyn_body_bits;
		}
		S.btype = (int)bm.type; S.max_blindex = max_blindex;
		meta[m] = bm;
		if (m == 0) a.n_blocks[s] = nblocks;
	}
	__syncthreads();
	for (int q = 0; q < nb; q++) {
		TreeScratch &S = S4[q];
		if (S.btype != 2) continue;  // (uniform)
		send_header(S);
		__syncthreads();
		BlockTables *bt = a.block_tables + ((size_t)s * a.max_blocks + m0 + q);
		for (int i = lane; i * 32 < (int)S.hdr_nbits; i += 64) bt->hdr_bits[i] = S.hdr_bits[i];
		if (lane == 0) meta[m0 + q].hdr_nbits = S.hdr_nbits;
	}
	__syncthreads();
	TREE_STAMP();
#ifdef CCT_TREE_PROF
	if (s == 0 && m0 == 0 && lane == 0) {
		printf("[tree prof] nb=%d:", nb);
		for (int i = 1; i < tpi; i++) printf(" %lld", tp[i] - tp[i - 1]);
		printf("  (hist | leaves0 heap0 rest0 | scan+leaves1 heap1 rest1 | scan_tree+leaves2 heap2 rest2 | final)\n");
	}
#endif
	// publish the code tables for the emit kernel
	for (int q = 0; q < nb; q++) {
		TreeScratch &S = S4[q];
		BlockTables *bt = a.block_tables + ((size_t)s * a.max_blocks + m0 + q);
		for (int i = lane; i < L_CODES; i += 64) { bt->lcode[i] = S.code[i]; bt->llen[i] = (uint8_t)S.len[i]; }
		if (lane < D_CODES) { bt->dcode[lane] = S.dcode[lane]; bt->dlen[lane] = (uint8_t)S.dlen[lane]; }
	}
}

// ------------------------------------------------------------------ 5. Adler-32 + layout
__global__ void __launch_bounds__(256) dfl_adler_kernel(DeflateArgs a)
{
	__shared__ unsigned long long sa[256], sb[256];
	const int s = blockIdx.x;
	const uint32_t L = a.in_sizes[s];
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	unsigned long long A = 0, B = 0;
	// 16 bytes per step: A += sum d, B += (L - i0) * sum d - sum k * d_k.  Eight steps' loads are in flight together (the rolled
	// loop waited for each one: one workgroup per slice, nothing else hides the latency)
	const uint32_t nfull = L / 16;
	constexpr int U = 8;
	for (uint32_t g0 = threadIdx.x; g0 < nfull; g0 += blockDim.x * U) {
		uint4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const uint4 *>(in + (size_t)min(g0 + (uint32_t)u * blockDim.x, nfull - 1) * 16);
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t g = g0 + (uint32_t)u * blockDim.x;
			if (g >= nfull) continue;
			const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
			uint32_t sd = 0, sk = 0;
#pragma unroll
			for (int k = 0; k < 16; k++) { const uint32_t d = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu; sd += d; sk += (uint32_t)k * d; }
			A += sd;
			B += (unsigned long long)(L - g * 16) * sd - sk;
		}
	}
	if (threadIdx.x == 0)
		for (uint32_t i = nfull * 16; i < L; i++) { const unsigned long long d = in[i]; A += d; B += (unsigned long long)(L - i) * d; }
	sa[threadIdx.x] = A; sb[threadIdx.x] = B;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if ((int)threadIdx.x < st) { sa[threadIdx.x] += sa[threadIdx.x + st]; sb[threadIdx.x] += sb[threadIdx.x + st]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		const uint32_t lo = (uint32_t)((1ull + sa[0]) % 65521ull);
		const uint32_t hi = (uint32_t)(((unsigned long long)L + sb[0]) % 65521ull);
		a.adler[s] = (hi << 16) | lo;
	}
}

// bit offsets of every block, .cct header, zlib header, Adler trailer, final size (one lane per slice)
__global__ void dfl_layout_kernel(DeflateArgs a, int n)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	uint8_t *out = a.out + (size_t)s * a.out_stride;
	for (int i = 0; i < 13; i++) out[i] = a.header13[i];
	out[13] = 0x78; out[14] = 0xDA;  // CMF/FLG for wbits 15, level 9 (deflate.c:819-836)
	uint64_t bit = 8ull * 15;
	BlockMeta *meta = a.block_meta + (size_t)s * a.max_blocks;
	const uint32_t nb = a.n_blocks[s];
	for (uint32_t m = 0; m < nb; m++) {
		BlockMeta bm = meta[m];
		bm.bit_off = bit;
		if (bm.type == 0) {  // _tr_stored_block: 3 header bits, bi_windup, LEN, NLEN, bytes
			bit = (bit + 3 + 7) & ~7ull;
			bit += 32 + 8ull * bm.stored_len;
		} else {
			bit += 3 + bm.hdr_nbits + bm.body_bits;
		}
		if (bm.last) bit = (bit + 7) & ~7ull;  // bi_windup
		meta[m] = bm;
	}
	const uint64_t byte = bit >> 3;
	const uint32_t ad = a.adler[s];
	out[byte + 0] = (uint8_t)(ad >> 24); out[byte + 1] = (uint8_t)(ad >> 16);
	out[byte + 2] = (uint8_t)(ad >> 8);  out[byte + 3] = (uint8_t)ad;
	a.out_sizes[s] = (uint32_t)(byte + 4);
}

// ------------------------------------------------------------------ 6. bit emission
__device__ __forceinline__ void or_bits(uint32_t *words, uint64_t bit, uint64_t value, int nbits)
{
	// value has nbits <= 48 significant bits; LSB-first packing (send_bits, trees.c:187-228)
	if (nbits == 0) return;
	const uint64_t w = bit >> 5;
	const int sh = (int)(bit & 31);
	const uint64_t lo = value << sh;                       // bits 0..63 of value << sh
	const uint64_t hi = sh ? (value >> (64 - sh)) : 0ull;  // bits 64.. (nbits <= 48, sh <= 31)
	atomicOr(&words[w], (uint32_t)lo);
	if (sh + nbits > 32) atomicOr(&words[w + 1], (uint32_t)(lo >> 32));
	if (sh + nbits > 64) atomicOr(&words[w + 2], (uint32_t)hi);
}

__global__ void __launch_bounds__(256) dfl_emit_kernel(DeflateArgs a)
{
	__shared__ uint32_t wsum[4];
	__shared__ unsigned long long s_run;
	const int s = blockIdx.y, m = blockIdx.x;
	if ((uint32_t)m >= a.n_blocks[s]) return;
	const BlockMeta bm = a.block_meta[(size_t)s * a.max_blocks + m];
	const BlockTables *bt = a.block_tables + ((size_t)s * a.max_blocks + m);
	uint8_t *out = a.out + (size_t)s * a.out_stride;
	uint32_t *words = reinterpret_cast<uint32_t *>(out);
	const uint8_t *in = a.in + (size_t)s * a.in_stride;
	uint64_t bit = bm.bit_off;
	if (bm.type == 0) {
		if (threadIdx.x == 0) or_bits(words, bit, (uint64_t)bm.last, 3);
		const uint64_t byte = ((bit + 3 + 7) & ~7ull) >> 3;
		if (threadIdx.x == 0) {
			out[byte] = (uint8_t)(bm.stored_len & 0xFF); out[byte + 1] = (uint8_t)(bm.stored_len >> 8);
			out[byte + 2] = (uint8_t)(~bm.stored_len & 0xFF); out[byte + 3] = (uint8_t)((~bm.stored_len >> 8) & 0xFF);
		}
		for (uint32_t i = threadIdx.x; i < bm.stored_len; i += blockDim.x) out[byte + 4 + i] = in[bm.in_begin + i];
		return;
	}
	if (threadIdx.x == 0) or_bits(words, bit, (uint64_t)((bm.type << 1) + bm.last), 3);
	bit += 3;
	if (bm.type == 2) {
		for (uint32_t i = threadIdx.x; i * 32 < bm.hdr_nbits; i += blockDim.x) {
			const int nb = (int)min(32u, bm.hdr_nbits - i * 32);
			or_bits(words, bit + 32ull * i, bt->hdr_bits[i], nb);
		}
		bit += bm.hdr_nbits;
	}
	const uint32_t *sym = a.sym + (size_t)s * a.in_stride + bm.first_sym;
	const bool dyn = bm.type == 2;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// code tables of this block in LDS; code bits of 256 symbols are merged into LDS words first, so the
	// output sees one atomic per 32-bit word instead of two or three per symbol
	__shared__ uint16_t t_lcode[L_CODES + 2], t_dcode[D_CODES + 2];
	__shared__ uint8_t t_llen[L_CODES + 2], t_dlen[D_CODES + 2];
	__shared__ uint32_t wbuf[400];  // 256 symbols x <= 48 bits = 384 words, + alignment word
	// the symbol -> code tables of trees.c, staged in LDS: a __constant__ lookup with a per-lane index is a global
	// load, and every iteration with a match in it would wait for four of them in a row
	__shared__ uint8_t t_length_code[256], t_dist_code[512], t_extra_l[32], t_extra_d[32];
	__shared__ uint16_t t_base_length[32], t_base_dist[32];
	{  // every table entry this lane stages is requested before the first is stored (clamped indices, no branches around the
		// loads): one memory round trip instead of a dozen in a row.  blockDim.x == 256
		const int t = threadIdx.x, t2 = min(t + 256, L_CODES - 1), td = min(t, D_CODES - 1), t29 = min(t, 28), t30 = min(t, 29);
		const uint16_t *lsrc = dyn ? bt->lcode : c_static_lcode, *dsrc = dyn ? bt->dcode : c_static_dcode;
		const uint8_t *llsrc = dyn ? bt->llen : c_static_llen;
		const uint16_t lc0 = lsrc[t], lc1 = lsrc[t2];
		const uint8_t ll0 = llsrc[t], ll1 = llsrc[t2];
		const uint16_t dc = dsrc[td];
		uint8_t dl = bt->dlen[td];
		if (!dyn) dl = 5;
		const uint8_t k0 = c_dist_code[t], k1 = c_dist_code[t + 256], lcd = c_length_code[t];
		const uint8_t xl = c_extra_lbits[t29], xd = c_extra_dbits[t30];
		const uint16_t bl = c_base_length[t29], bd = c_base_dist[t30];
		t_lcode[t] = lc0; t_llen[t] = ll0;
		if (t + 256 < L_CODES) { t_lcode[t + 256] = lc1; t_llen[t + 256] = ll1; }
		if (t < D_CODES) { t_dcode[t] = dc; t_dlen[t] = dl; }
		t_dist_code[t] = k0; t_dist_code[t + 256] = k1; t_length_code[t] = lcd;
		if (t < 29) { t_extra_l[t] = xl; t_base_length[t] = bl; }
		if (t < 30) { t_extra_d[t] = xd; t_base_dist[t] = bd; }
	}
	if (threadIdx.x == 0) s_run = bit;
	for (int i = threadIdx.x; i < 400; i += blockDim.x) wbuf[i] = 0;
	__syncthreads();
	// the symbols of the next 256 are requested before these are coded (index clamped, not tested: a load inside a
	// conditional is waited for on the spot)
	const uint32_t last_sym = bm.nsym ? bm.nsym - 1 : 0;
	uint32_t vnext = sym[min((uint32_t)threadIdx.x, last_sym)];
	for (uint32_t base = 0; base <= bm.nsym; base += blockDim.x) {  // one extra slot for END_BLOCK
		const uint32_t i = base + threadIdx.x;
		uint64_t bits = 0;
		int nb = 0;
		const uint32_t v = vnext;
		vnext = sym[min(i + blockDim.x, last_sym)];
		if (i < bm.nsym) {
			uint32_t dist = v >> 16;
			const uint32_t lc = v & 0xFFu;
			if (dist == 0) {
				bits = t_lcode[lc];
				nb = t_llen[lc];
			} else {  // compress_block, trees.c:1070-1110
				int code = t_length_code[lc];
				const int lsym = code + 256 + 1;
				bits = t_lcode[lsym];
				nb = t_llen[lsym];
				int extra = t_extra_l[code];
				if (extra) { bits |= (uint64_t)(lc - t_base_length[code]) << nb; nb += extra; }
				dist--;
				code = dist < 256 ? t_dist_code[dist] : t_dist_code[256 + (dist >> 7)];
				bits |= (uint64_t)t_dcode[code] << nb;
				nb += t_dlen[code];
				extra = t_extra_d[code];
				if (extra) { bits |= (uint64_t)(dist - t_base_dist[code]) << nb; nb += extra; }
			}
		} else if (i == bm.nsym) {
			bits = t_lcode[END_BLOCK];
			nb = t_llen[END_BLOCK];
		}
		// exclusive scan of bit counts over the workgroup
		uint32_t inc = (uint32_t)nb;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d);
			if (lane >= d) inc += t;
		}
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t wb = 0, tot = 0;
		for (int w = 0; w < 4; w++) { if (w < wave) wb += wsum[w]; tot += wsum[w]; }
		const uint64_t run = s_run;
		const uint32_t lbit = (uint32_t)(run & 31) + wb + inc - (uint32_t)nb;  // bit position inside wbuf
		if (nb) {
			const uint32_t w0 = lbit >> 5, sh = lbit & 31;
			const uint64_t lo = bits << sh;
			atomicOr(&wbuf[w0], (uint32_t)lo);
			if (sh + nb > 32) atomicOr(&wbuf[w0 + 1], (uint32_t)(lo >> 32));
			if (sh + nb > 64) atomicOr(&wbuf[w0 + 2], (uint32_t)(bits >> (64 - sh)));
		}
		__syncthreads();
		const uint32_t nwords = ((uint32_t)(run & 31) + tot + 31) >> 5;
		const uint64_t gw = run >> 5;
		for (uint32_t k = threadIdx.x; k < nwords; k += blockDim.x) {
			const uint32_t v = wbuf[k];
			if (v) atomicOr(&words[gw + k], v);  // edge words are shared with the neighbouring 256 symbols / blocks
			wbuf[k] = 0;
		}
		if (threadIdx.x == 0) s_run = run + tot;
		__syncthreads();
	}
}

// files back to back (exclusive scan of sizes by one workgroup, then a grid copy) for ONE device->host copy
__global__ void __launch_bounds__(256) dfl_pack_offsets_kernel(const uint32_t *sizes, int n, uint64_t *offsets, int exact)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		uint64_t acc = 0;
		for (int i = 0; i < n; i++) { offsets[i] = acc; acc += exact ? sizes[i] : ((sizes[i] + 15u) & ~15u); }
		offsets[n] = acc;
	}
}
__global__ void __launch_bounds__(256) dfl_pack_kernel(const uint8_t *src, size_t stride, const uint32_t *sizes,
                                                       const uint64_t *offsets, uint8_t *dst)
{
	const int s = blockIdx.y;
	const uint32_t n16 = (sizes[s] + 15u) >> 4;
	const uint4 *in = reinterpret_cast<const uint4 *>(src + (size_t)s * stride);
	uint4 *out = reinterpret_cast<uint4 *>(dst + offsets[s]);
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) out[i] = in[i];
}

__global__ void dfl_offsets2_kernel(DeflateArgs a, int n)
{
	// blk_entry := "not entered" for the walk kernel
	const size_t per = a.in_stride / 64;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * per; i += (size_t)gridDim.x * blockDim.x)
		a.blk_entry[i] = 0xFFFFFFFFu;
}

__global__ void dfl_offsets_kernel(DeflateArgs a, int n)
{
	if (blockIdx.x == 0) {
		for (int s = threadIdx.x; s < n; s += blockDim.x) { a.postloop_lit[s] = 0; a.heavy_count[s] = 0; a.deep_count[s] = 0; a.run_end_count[s] = 0; }  // (run_end_count: an empty slice has no chunk that would write it)
		if (threadIdx.x == 0) *a.gen = *a.gen % GEN_MAX + 1u;  // tag of this pass's match records (see MatchRec)
	}
	// run-list counters of dfl_run_len_kernel (a memset node of a few megabytes at the head of the graph waited for the copy
	// engine: with the files of the pass before still on the wire a pass of 1024^2 slices started 3 ms late)
	const size_t nrc = (size_t)n * 2 * (size_t)a.run_chunks;
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nrc; t += (size_t)gridDim.x * blockDim.x) a.run_counts[t] = 0;
}

}  // namespace

// host: trees.c tr_static_init tables -> constant memory
hipError_t deflate_init_tables()
{
	uint8_t length_code[256]; uint16_t base_length[29]; uint8_t dist_code[512]; uint16_t base_dist[30];
	uint16_t static_lcode[288]; uint8_t static_llen[288]; uint16_t static_dcode[30];
	static const int xl[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
	static const int xd[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
	int length = 0, code, n, dist = 0;
	for (code = 0; code < 28; code++) {
		base_length[code] = (uint16_t)length;
		for (n = 0; n < (1 << xl[code]); n++) length_code[length++] = (uint8_t)code;
	}
	length_code[length - 1] = (uint8_t)code;
	base_length[28] = 0;
	for (code = 0; code < 16; code++) {
		base_dist[code] = (uint16_t)dist;
		for (n = 0; n < (1 << xd[code]); n++) dist_code[dist++] = (uint8_t)code;
	}
	dist >>= 7;
	for (; code < 30; code++) {
		base_dist[code] = (uint16_t)(dist << 7);
		for (n = 0; n < (1 << (xd[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
	}
	uint16_t bl_count[16] = {0}, next_code[16];
	n = 0;
	while (n <= 143) { static_llen[n++] = 8; bl_count[8]++; }
	while (n <= 255) { static_llen[n++] = 9; bl_count[9]++; }
	while (n <= 279) { static_llen[n++] = 7; bl_count[7]++; }
	while (n <= 287) { static_llen[n++] = 8; bl_count[8]++; }
	unsigned c = 0;
	for (int bits = 1; bits <= 15; bits++) { c = (c + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)c; }
	auto rev = [](unsigned v, int len) { unsigned r = 0; do { r |= v & 1; v >>= 1; r <<= 1; } while (--len > 0); return r >> 1; };
	for (n = 0; n < 288; n++) static_lcode[n] = (uint16_t)rev(next_code[static_llen[n]]++, static_llen[n]);
	for (n = 0; n < 30; n++) static_dcode[n] = (uint16_t)rev((unsigned)n, 5);
	hipError_t e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_length_code), length_code, sizeof length_code)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_base_length), base_length, sizeof base_length)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_dist_code), dist_code, sizeof dist_code)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_base_dist), base_dist, sizeof base_dist)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_static_lcode), static_lcode, sizeof static_lcode)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_static_llen), static_llen, sizeof static_llen)) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_static_dcode), static_dcode, sizeof static_dcode)) != hipSuccess) return e;
	return hipSuccess;
}

size_t deflate_sort_temp_bytes(size_t total, int n)
{
	(void)total; (void)n;
	return 0;  // the sort keeps its state in LDS (dfl_sort_pass_kernel)
}

// One DEFLATE pass.  `side` + `ev` (four events without timing), when given, let kernels that do not depend on each other share
// the chip: the run lists are written while the match kernel runs (its waves fill the chip, so the side kernels mostly end
// with it -- but they no longer come after it), and the two tail-bound matchers (long chains: a few waves walking thousands
// of entries; runs: a binary search + scan per position) run side by side, with the Adler-32 sums behind the shorter one --
//     main:  run lengths, sort A, sort B ─┬─ match ─────┬─ run matcher ──────────┬─ decisions, walk, symbols, trees, emit
//     side:                               └─ run lists ─┴─ heavy matcher, Adler ─┘
// Under stream capture this becomes the same fork / join in the graph.  Next to the SORT passes nothing may run: pass B took
// 0.60 instead of 0.34 ms beside the run-list kernels (profiles/r03_deflate_fork.log).
hipError_t launch_deflate(const DeflateArgs &a, int n, void *sort_temp, size_t sort_temp_bytes, hipStream_t st, hipStream_t side,
                          const hipEvent_t *ev)
{
	hipError_t e;
	if ((e = hipMemsetAsync(a.out, 0, (size_t)n * a.out_stride, st)) != hipSuccess) return e;
	if ((e = hipMemsetAsync(a.sort_hist, 0, (size_t)n * 384 * 4, st)) != hipSuccess) return e;
	hipLaunchKernelGGL(dfl_offsets_kernel, dim3(64), dim3(256), 0, st, a, n);  // (also zeroes run_counts)
	const int gx = (int)std::min<size_t>(64, (a.in_stride + 255) / 256);
	(void)sort_temp; (void)sort_temp_bytes;
	hipLaunchKernelGGL(dfl_run_len_kernel, dim3(gx, n), dim3(256), 0, st, a);       // run-length words, sort histograms, run-list counts
	hipLaunchKernelGGL(dfl_sort_pass_kernel<true>, dim3(n), dim3(1024), 0, st, a);   // in -> rec_in by hash & 255
	hipLaunchKernelGGL(dfl_sort_pass_kernel<false>, dim3(n), dim3(1024), 0, st, a);  // -> rec_out by hash >> 8
	const bool fork = side != nullptr && ev != nullptr;
	hipStream_t s2 = fork ? side : st;
	if (fork) {
		if ((e = hipEventRecord(ev[0], st)) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(s2, ev[0], 0)) != hipSuccess) return e;
	}
	hipLaunchKernelGGL(dfl_run_lists_kernel, dim3(a.run_chunks, n), dim3(256), 0, s2, a);
	hipLaunchKernelGGL(dfl_run_info_kernel, dim3(8, n), dim3(256), 0, s2, a);
	// wide records: one 256-lane block per 256 positions (more of them in flight hide the scattered accesses better than grid-stride
	// loops); compact records: the kernel is bound by its instructions, and 256 blocks per slice with four turns each measured best
	// (486 us against 499 / 510 with 128 / 512, profiles/r03_match_grid.log)
	const int gm_cap = a.pos_mask == COMPACT_POS_MASK ? 256 : 2048;
	const int gm = (int)std::min<size_t>(gm_cap, (a.in_stride + 255) / 256), n8 = (n + 7) & ~7;  // see xcd_slice()
	if (a.pos_mask == COMPACT_POS_MASK) hipLaunchKernelGGL(dfl_match_kernel<true>, dim3(gm, n8), dim3(256), 0, st, a, n);
	else if (a.pos_mask == 0xFFFFFFFFu) hipLaunchKernelGGL(dfl_match_kernel<false>, dim3(gm, n8), dim3(256), 0, st, a, n);
	else return hipErrorInvalidValue;
	if (fork) {  // both have what the other produced: the run matcher the run lists, the heavy matcher the queue of long chains
		if ((e = hipEventRecord(ev[1], st)) != hipSuccess) return e;
		if ((e = hipEventRecord(ev[2], s2)) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(s2, ev[1], 0)) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(st, ev[2], 0)) != hipSuccess) return e;
	}
	hipLaunchKernelGGL(dfl_match_heavy_kernel, dim3(gx, n8), dim3(256), 0, s2, a, n);
	hipLaunchKernelGGL(dfl_adler_kernel, dim3(n), dim3(256), 0, s2, a);
	hipLaunchKernelGGL(dfl_match_run_kernel, dim3(gx, n8), dim3(256), 0, st, a, n);
	if (fork) {
		if ((e = hipEventRecord(ev[3], s2)) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(st, ev[3], 0)) != hipSuccess) return e;
	}
	hipLaunchKernelGGL(dfl_rec_kernel, dim3(gx, n), dim3(256), 0, st, a);
	hipLaunchKernelGGL(dfl_offsets2_kernel, dim3(256), dim3(256), 0, st, a, n);
	hipLaunchKernelGGL(dfl_walk_kernel, dim3(n), dim3(WALK_T), 0, st, a, n);
	hipLaunchKernelGGL(dfl_symbols_kernel, dim3(gx, n), dim3(256), 0, st, a);
	hipLaunchKernelGGL(dfl_tree_kernel, dim3((a.max_blocks + TREE_BLOCKS - 1) / TREE_BLOCKS, n), dim3(64), 0, st, a);
	hipLaunchKernelGGL(dfl_layout_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a, n);
	hipLaunchKernelGGL(dfl_emit_kernel, dim3(a.max_blocks, n), dim3(256), 0, st, a);
	return hipGetLastError();
}

// exact = 1: files touch each other (archive layout); bytes are moved one at a time since the
// destinations are unaligned
__global__ void __launch_bounds__(256) dfl_pack_exact_kernel(const uint8_t *src, size_t stride, const uint32_t *sizes,
                                                             const uint64_t *offsets, uint8_t *dst)
{
	const int s = blockIdx.y;
	const uint32_t nbytes = sizes[s];
	const uint8_t *in = src + (size_t)s * stride;
	uint8_t *out = dst + offsets[s];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += gridDim.x * blockDim.x) out[i] = in[i];
}

hipError_t launch_pack(const uint8_t *src, size_t stride, const uint32_t *sizes, int n, uint64_t *offsets, uint8_t *dst,
                       int exact, hipStream_t st)
{
	hipLaunchKernelGGL(dfl_pack_offsets_kernel, dim3(1), dim3(64), 0, st, sizes, n, offsets, exact);
	if (exact) hipLaunchKernelGGL(dfl_pack_exact_kernel, dim3(32, n), dim3(256), 0, st, src, stride, sizes, offsets, dst);
	else hipLaunchKernelGGL(dfl_pack_kernel, dim3(16, n), dim3(256), 0, st, src, stride, sizes, offsets, dst);
	return hipGetLastError();
}

}  // namespace cct
