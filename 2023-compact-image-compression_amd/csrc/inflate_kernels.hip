// Device INFLATE (RFC 1950/1951) for the decode side: zlib.decompress(file_bytes[13:]) of the reference
// (src/codec/core.py:421).  zlib is a third-party dependency of the reference; inflating is fully
// specified by the stream format, so any correct decoder returns the same bytes (and the Adler-32
// trailer is verified like zlib does).
//
// One workgroup of 8 waves per stream.  The format carries no symbol index, so the position of symbol k+1 is
// only known after symbol k has been decoded; a single lane walking that chain spends ~10^3 cycles per symbol.
// The workgroup breaks the chain speculatively (Huffman codes re-synchronise after a few symbols):
//   * a round covers NT (512) segments of SEG_BITS (256) compressed bits, one per lane; lane 0 starts at the true position,
//     the others at their segment boundary, and every lane decodes until it crosses into the next segment;
//   * lanes then restart from where their predecessor really landed until no start moves any more (lane k is
//     final after k passes at the latest; in practice 2-3 passes), which yields the true chain of the round;
//   * a prefix sum over output bytes gives every lane its output offset; lanes decode once more, writing
//     literals to the output ring and LZ77 copies to a list;
//   * block headers are parsed redundantly by every lane (workgroup-uniform control flow); the decode tables of each
//     block are filled by gather, one lane per entry, from the canonical-code intervals (see "canonical Huffman codes"
//     below): an 11-bit root for literal/length whose entries hold TWO literals when both codes fit, a 10-bit root for
//     distances, no sub-tables (a longer code is decoded from the intervals directly);
//   * a wave pays for every branch one of its lanes takes, so the symbol loop runs a few literal-only steps per general
//     step (walk_segment);
//   * the copies of a round are resolved a batch at a time, one lane per copy, in as many passes as the longest
//     dependency chain of the batch; chains of run copies collapse into one pattern fill;
//   * compressed input is staged through LDS in 4 KiB chunks, output through a 64 KiB LDS ring that
//     is flushed to HBM as aligned 16-byte stores and Adler-summed on the way out.
// Slices are independent, so 256 streams occupy 256 CUs at once.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

// Lanes per stream: 512 when the decode has the device to itself (two waves per SIMD: the walk of one hides the LDS round trips
// of the other; 1.66 -> 1.38 ms per 256 streams), 256 next to an encode batch (measured in the bench: with 512 lanes and 139 KB of
// LDS per stream INFLATE ran 1.97 ms against 1.77 ms and the DEFLATE pass beside it 5.3 against 5.1 ms).  launch_inflate() picks.
template <int LANES>
struct Geo {
	static constexpr int NT = LANES;                                  // lanes per stream
	static constexpr int NWV = LANES / 64;                            // waves per stream
	static constexpr int INF_IN = LANES >= 512 ? 32768 : 16384;       // staged input window (a round reads NT * SEG_BITS / 8 bytes + a chunk of slack)
	static constexpr int ROUND_OUT_BUDGET = LANES >= 512 ? 28672 : 24576;  // output bytes of a round (ring = 32 KiB window + round)
};
#define GEO_CONSTANTS constexpr int NT = G::NT, NWV = G::NWV, INF_IN = G::INF_IN, ROUND_OUT_BUDGET = G::ROUND_OUT_BUDGET; (void)NT; (void)NWV; (void)INF_IN; (void)ROUND_OUT_BUDGET
constexpr int INF_RING = 65536, INF_RMASK = INF_RING - 1, INF_FLUSH = 4096;
constexpr int INF_CHUNK = 4096;
#ifndef CCT_INF_LL_BITS
#define CCT_INF_LL_BITS 12
#endif
constexpr int LL_BITS = CCT_INF_LL_BITS;
constexpr int D_BITS = 10;
constexpr uint32_t K_LIT = 1u << 24, K_LEN = 2u << 24, K_EOB = 3u << 24, K_TWO = 1u << 26;
#ifndef CCT_INF_SEG_BITS
#define CCT_INF_SEG_BITS 256
#endif
constexpr int SEG_BITS = CCT_INF_SEG_BITS;    // compressed bits per lane and round
constexpr int LANE_OUT_CAP = 16384;           // a lane stops (and ends the round) once it has produced this much
constexpr int MLIST_CAP = 2048;               // LZ77 copies a round may hold (one lane alone: at most SEG_BITS / 2)
constexpr uint32_t SEG_EOB = 1, SEG_BAD = 2, SEG_CUT = 4;
template <class G>
constexpr bool geo_ok()
{
	return 32768 + G::ROUND_OUT_BUDGET <= INF_RING          // the ring holds the 32 KiB window and the output of a round
	       && LANE_OUT_CAP + 512 <= G::ROUND_OUT_BUDGET        // lane 0 alone always fits a round
	       && G::NT * SEG_BITS / 8 + 32 + INF_CHUNK <= G::INF_IN  // the staged window covers a round's input
	       && G::NWV >= 4;                                     // build_tables: a wave takes at most two chunks of symbols
}
static_assert(geo_ok<Geo<256>>() && geo_ok<Geo<512>>(), "INFLATE geometry");

__constant__ uint16_t c_lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t c_lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint16_t c_dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__constant__ uint8_t c_dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t c_clorder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

constexpr int CL_BITS = 7;  // code-length codes are at most 7 bits long: one flat table

// canonical code description, see "canonical Huffman codes" below
struct CanonLds {
	uint32_t lim[16], adj[16], offs[16];
};

template <class G>
struct InfShared {
	static constexpr int NT = G::NT, NWV = G::NWV, INF_IN = G::INF_IN;
	uint8_t ring[INF_RING];
	alignas(16) uint8_t inbuf[INF_IN];
	// root decode tables (LL_BITS / D_BITS); entry layout at ll_entry() below; 0 = no code of root length or less
	uint32_t ll_tab[1 << LL_BITS], d_tab[1 << D_BITS];
	CanonLds ll_canon, d_canon;                 // ll_canon doubles as the code-length code's while a header is parsed
	uint32_t ll_sent[288], d_sent[32];          // entry of every coded symbol by (length, symbol)
	uint8_t chunkcnt[5 * 16];                   // table build: codes of every length in each chunk of 64 symbols
	uint32_t cnt[16];
	uint8_t cl_tab[1 << CL_BITS];               // symbol << 3 | code length, 0 = no such code
	uint8_t lens[320];
	uint2 mlist[MLIST_CAP];                     // x = destination, y = length | (distance - 1) << 9
	uint16_t lbase[29], dbase[30];
	uint8_t lext[29], dext[30];
	// round state
	uint32_t land[NT];                          // landing position of every lane, bits from the round's origin
	uint32_t wsum_b[NWV], wsum_m[NWV], wred[2 * NWV];
	uint32_t rres[4];                           // result of the round: bytes, copies, landing position, flags of its last lane
	int ok;
};

struct BitReader {    // workgroup-uniform reader used for headers; every lane holds the same state
	const uint8_t *src;   // 16-byte aligned base of the staged stream
	uint64_t avail;       // bytes of the whole input buffer readable from src (zeros are fed beyond)
	uint64_t nbytes;      // bytes from src to the end of THIS stream (consuming more = truncated stream)
	uint64_t bytepos;     // next byte to move into the bit buffer (relative to src), multiple of 4
	uint64_t staged_end;  // inbuf holds [staged_end - INF_IN, staged_end), staged_end % INF_CHUNK == 0
	uint64_t buf;
	int cnt;
};

template <class G>
__device__ __forceinline__ void stage_chunk(InfShared<G> &S, BitReader &br)
{
	GEO_CONSTANTS;
	__syncthreads();  // nobody still reads the chunk that is about to be replaced
	uint4 *dst = reinterpret_cast<uint4 *>(S.inbuf + (br.staged_end & (INF_IN - 1)));
	const uint4 *srcv = reinterpret_cast<const uint4 *>(br.src + br.staged_end);
	for (int t = threadIdx.x; t < INF_CHUNK / 16; t += NT) {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (br.staged_end + (uint64_t)t * 16 + 16 <= br.avail) v = srcv[t];
		dst[t] = v;
	}
	br.staged_end += INF_CHUNK;
	__syncthreads();
}

template <class G>
__device__ __forceinline__ void refill(InfShared<G> &S, BitReader &br)
{
	GEO_CONSTANTS;
	while (br.cnt <= 32) {
		while (br.bytepos + 4 > br.staged_end) stage_chunk(S, br);
		const uint32_t w = *reinterpret_cast<const uint32_t *>(S.inbuf + (br.bytepos & (INF_IN - 1)));
		br.buf |= (uint64_t)w << br.cnt;
		br.cnt += 32;
		br.bytepos += 4;
	}
}
// bytes of the stream consumed so far (whole bytes that contain at least one consumed bit)
__device__ __forceinline__ uint64_t consumed_bytes(const BitReader &br) { return br.bytepos - (uint64_t)(br.cnt >> 3); }

__device__ __forceinline__ uint32_t getbits(BitReader &br, int n)
{
	const uint32_t v = (uint32_t)(br.buf & ((1ull << n) - 1ull));
	br.buf >>= n;
	br.cnt -= n;
	return v;
}

// ---- canonical Huffman codes (RFC 1951 3.2.2) as gather tables ---------------------------------------------------
// The codes of length L are the consecutive L-bit values first[L] .. first[L] + cnt[L] - 1, handed out in symbol order,
// and first[L + 1] = (first[L] + cnt[L]) << 1.  Written left-aligned in 15 bits (MSB = first stream bit) the code space is
// therefore cut into consecutive intervals, one per length, in increasing order of length:
//     a 15-bit string v opens with a code of length L  <=>  lim[L - 1] <= v < lim[L],   lim[L] = (first[L] + cnt[L]) << (15 - L)
// and that code is the (v >> (15 - L)) - first[L]-th of its length.  With the coded symbols listed by (length, symbol) its
// entry sits at position (v >> (15 - L)) + adj[L], adj[L] = offs[L] - first[L].  Nothing but lim[] and adj[] is needed, so
// every entry of a lookup table is filled independently -- one lane per entry instead of one scatter loop per symbol --
// and a code longer than the root table is decoded from lim[] / adj[] directly (no sub-tables).

// per-length counts -> lim / adj / offs, by one thread; returns zlib's verdict on the set (inflate_table, inftrees.c):
// over-subscribed, or incomplete unless it is the single one-bit code (or, for distances, no code at all)
__device__ __forceinline__ bool canon_from_counts(const uint32_t *cnt, CanonLds &c, bool codes_type)
{
	int left = 1, maxlen = 0;
	uint32_t code = 0, off = 0;
	bool ok = true;
	c.lim[0] = 0; c.adj[0] = 0; c.offs[0] = 0;
	for (int len = 1; len <= 15; len++) {
		const uint32_t cn = cnt[len];
		left = (left << 1) - (int)cn;
		if (left < 0) ok = false;
		if (cn) maxlen = len;
		c.lim[len] = ok ? (code + cn) << (15 - len) : 0u;
		c.adj[len] = off - code; c.offs[len] = off;
		code = (code + cn) << 1;
		off += cn;
	}
	if (left > 0 && maxlen != 0 && (codes_type || maxlen != 1)) ok = false;
	if (codes_type && maxlen == 0) ok = false;  // zlib goes on and fails at the missing end-of-block code
	return ok;
}

// code-length alphabet (19 symbols, <= 7 bits) from S.lens[0..19); S.ok = 0 if zlib would refuse the set
template <class G>
__device__ void build_cl(InfShared<G> &S)
{
	if (threadIdx.x < 16) S.cnt[threadIdx.x] = 0;
	__syncthreads();
	if (threadIdx.x < 19) { const int l = S.lens[threadIdx.x] & 7; if (l) atomicAdd(&S.cnt[l], 1u); }  // 3-bit fields: lengths are 0..7
	__syncthreads();
	if (threadIdx.x == 0) S.ok = canon_from_counts(S.cnt, S.ll_canon, true) ? 1 : 0;
	__syncthreads();
	if (threadIdx.x < (1 << CL_BITS) && S.ok) {
		const uint32_t v = (__brev((uint32_t)threadIdx.x) >> (32 - CL_BITS)) << (15 - CL_BITS);
		int L = 0;
		for (int len = CL_BITS; len >= 1; len--) if (v < S.ll_canon.lim[len]) L = len;
		uint8_t e = 0;
		if (L) {  // the want-th symbol of length L
			const uint32_t want = (v >> (15 - L)) + S.ll_canon.adj[L] - S.ll_canon.offs[L];
			uint32_t seen = 0;
			int sym = 0;
			for (int s = 0; s < 19; s++) {
				const bool hit = (S.lens[s] & 7) == L;
				if (hit && seen == want) sym = s;
				seen += hit ? 1u : 0u;
			}
			e = (uint8_t)((sym << 3) | L);
		}
		S.cl_tab[threadIdx.x] = e;
	}
	__syncthreads();
}

// decode-table entries.  bits 0-3: code bits to drop (both literals of a pair), 4-7: literal: bits of the first literal;
// length / distance: extra bits, 8-23: literals (first at 8, second at 16) or base value, 24-25 kind, 26: two literals
template <class G>
__device__ __forceinline__ uint32_t ll_entry(const InfShared<G> &S, int s, int bits)
{
	if (s < 256) return K_LIT | ((uint32_t)s << 8) | ((uint32_t)bits << 4) | (uint32_t)bits;
	if (s == 256) return K_EOB | (uint32_t)bits;
	if (s - 257 >= 29) return 0;  // 286, 287: invalid
	return K_LEN | ((uint32_t)S.lbase[s - 257] << 8) | ((uint32_t)S.lext[s - 257] << 4) | (uint32_t)bits;
}
template <class G>
__device__ __forceinline__ uint32_t d_entry(const InfShared<G> &S, int s, int bits)
{
	if (s >= 30) return 0;
	return K_LEN | ((uint32_t)S.dbase[s] << 8) | ((uint32_t)S.dext[s] << 4) | (uint32_t)bits;
}

// Lookup tables from S.lens[0..n), by the whole workgroup.  sent[]: the entry of every coded symbol, in (length, symbol)
// order.  Root table: FB bits; a literal/length entry holds two literals when both codes fit.  S.ok = 0: zlib would
// refuse the code lengths.
template <bool DIST, class G>
__device__ void build_tables(InfShared<G> &S, int n)
{
	GEO_CONSTANTS;
	constexpr int FB = DIST ? D_BITS : LL_BITS;
	uint32_t *tab = DIST ? S.d_tab : S.ll_tab;
	uint32_t *sent = DIST ? S.d_sent : S.ll_sent;
	CanonLds &C = DIST ? S.d_canon : S.ll_canon;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid < 16) S.cnt[tid] = 0;
	__syncthreads();
	for (int s = tid; s < n; s += NT) { const int l = S.lens[s]; if (l) atomicAdd(&S.cnt[l], 1u); }
	__syncthreads();
	if (tid == 0) S.ok = canon_from_counts(S.cnt, C, false) ? 1 : 0;
	// entries in (length, symbol) order: chunk k = symbols [64k, 64k + 64), one wave each; rank inside the chunk by ballots,
	// chunks before it from per-chunk counts
	constexpr int NCH = DIST ? 1 : 5;
	static_assert(NWV >= 4, "chunk loop below gives a wave at most two chunks");
	uint32_t rank0 = 0, rank1 = 0;
	int l0 = 0, l1 = 0;
	for (int k = wave; k < NCH; k += NWV) {
		const int s = k * 64 + lane;
		const int l = s < n ? (int)S.lens[s] : 0;
		uint32_t r = 0, mine = 0;
#pragma unroll
		for (int b = 1; b <= 15; b++) {
			const uint64_t bal = __ballot(l == b);
			if (l == b) r = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
			if (lane == b) mine = (uint32_t)__popcll(bal);
		}
		if (k == wave) { rank0 = r; l0 = l; } else { rank1 = r; l1 = l; }
		if (lane < 16) S.chunkcnt[k * 16 + lane] = (uint8_t)mine;
	}
	__syncthreads();
	if (!S.ok) { __syncthreads(); return; }
	for (int k = wave; k < NCH; k += NWV) {
		const int s = k * 64 + lane, l = k == wave ? l0 : l1;
		if (l) {
			uint32_t pos = C.offs[l] + (k == wave ? rank0 : rank1);
			for (int q = 0; q < k; q++) pos += S.chunkcnt[q * 16 + l];
			sent[pos] = DIST ? d_entry(S, s, l) : ll_entry(S, s, l);
		}
	}
	uint32_t lim[FB + 1], adj[FB + 1];
#pragma unroll
	for (int len = 1; len <= FB; len++) { lim[len] = C.lim[len]; adj[len] = C.adj[len]; }
	__syncthreads();
	for (int idx = tid; idx < (1 << FB); idx += NT) {
		const uint32_t rev = __brev((uint32_t)idx) >> (32 - FB);
		const uint32_t v = rev << (15 - FB);
		int L1 = 0;
		uint32_t a1 = 0;
#pragma unroll
		for (int len = FB; len >= 1; len--) if (v < lim[len]) { L1 = len; a1 = adj[len]; }
		uint32_t e = 0;
		if (L1) {
			e = sent[(v >> (15 - L1)) + a1];
			if (!DIST && (e & (3u << 24)) == K_LIT && L1 < FB) {  // a second literal in the bits that remain?
				const uint32_t v2 = ((rev << L1) & ((1u << FB) - 1u)) << (15 - FB);
				int L2 = 0;
				uint32_t a2 = 0;
#pragma unroll
				for (int len = FB; len >= 1; len--) if (v2 < lim[len]) { L2 = len; a2 = adj[len]; }
				if (L2 && L1 + L2 <= FB) {
					const uint32_t e2 = sent[(v2 >> (15 - L2)) + a2];
					if ((e2 & (3u << 24)) == K_LIT) e = K_LIT | K_TWO | (((e2 >> 8) & 0xFFu) << 16) | (e & 0xFF00u) | ((uint32_t)L1 << 4) | (uint32_t)(L1 + L2);
				}
			}
		}
		tab[idx] = e;
	}
	__syncthreads();
}

// per-lane bit reader over the staged input; bit positions are 32-bit offsets from the round's origin dword
struct LaneBits { uint64_t buf; int cnt; uint32_t next; };  // next = dword index relative to br.src

template <class G>
__device__ __forceinline__ void lane_init(const InfShared<G> &S, LaneBits &lb, uint32_t org_dword, uint32_t rel_bit)
{
	GEO_CONSTANTS;
	const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
	const uint32_t idx = org_dword + (rel_bit >> 5), sh = rel_bit & 31u;
	const uint64_t w = (uint64_t)in32[idx & (INF_IN / 4 - 1)] | ((uint64_t)in32[(idx + 1) & (INF_IN / 4 - 1)] << 32);
	lb.buf = w >> sh; lb.cnt = 64 - (int)sh; lb.next = idx + 2;
}
template <class G>
__device__ __forceinline__ void lane_refill(const InfShared<G> &S, LaneBits &lb)
{
	GEO_CONSTANTS;
	if (lb.cnt <= 32) {
		const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
		lb.buf |= (uint64_t)in32[lb.next & (INF_IN / 4 - 1)] << lb.cnt;
		lb.cnt += 32; lb.next++;
	}
}
__device__ __forceinline__ uint32_t lane_pos(const LaneBits &lb, uint32_t org_dword) { return (lb.next - org_dword) * 32u - (uint32_t)lb.cnt; }

// Decode the symbols that START in [start, end) (one lane; bit positions relative to the round's origin dword).
// EMIT = false: only measure (landing position, output bytes, copies).  EMIT = true: literals go to the ring
// at `o`, copies to the list at `mi`.
// A wave pays for every branch any of its lanes takes, and the rare symbols (copies, codes longer than the root table,
// end of block) are rare per lane but not per wave.  So the loop runs FAST_STEPS literal-only steps (no branch: a lane
// whose next symbol is not a literal simply does not move) for every general step.  The landing position must not depend
// on how a lane got there (restarts compare landings): the second literal of a pair is taken only if it starts before
// `end`, which makes the landing the first symbol boundary at or after `end` whatever the pairing.
#ifndef CCT_INF_FAST_STEPS
#define CCT_INF_FAST_STEPS 3
#endif
constexpr int FAST_STEPS = CCT_INF_FAST_STEPS;
template <bool EMIT, class G>
__device__ __forceinline__ void walk_segment(InfShared<G> &S, uint32_t org_dword, uint32_t start, uint32_t end, uint32_t &land,
                                             uint32_t &nbytes, uint32_t &nmatch, uint32_t &flags, uint32_t o, uint32_t mi,
                                             uint32_t *steps = nullptr)
{
	GEO_CONSTANTS;
	const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
	LaneBits lb;
	lane_init(S, lb, org_dword, start);
	uint32_t p = start;  // bit position of the next symbol
	nbytes = 0; nmatch = 0; flags = 0;
	// interval limits of the codes longer than the root tables: wave-uniform, fetched once per walk
	uint32_t ll_lim[16], ll_adj[16], d_lim[16], d_adj[16];
#pragma unroll
	for (int len = LL_BITS; len <= 15; len++) { ll_lim[len] = S.ll_canon.lim[len]; ll_adj[len] = S.ll_canon.adj[len]; }
#pragma unroll
	for (int len = D_BITS; len <= 15; len++) { d_lim[len] = S.d_canon.lim[len]; d_adj[len] = S.d_canon.adj[len]; }
	auto long_ll = [&](uint32_t lo) -> uint32_t {  // a code longer than the root table (or none): the interval test, see CanonLds
		const uint32_t v = __brev(lo) >> 17;
		int L = 0;
		uint32_t a = 0;
#pragma unroll
		for (int k = 15; k > LL_BITS; k--) if (v < ll_lim[k]) { L = k; a = ll_adj[k]; }
		if (v < ll_lim[LL_BITS]) L = 0;  // a short code whose symbol is not valid (root entry 0)
		return L ? S.ll_sent[(v >> (15 - L)) + a] : 0u;
	};
	auto long_d = [&](uint32_t lo) -> uint32_t {
		const uint32_t v = __brev(lo) >> 17;
		int L = 0;
		uint32_t a = 0;
#pragma unroll
		for (int k = 15; k > D_BITS; k--) if (v < d_lim[k]) { L = k; a = d_adj[k]; }
		if (v < d_lim[D_BITS]) L = 0;
		return L ? S.d_sent[(v >> (15 - L)) + a] : 0u;
	};
	// invariant at the top of a step: more than 21 valid bits in lb.buf (a root lookup needs 11)
	auto literal_step = [&](uint32_t e, bool live) {
		const bool lit = live && (e & (3u << 24)) == K_LIT;
		const uint32_t l1 = (e >> 4) & 15u;
		const bool two = (e & K_TWO) != 0 && p + l1 < end;
		const uint32_t used = lit ? (two ? (e & 15u) : l1) : 0u;
		if (EMIT && lit) {
			S.ring[(o + nbytes) & INF_RMASK] = (uint8_t)(e >> 8);
			if (two) S.ring[(o + nbytes + 1u) & INF_RMASK] = (uint8_t)(e >> 16);
		}
		nbytes += lit ? (two ? 2u : 1u) : 0u;
		lb.buf >>= used; lb.cnt -= (int)used; p += used;
		return lit;
	};
	for (;;) {
#pragma unroll
		for (int k = 0; k < FAST_STEPS; k++) {
			const uint32_t w = in32[lb.next & (INF_IN / 4 - 1)];  // next dword, in flight together with the table lookup
			const uint32_t e = S.ll_tab[(uint32_t)lb.buf & ((1u << LL_BITS) - 1u)];
			const bool take = lb.cnt <= 32;
			lb.buf |= take ? (uint64_t)w << (lb.cnt & 63) : 0ull;
			lb.cnt += take ? 32 : 0; lb.next += take ? 1u : 0u;
			literal_step(e, p < end);
		}
		if (p >= end) break;
		if (steps) ++*steps;
		if (nbytes >= (uint32_t)LANE_OUT_CAP) { flags = SEG_CUT; break; }  // keeps a round inside the output ring
		lane_refill(S, lb);
		const uint32_t lo = (uint32_t)lb.buf;
		uint32_t e = S.ll_tab[lo & ((1u << LL_BITS) - 1u)];
		if (e == 0) {
			if (steps) ++steps[1];
			e = long_ll(lo);
			if (e == 0) { flags = SEG_BAD; break; }
		}
		if (steps && (e & (3u << 24)) == K_LEN) ++steps[2];
		if (!literal_step(e, true)) {
			const uint32_t kind = e & (3u << 24);
			if (kind == K_EOB) { lb.buf >>= (e & 15u); lb.cnt -= (int)(e & 15u); p += e & 15u; flags = SEG_EOB; break; }
			const uint32_t cb = e & 15u, xb = (e >> 4) & 15u;
			const uint32_t val = ((e >> 8) & 0xFFFFu) + ((lo >> cb) & ((1u << xb) - 1u));  // code <= 15, extra <= 5 bits: inside lo
			lb.buf >>= (cb + xb); lb.cnt -= (int)(cb + xb); p += cb + xb;
			lane_refill(S, lb);
			const uint32_t dlo = (uint32_t)lb.buf;
			uint32_t de = S.d_tab[dlo & ((1u << D_BITS) - 1u)];
			if (de == 0) {
				de = long_d(dlo);
				if (de == 0) { flags = SEG_BAD; break; }
			}
			const uint32_t dcb = de & 15u, dxb = (de >> 4) & 15u;
			const uint32_t dist = ((de >> 8) & 0xFFFFu) + ((dlo >> dcb) & ((1u << dxb) - 1u));  // code <= 15, extra <= 13 bits
			lb.buf >>= (dcb + dxb); lb.cnt -= (int)(dcb + dxb); p += dcb + dxb;
			if (EMIT) {
				if (dist > o + nbytes) { flags = SEG_BAD; break; }  // distance too far back
				S.mlist[mi + nmatch] = make_uint2(o + nbytes, val | ((dist - 1u) << 9));
			}
			nbytes += val; nmatch++;
		}
		lane_refill(S, lb);
	}
	land = p;
}

// The code-length sequence of a dynamic block header (<= 316 entries, run-length coded with the 19-symbol code) is one
// more dependent chain: where symbol k + 1 starts is known after symbol k.  It is short (<= 4424 bits), so instead of
// speculating the whole workgroup ranks the list: every bit position of a CL_WIN-bit window looks up "the symbol that would
// start here" (next position, entries produced), nine rounds of pointer doubling make that "2^k symbols on", and before
// each doubling the positions already known to be on the true chain (position 0 at first) mark the position 2^k symbols
// after them with its entry index.  After nine rounds every chain symbol within 511 symbols of the start is marked -- more than
// the 316 entries a sequence can have -- and the marked positions write their entries.  A "repeat the previous length"
// symbol (16) writes markers that a scan resolves afterwards.  The arrays live in the copy list and the root tables, idle while a header is parsed.
constexpr int CL_WIN = 2048, CL_WINX = CL_WIN + 16;  // positions CL_WIN.. are "outside": a symbol is at most 14 bits long
constexpr int CL_LEVELS = 9;
constexpr uint32_t CL_BAD = 1, CL_OVER = 2;
constexpr uint16_t CL_NONE = 0xFFFF, CL_SAT = 0x7FFF;
constexpr uint8_t CL_PREV = 0xFF;  // marker: same as the entry before (lengths are <= 15)
static_assert(CL_WINX * sizeof(uint16_t) <= MLIST_CAP * sizeof(uint2) && 2 * CL_WINX <= (1 << LL_BITS) + (1 << D_BITS), "the ranking arrays fit the copy list and the root tables");

// symbol that starts at bit `bit` (relative to br.src): bits it takes (0: no such code) and entries it produces / their value
template <class G>
__device__ __forceinline__ uint32_t cl_symbol(const InfShared<G> &S, uint64_t bit, uint32_t &rep, uint8_t &val)
{
	GEO_CONSTANTS;
	const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
	const uint32_t idx = (uint32_t)(bit >> 5), sh = (uint32_t)bit & 31u;
	const uint64_t w = (uint64_t)in32[idx & (INF_IN / 4 - 1)] | ((uint64_t)in32[(idx + 1) & (INF_IN / 4 - 1)] << 32);
	const uint32_t lo = (uint32_t)(w >> sh);
	const uint32_t ce = S.cl_tab[lo & ((1u << CL_BITS) - 1u)];
	if (ce == 0) { rep = 0; val = 0; return 0; }
	const uint32_t sym = ce >> 3, cb = ce & 7u;
	const uint32_t xb = sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u;
	const uint32_t xv = (lo >> cb) & ((1u << xb) - 1u);
	rep = sym < 16 ? 1u : sym == 18 ? 11u + xv : 3u + xv;
	val = sym < 16 ? (uint8_t)sym : sym == 16 ? CL_PREV : (uint8_t)0;
	return cb + xb;
}

// Decodes `total` entries starting at bit b0 into S.lens (CL_PREV markers unresolved); returns 0 and the bit position after
// the sequence, or an error flag.  Called by the whole workgroup, uniform result.
template <class G>
__device__ uint32_t cl_sequence(InfShared<G> &S, uint64_t b0, uint32_t total, uint64_t &bend)
{
	GEO_CONSTANTS;
	constexpr int PER = CL_WIN / NT;  // window positions per lane
	// (next position | entries << 16) per position, two copies (a round reads one and writes the other: one barrier per round) in
	// the root tables, which are dead while a header is parsed; the chain marks (entry index, CL_NONE = not on the chain) in the copy list
	uint32_t *A = S.ll_tab, *B = S.ll_tab + CL_WINX;
	uint16_t *ix = reinterpret_cast<uint16_t *>(S.mlist);
	const int tid = threadIdx.x;
	uint64_t bp = b0;
	uint32_t done = 0;
	for (;;) {
		const uint32_t remaining = total - done;
		for (int i = tid; i < CL_WINX; i += NT) {
			uint32_t u = (uint32_t)i, rep = 0;
			if (i < CL_WIN) {
				uint8_t v;
				u += cl_symbol(S, bp + (uint64_t)i, rep, v);  // a position without a code points at itself
			} else B[i] = u;                                   // outside positions never change
			A[i] = u | (rep << 16);
			ix[i] = i == 0 ? (uint16_t)0 : CL_NONE;
		}
		if (tid == 0) { S.rres[0] = 0xFFFFFFFFu; S.rres[1] = 0xFFFFFFFFu; S.rres[2] = 0; }
		__syncthreads();
		for (int level = 0; level < CL_LEVELS; level++) {
			// a position marked during this very round may or may not be seen by another lane: either way what that lane writes is
			// a true chain position with its true index
#pragma unroll
			for (int k = 0; k < PER; k++) {
				const int i = tid + k * NT;
				const uint32_t w = A[i], j = w & 0xFFFFu;
				const uint32_t wj = A[j];
				const uint32_t x = ix[i];
				B[i] = (wj & 0xFFFFu) | (min((w >> 16) + (wj >> 16), (uint32_t)CL_SAT) << 16);
				if (x != CL_NONE && j != (uint32_t)i) ix[j] = (uint16_t)min(x + (w >> 16), (uint32_t)CL_SAT);
			}
			uint32_t *t = A; A = B; B = t;
			__syncthreads();
		}
		// marked positions before the end of the sequence write their entries; the one AT the end is where the block body starts
		uint32_t fl = 0;
		for (int i = tid; i < CL_WINX; i += NT) {
			const uint32_t x = ix[i];
			if (x == CL_NONE || x > remaining) continue;
			if (x == remaining) { S.rres[0] = (uint32_t)i; continue; }          // unique: every symbol produces an entry
			if (i >= CL_WIN) { S.rres[1] = (uint32_t)i | (x << 16); continue; }  // the chain leaves the window here (unique as well)
			uint32_t rep; uint8_t v;
			if (cl_symbol(S, bp + (uint64_t)i, rep, v) == 0) { fl |= CL_BAD; continue; }
			if (rep > remaining - x) { fl |= CL_OVER; continue; }               // repeat beyond the last entry
			for (uint32_t t = 0; t < rep; t++) S.lens[done + x + t] = v;
		}
		if (__syncthreads_or(fl != 0)) return CL_BAD;
		const uint32_t endpos = S.rres[0], cont = S.rres[1];
		__syncthreads();  // everybody has read the result slots before the next window resets them
		if (endpos != 0xFFFFFFFFu) { bend = bp + endpos; return 0; }
		if (cont == 0xFFFFFFFFu || (cont >> 16) == 0) return CL_BAD;  // cannot happen for a chain without a bad position
		bp += cont & 0xFFFFu;
		done += cont >> 16;
	}
}

// smallest lane index (0..NT-1) whose predicate is set, NT if none; all lanes call it (one barrier inside, and
// what was written to LDS before the call is visible to everybody after it)
template <class G>
__device__ __forceinline__ int first_lane_with(InfShared<G> &S, bool pred, int slot)
{
	GEO_CONSTANTS;
	const uint64_t bal = __ballot(pred);
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) S.wred[slot * NWV + wave] = bal ? (uint32_t)(wave * 64 + __ffsll((long long)bal) - 1) : (uint32_t)NT;
	__syncthreads();
	uint32_t r = (uint32_t)NT;
#pragma unroll
	for (int w = 0; w < NWV; w++) r = min(r, S.wred[slot * NWV + w]);
	return (int)r;
}

// tuning runs only (CCT_INF_PROF=1): cycles per phase and event counts of every stream, written by lane 0
enum { P_M_SETUP, P_HDR, P_TABLES, P_T_CLLENS, P_T_CL, P_T_CLWALK, P_T_LL, P_STAGE, P_WALK0, P_RESTART, P_SCAN, P_EMIT, P_MATCH, P_FLUSH, P_TOTAL, C_BLOCKS, C_ROUNDS, C_PASSES, C_MATCHES,
       C_DEP, C_BATCH, C_LASTL, C_STEPMAX, C_STEPSUM, C_MATCHMAX, C_W0CYC, C_W0STEPS, P_N };
#define PROF_T(slot) do { if (PROF) { const long long t_ = clock64(); prof[slot] += (uint64_t)(t_ - tprev); tprev = t_; } } while (0)
#define PROF_C(slot, v) do { if (PROF) prof[slot] += (uint64_t)(v); } while (0)

template <class G, bool PROF>
__global__ void __launch_bounds__(G::NT) inflate_kernel(InflateArgs a, uint64_t *prof_out)
{
	GEO_CONSTANTS;
	uint64_t prof[P_N] = {};
	long long tprev = PROF ? clock64() : 0;
	const long long tstart = tprev;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
	InfShared<G> &S = *reinterpret_cast<InfShared<G> *>(smem_raw);
	const int s = blockIdx.x;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint64_t f0 = a.offsets[s] + (uint64_t)a.skip, f1 = a.offsets[s + 1];
	uint8_t *out = a.out + (size_t)s * a.out_stride;
	uint32_t err = 0;                 // identical in every lane: control flow is workgroup-uniform
	uint32_t pos = 0, flushed = 0;
	uint32_t adA = 1, adB = 0;

	BitReader br;
	const uint64_t al = f0 & ~(uint64_t)15;
	br.src = a.in + al;
	br.avail = a.in_total > al ? a.in_total - al : 0;
	br.nbytes = f1 > al ? f1 - al : 0;
	br.bytepos = (f0 - al) & ~(uint64_t)3;
	br.staged_end = br.bytepos & ~(uint64_t)(INF_CHUNK - 1);
	br.buf = 0; br.cnt = 0;
	if (f1 < f0 + 6) err = CCT_ST_ZLIB;  // shorter than header + trailer
	if (tid < 29) { S.lbase[tid] = c_lbase[tid]; S.lext[tid] = c_lext[tid]; }
	if (tid < 30) { S.dbase[tid] = c_dbase[tid]; S.dext[tid] = c_dext[tid]; }
	__syncthreads();

	auto flush_to = [&](uint32_t upto) {  // ring [flushed, upto) -> HBM as 16-byte stores, Adler-32 on the way
		while (flushed < upto) {
			const uint32_t n = min((uint32_t)INF_FLUSH, upto - flushed);
			if ((size_t)flushed + ((n + 15) & ~15u) > a.out_stride) { err |= CCT_ST_STREAM; flushed += n; continue; }  // longer than any valid payload
			uint32_t sa = 0, sb = 0;  // A += sum d ; B += n * A_old + sum (n - i) d_i
			for (uint32_t t = (uint32_t)tid * 16; t < n; t += NT * 16) {  // flushed is a multiple of INF_FLUSH here
				const uint4 v = *reinterpret_cast<const uint4 *>(S.ring + ((flushed + t) & INF_RMASK));
				const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const uint32_t d = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
					if (t + k < n) { sa += d; sb += (n - t - k) * d; }
				}
				*reinterpret_cast<uint4 *>(out + flushed + t) = v;
			}
			for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
			__syncthreads();
			if (lane == 0) { S.wsum_b[wave] = sa; S.wsum_m[wave] = sb; }
			__syncthreads();
			sa = 0; sb = 0;
#pragma unroll
			for (int w = 0; w < NWV; w++) { sa += S.wsum_b[w]; sb += S.wsum_m[w]; }
			adB = (uint32_t)(((uint64_t)adB + (uint64_t)n * adA + sb) % 65521u);
			adA = (adA + sa) % 65521u;
			flushed += n;
		}
	};
	auto runaway = [&]() -> bool { return consumed_bytes(br) > br.nbytes + 16; };  // decoding zeros past the end

	if (!err) {
		refill(S, br);
		getbits(br, (int)((f0 - al) & 3u) * 8);
		refill(S, br);
		const uint32_t cmf = getbits(br, 8), flg = getbits(br, 8);
		if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) err = CCT_ST_ZLIB;  // incorrect header check
	}
	bool last = false;
	PROF_T(P_HDR);
	while (!err && !last) {
		refill(S, br);
		last = getbits(br, 1) != 0;
		PROF_C(C_BLOCKS, 1);
		const uint32_t type = getbits(br, 2);
		if (type == 0) {  // stored
			const int drop = br.cnt & 7;
			getbits(br, drop);
			refill(S, br);
			const uint32_t len = getbits(br, 16);
			refill(S, br);
			const uint32_t nlen = getbits(br, 16);
			if ((len ^ 0xFFFFu) != nlen) { err = CCT_ST_ZLIB; break; }
			for (uint32_t i = 0; i < len; i++) {  // byte-wise through the bit buffer keeps one input path
				refill(S, br);
				const uint32_t b = getbits(br, 8);
				if (tid == 0) S.ring[pos & INF_RMASK] = (uint8_t)b;
				pos++;
				if (pos - flushed >= 2 * INF_FLUSH) { __syncthreads(); flush_to(pos & ~(uint32_t)(INF_FLUSH - 1)); }
			}
			__syncthreads();
			if (runaway()) { err = CCT_ST_ZLIB; break; }
			continue;
		}
		if (type == 3) { err = CCT_ST_ZLIB; break; }
		if (type == 1) {  // fixed codes
			__syncthreads();
			for (int i = tid; i < 288; i += NT) S.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
			__syncthreads();
			build_tables<false>(S, 288);
			for (int i = tid; i < 32; i += NT) S.lens[i] = 5;  // fixed codes name 32 distance symbols (30, 31 invalid)
			__syncthreads();
			build_tables<true>(S, 32);
		} else {          // dynamic codes
			refill(S, br);
			const int nlen = (int)getbits(br, 5) + 257, ndist = (int)getbits(br, 5) + 1, ncode = (int)getbits(br, 4) + 4;
			if (nlen > 286 || ndist > 30) { err = CCT_ST_ZLIB; break; }
			__syncthreads();
			if (tid < 19) S.lens[tid] = 0;
			__syncthreads();
			{  // ncode 3-bit fields, one lane each, read at their bit offsets
				while (br.bytepos + 16 > br.staged_end) stage_chunk(S, br);
				const uint64_t f0b = br.bytepos * 8u - (uint64_t)br.cnt;
				if (tid < ncode) {
					const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
					const uint64_t bit = f0b + 3u * (uint32_t)tid;
					const uint32_t idx = (uint32_t)(bit >> 5), sh = (uint32_t)bit & 31u;
					const uint64_t w = (uint64_t)in32[idx & (INF_IN / 4 - 1)] | ((uint64_t)in32[(idx + 1) & (INF_IN / 4 - 1)] << 32);
					S.lens[c_clorder[tid]] = (uint8_t)((w >> sh) & 7u);
				}
				const uint64_t f1b = f0b + 3u * (uint32_t)ncode;
				br.bytepos = (f1b >> 5) * 4u; br.buf = 0; br.cnt = 0;
				refill(S, br);
				getbits(br, (int)(f1b & 31u));
			}
			__syncthreads();
			PROF_T(P_T_CLLENS);
			build_cl(S);
			PROF_T(P_T_CL);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
			// the code lengths of both alphabets, run-length coded; S.lens is reused after cl is built
			{
				while (br.bytepos + 2048 > br.staged_end) stage_chunk(S, br);  // the sequence is < 700 bytes
				const uint32_t total = (uint32_t)(nlen + ndist);
				const uint64_t b0 = br.bytepos * 8u - (uint64_t)br.cnt;
				uint64_t bp = b0;
				const uint32_t cl_err = cl_sequence(S, b0, total, bp);
				if (wave == 0) {
					uint32_t lerr = cl_err;
					if (!lerr) {  // "same as the entry before": every lane owns five consecutive entries
						uint8_t v[5];
						uint32_t lastv = 0x100;  // last definite value in this lane's stretch (0x100: none)
#pragma unroll
						for (int k = 0; k < 5; k++) {
							const uint32_t e = (uint32_t)lane * 5 + k;
							v[k] = e < total ? S.lens[e] : (uint8_t)0;
							if (e < total && v[k] != CL_PREV) lastv = v[k];
						}
						uint32_t carry = lastv;  // inclusive "last definite value up to this lane"
#pragma unroll
						for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(carry, d, 64); if (lane >= d && carry == 0x100) carry = o; }
						uint32_t prev = __shfl_up(carry, 1, 64);
						if (lane == 0) prev = 0x100;
#pragma unroll
						for (int k = 0; k < 5; k++) {
							const uint32_t e = (uint32_t)lane * 5 + k;
							if (e < total) {
								if (v[k] == CL_PREV) { if (prev == 0x100) lerr = 1; else S.lens[e] = (uint8_t)prev; }
								else prev = v[k];
							}
						}
						if (__ballot(lerr != 0)) lerr = 1;  // a repeat with nothing before it
					}
					if (lane == 0) { S.rres[0] = (uint32_t)bp; S.rres[1] = (uint32_t)(bp >> 32); S.rres[2] = lerr; }
				}
				__syncthreads();
				if (S.rres[2]) err = CCT_ST_ZLIB;
				const uint64_t bend = (uint64_t)S.rres[0] | ((uint64_t)S.rres[1] << 32);
				br.bytepos = (bend >> 5) * 4u; br.buf = 0; br.cnt = 0;
				refill(S, br);
				getbits(br, (int)(bend & 31u));
			}
			if (err) break;
			__syncthreads();
			PROF_T(P_T_CLWALK);
			if (S.lens[256] == 0) { err = CCT_ST_ZLIB; break; }  // no end-of-block code
			// distance lengths follow the literal/length lengths: move them to their own array slot first
			uint8_t dl = 0;
			if (tid < ndist) dl = S.lens[nlen + tid];
			__syncthreads();
			build_tables<false>(S, nlen);
			PROF_T(P_T_LL);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
			if (tid < 32) S.lens[tid] = tid < ndist ? dl : 0;
			__syncthreads();
			build_tables<true>(S, ndist);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
		}
		// ---- symbols of this block, in speculative rounds (see the header comment)
		uint64_t bitpos = br.bytepos * 8u - (uint64_t)br.cnt;  // true position of the next symbol
		PROF_T(P_TABLES);
		for (bool block_done = false; !block_done && !err;) {
			__syncthreads();
			if (pos - flushed >= (uint32_t)INF_FLUSH) flush_to(pos & ~(uint32_t)(INF_FLUSH - 1));
			PROF_T(P_FLUSH);
			PROF_C(C_ROUNDS, 1);
			if ((bitpos >> 3) > br.nbytes + 16) { err |= CCT_ST_ZLIB; break; }  // decoding zeros past the end
			const uint64_t need_end = ((bitpos + (uint64_t)NT * SEG_BITS) >> 3) + 32;
			while (need_end > br.staged_end) stage_chunk(S, br);
			PROF_T(P_STAGE);
			const uint32_t org_dword = (uint32_t)(bitpos >> 5);      // lanes address bits relative to this dword
			const uint32_t org_bit = (uint32_t)bitpos & 31u;
			const uint32_t seg_end = org_bit + (uint32_t)(tid + 1) * SEG_BITS;
			uint32_t start = org_bit + (uint32_t)tid * SEG_BITS, land, nb, nm, fl;
			uint32_t nsteps[3] = {0, 0, 0};
			const long long tw0 = PROF ? clock64() : 0;
			walk_segment<false>(S, org_dword, start, seg_end, land, nb, nm, fl, 0, 0, PROF ? nsteps : nullptr);
			if (PROF) {
				const long long tw1 = clock64();
				uint32_t mx = nsteps[0];
				for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
				PROF_C(C_W0CYC, tw1 - tw0); PROF_C(C_W0STEPS, mx);
				if (tid == 0) { S.rres[0] = 0; S.rres[1] = 0; S.rres[2] = 0; }
				__syncthreads();
				atomicMax(&S.rres[0], nsteps[0]); atomicAdd(&S.rres[1], nsteps[0]); atomicAdd(&S.rres[2], nsteps[1]);
				__syncthreads();
				PROF_C(C_STEPMAX, S.rres[0]); PROF_C(C_STEPSUM, S.rres[1]); PROF_C(C_MATCHMAX, S.rres[2]);
				__syncthreads();
			}
			PROF_T(P_WALK0);
			for (;;) {  // restart from where the previous lane really landed until nothing moves
				__syncthreads();
				PROF_C(C_PASSES, 1);
				S.land[tid] = land;
				const int first = first_lane_with(S, fl != 0, 0);
				const uint32_t pl = tid ? S.land[tid - 1] : start;
				const bool moved = tid > 0 && tid <= first && pl != start;
				if (!__syncthreads_or(moved)) break;
				if (moved) { start = pl; walk_segment<false>(S, org_dword, start, seg_end, land, nb, nm, fl, 0, 0); }
			}
			// the chain is true up to the first lane that saw end-of-block or an invalid code; cut the round
			// where the ring or the copy list would overflow (lane 0 always fits)
			PROF_T(P_RESTART);
			int lastl = min(first_lane_with(S, fl != 0, 1), NT - 1);
			uint32_t cb = nb, cm = nm;  // inclusive prefix sums over the workgroup
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t ob = __shfl_up(cb, d, 64), om = __shfl_up(cm, d, 64);
				if (lane >= d) { cb += ob; cm += om; }
			}
			if (lane == 63) { S.wsum_b[wave] = cb; S.wsum_m[wave] = cm; }
			__syncthreads();
			for (int w = 0; w < wave; w++) { cb += S.wsum_b[w]; cm += S.wsum_m[w]; }
			const int over = first_lane_with(S, tid > 0 && (cb > (uint32_t)ROUND_OUT_BUDGET || cm > (uint32_t)MLIST_CAP), 0);
			if (over < NT) lastl = min(lastl, over - 1);
			const bool mine = tid <= lastl;
			const uint32_t o = pos + cb - nb, mi = cm - nm;
			uint32_t efl = 0;
			PROF_T(P_SCAN);
			PROF_C(C_LASTL, lastl + 1);
			if (mine) {
				uint32_t l2, b2, m2;
				walk_segment<true>(S, org_dword, start, seg_end, l2, b2, m2, efl, o, mi);
			}
			if (tid == lastl) { S.rres[0] = cb; S.rres[1] = cm; S.rres[2] = land; S.rres[3] = fl; }
			if (__syncthreads_or(mine && (efl & SEG_BAD))) { err |= CCT_ST_ZLIB; break; }
			const uint32_t round_bytes = S.rres[0], nmatch_round = S.rres[1], round_land = S.rres[2], lfl = S.rres[3];
			PROF_T(P_EMIT);
			PROF_C(C_MATCHES, nmatch_round);
			// LZ77 copies of the round, NT at a time, one lane per copy.  Literals are in place already, so a copy only has
			// to wait for the earlier copies whose destination lies inside its source (few do: most copies are 3-4 bytes
			// from kilobytes back).  Destinations are disjoint and ascending, so those are a contiguous stretch of the list,
			// found once by binary search; then rounds of "run what is ready" until the batch is done (the first copy that
			// has not run is always ready).  Short copies are done by their lane, long ones by its wave.
			// A run (a byte or a short pattern repeated for kilobytes) arrives as a chain of adjacent overlapping copies of
			// the same distance, each reading the tail of the one before.  Such a copy continues the pattern that precedes
			// the head of its chain, so it takes its bytes from there and waits for nothing inside the chain.
			constexpr uint32_t SHORT_COPY = 16;
			for (uint32_t k0 = 0; k0 < nmatch_round; k0 += NT) {
				const uint32_t k = k0 + (uint32_t)tid;
				const bool valid = k < nmatch_round;
				uint2 m = make_uint2(0, 0);
				if (valid) m = S.mlist[k];
				const uint32_t len = m.y & 511u, dist = (m.y >> 9) + 1u;
				const bool periodic = dist < len;
				uint32_t head = k, hx = m.x;
				if (valid && periodic)
					while (head > k0) {
						const uint2 mm = S.mlist[head - 1];
						const uint32_t ml = mm.y & 511u;
						if ((mm.y >> 9) + 1u != dist || dist >= ml || mm.x + ml != hx) break;
						hx = mm.x; head--;
					}
				// bytes come from [s0, s0 + len) or, periodic, from the `dist` bytes at s0 starting at `phase`
				const uint32_t s0 = hx - dist, s1 = periodic ? hx : s0 + len;
				const uint32_t phase = periodic ? (m.x - hx) % dist : 0u;
				uint32_t lo = head;  // first copy of the batch whose destination ends after s0
				if (valid && head > k0 && s1 > S.mlist[k0].x) {
					uint32_t a = k0, b = head;
					while (a < b) {
						const uint32_t mid = (a + b) >> 1;
						const uint2 mm = S.mlist[mid];
						if (mm.x + (mm.y & 511u) > s0) b = mid; else a = mid + 1;
					}
					lo = a;
				}
				S.land[tid] = 0;  // "has run" flags of the batch (the landing positions are not needed any more)
				__syncthreads();
				PROF_T(P_M_SETUP);
				bool pending = valid;
				do {
					PROF_C(C_BATCH, 1);
					bool ready = pending;
					for (uint32_t j = lo; ready && j < head; j++) {
						if (S.mlist[j].x >= s1) break;
						if (!S.land[j - k0]) ready = false;
					}
					if (ready && len <= SHORT_COPY) {
						if (!periodic) {  // all loads, then all stores: one LDS round trip instead of one per byte
							uint8_t b[SHORT_COPY];
#pragma unroll
							for (uint32_t i = 0; i < SHORT_COPY; i++) b[i] = i < len ? S.ring[(s0 + i) & INF_RMASK] : (uint8_t)0;
#pragma unroll
							for (uint32_t i = 0; i < SHORT_COPY; i++) if (i < len) S.ring[(m.x + i) & INF_RMASK] = b[i];
						} else {
							uint32_t r = phase;
							for (uint32_t i = 0; i < len; i++) {
								S.ring[(m.x + i) & INF_RMASK] = S.ring[(s0 + r) & INF_RMASK];
								if (++r == dist) r = 0;
							}
						}
					}
					for (uint64_t lm = __ballot(ready && len > SHORT_COPY); lm; lm &= lm - 1) {
						const int src = __ffsll((long long)lm) - 1;
						const uint32_t bx = (uint32_t)__shfl((int)m.x, src), bs = (uint32_t)__shfl((int)s0, src);
						const uint32_t bl = (uint32_t)__shfl((int)len, src), bd = (uint32_t)__shfl((int)dist, src);
						const uint32_t bp = (uint32_t)__shfl((int)phase, src);
						if (bd < bl) {  // the pattern of bd bytes at bs, from phase bp on
							uint32_t r = (bp + (uint32_t)lane) % bd;
							const uint32_t step = 64u % bd;
							for (uint32_t i = lane; i < bl; i += 64) {
								S.ring[(bx + i) & INF_RMASK] = S.ring[(bs + r) & INF_RMASK];
								r += step; if (r >= bd) r -= bd;
							}
						} else {
							for (uint32_t i = lane; i < bl; i += 64) S.ring[(bx + i) & INF_RMASK] = S.ring[(bs + i) & INF_RMASK];
						}
					}
					if (ready) { S.land[tid] = 1; pending = false; }  // after the bytes: LDS operations of a wave complete in order
				} while (__syncthreads_or(pending));
			}
			PROF_T(P_MATCH);
			pos += round_bytes;
			bitpos = (uint64_t)org_dword * 32u + round_land;
			if (lfl & SEG_BAD) { err |= CCT_ST_ZLIB; break; }
			if (lfl & SEG_EOB) block_done = true;
		}
		if (err) break;
		// hand the position back to the workgroup-uniform reader
		__syncthreads();
		br.bytepos = (bitpos >> 5) * 4u; br.buf = 0; br.cnt = 0;
		refill(S, br);
		getbits(br, (int)(bitpos & 31u));
		if (runaway()) err |= CCT_ST_ZLIB;
	}
	if (!err) {
		__syncthreads();
		flush_to(pos);
		// Adler-32 trailer, big-endian, after the bit reader is byte aligned
		getbits(br, br.cnt & 7);
		uint32_t want = 0;
		for (int k = 0; k < 4; k++) { refill(S, br); want = (want << 8) | getbits(br, 8); }
		if (consumed_bytes(br) > br.nbytes) err = CCT_ST_ZLIB;         // incomplete or truncated stream
		else if (want != ((adB << 16) | adA)) err = CCT_ST_ZLIB;        // incorrect data check
	}
	if (tid == 0) {
		a.out_sizes[s] = (err & CCT_ST_ZLIB) ? 0u : pos;
		a.status[s] = err;
		if (PROF) {
			prof[P_TOTAL] = (uint64_t)(clock64() - tstart);
			for (int i = 0; i < P_N; i++) prof_out[(size_t)s * P_N + i] = prof[i];
		}
	}
}

}  // namespace

template <class G>
static hipError_t launch_inflate_geo(const InflateArgs &a, int n, hipStream_t st)
{
	const size_t lds = sizeof(InfShared<G>);
	static const bool want_prof = getenv("CCT_INF_PROF") != nullptr;
	if (want_prof) {  // tuning runs only: synchronises and prints per-phase averages
		static const char *names[P_N] = {"match_setup", "hdr", "tables(rest)", "t_cllens", "t_buildcl", "t_clwalk", "t_ll", "stage", "walk0", "restart", "scan", "emit", "match", "flush", "TOTAL", "#blocks", "#rounds",
		                                 "#passes", "#matches", "#dependent", "#batches", "#lanes_used", "sum_round_max_gensteps", "sum_gensteps",
		                                 "sum_long_codes", "wave0_walk_cycles", "wave0_outer_steps"};
		uint64_t *d_prof = nullptr;
		hipError_t e = hipMalloc(&d_prof, (size_t)n * P_N * 8);
		if (e != hipSuccess) return e;
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel<G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL((inflate_kernel<G, true>), dim3(n), dim3(G::NT), lds, st, a, d_prof);
		std::vector<uint64_t> h((size_t)n * P_N);
		e = hipMemcpyAsync(h.data(), d_prof, h.size() * 8, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		(void)hipFree(d_prof);
		if (e != hipSuccess) return e;
		fprintf(stderr, "[inflate prof] n=%d lanes=%d, per stream:", n, G::NT);
		for (int i = 0; i < P_N; i++) {
			double sum = 0;
			for (int s = 0; s < n; s++) sum += (double)h[(size_t)s * P_N + i];
			fprintf(stderr, " %s=%.0f", names[i], sum / n);
		}
		fprintf(stderr, "\n");
		return hipSuccess;
	}
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel<G, false>),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL((inflate_kernel<G, false>), dim3(n), dim3(G::NT), lds, st, a, (uint64_t *)nullptr);
	return hipGetLastError();
}

hipError_t launch_inflate(const InflateArgs &a, int n, hipStream_t st, int lanes)
{
	return lanes >= 512 ? launch_inflate_geo<Geo<512>>(a, n, st) : launch_inflate_geo<Geo<256>>(a, n, st);
}

}  // namespace cct
