// Device INFLATE (RFC 1950/1951) for the decode side: zlib.decompress(file_bytes[13:]) of the reference
// (src/codec/core.py:421).  zlib is a third-party dependency of the reference; inflating is fully
// specified by the stream format, so any correct decoder returns the same bytes (and the Adler-32
// trailer is verified like zlib does).
//
// One workgroup of 4 waves per stream.  The format carries no symbol index, so the position of symbol k+1 is
// only known after symbol k has been decoded; a single lane walking that chain spends ~10^3 cycles per symbol.
// The workgroup breaks the chain speculatively (Huffman codes re-synchronise after a few symbols):
//   * a round covers 256 segments of SEG_BITS (256) compressed bits, one per lane; lane 0 starts at the true position,
//     the others at their segment boundary, and every lane decodes until it crosses into the next segment;
//   * lanes then restart from where their predecessor really landed until no start moves any more (lane k is
//     final after k passes at the latest; in practice 2-3 passes), which yields the true chain of the round;
//   * a prefix sum over output bytes gives every lane its output offset; lanes decode once more, writing
//     literals to the output ring and LZ77 copies to a list that is then resolved in stream order
//     (independent copies by one lane each, a dependent/overlapping one by the whole workgroup);
//   * block headers are parsed redundantly by every lane (workgroup-uniform control flow); the canonical code
//     tables of each block are built by wave 0 (two-level LSB-first lookup tables: 11-bit root for
//     literal/length, 10-bit for distance, sub-tables for longer codes, base value and extra-bit count packed
//     in the entry);
//   * compressed input is staged through LDS in 4 KiB chunks, output through a 64 KiB LDS ring that
//     is flushed to HBM as aligned 16-byte stores and Adler-summed on the way out.
// Slices are independent, so 256 streams occupy 256 CUs at once.
#include <hip/hip_runtime.h>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

constexpr int NT = 256;                       // lanes per stream
constexpr int NWV = NT / 64;                  // waves per stream
constexpr int INF_RING = 65536, INF_RMASK = INF_RING - 1, INF_FLUSH = 4096;
constexpr int INF_IN = 16384, INF_CHUNK = 4096;
constexpr int LL_BITS = 11, D_BITS = 10;
constexpr int LL_SUB = 1280, D_SUB = 256;     // a complete sub-tree of depth 4 has >= 5 leaves: <= 16/5 entries per long code
constexpr uint32_t K_LIT = 1u << 24, K_LEN = 2u << 24, K_EOB = 3u << 24;
constexpr int SEG_BITS = 256;                 // compressed bits per lane and round
constexpr int ROUND_OUT_BUDGET = 24576;       // output bytes of a round
constexpr int LANE_OUT_CAP = 16384;           // a lane stops (and ends the round) once it has produced this much
constexpr int MLIST_CAP = 2048;               // LZ77 copies a round may hold (one lane alone: at most SEG_BITS / 2)
constexpr uint32_t SEG_EOB = 1, SEG_BAD = 2, SEG_CUT = 4;

__constant__ uint16_t c_lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t c_lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint16_t c_dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__constant__ uint8_t c_dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t c_clorder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

constexpr int CL_BITS = 7;  // code-length codes are at most 7 bits long: one flat table

struct InfShared {
	uint8_t ring[INF_RING];
	alignas(16) uint8_t inbuf[INF_IN];
	// two-level decode tables (root LL_BITS / D_BITS, sub-tables for longer codes).  Entry: bits 0-3 code bits
	// to drop, 4-7 extra bits, 8-23 value (literal, base length, base distance), 24-25 kind; bit 31 = pointer to a
	// sub-table (bits 0-3 its index width, 8-23 its offset); 0 = no such code
	uint32_t ll_tab[1 << LL_BITS], d_tab[1 << D_BITS];
	uint32_t ll_sub[LL_SUB], d_sub[D_SUB];
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

__device__ __forceinline__ void stage_chunk(InfShared &S, BitReader &br)
{
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

__device__ __forceinline__ void refill(InfShared &S, BitReader &br)
{
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

// code-length alphabet (19 symbols, <= 7 bits) from S.lens[0..19), by thread 0; S.ok = 0 if over-subscribed
__device__ void build_cl(InfShared &S)
{
	for (int i = threadIdx.x; i < (1 << CL_BITS); i += NT) S.cl_tab[i] = 0;
	__syncthreads();
	if (threadIdx.x == 0) {
		int count[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ok = 1, left = 1;
		for (int s = 0; s < 19; s++) count[S.lens[s] & 7]++;  // 3-bit fields: lengths are 0..7
		uint32_t next[8], c = 0;
		next[0] = 0;
		for (int len = 1; len <= 7; len++) {
			left = (left << 1) - count[len];
			if (left < 0) ok = 0;
			c = (c + (len > 1 ? (uint32_t)count[len - 1] : 0u)) << 1;
			next[len] = c;
		}
		for (int s = 0; ok && s < 19; s++) {
			const int l = S.lens[s];
			if (!l) continue;
			const uint32_t r = __brev(next[l]++) >> (32 - l);
			for (uint32_t k = r; k < (1u << CL_BITS); k += (1u << l)) S.cl_tab[k] = (uint8_t)((s << 3) | l);
		}
		S.ok = ok;
	}
	__syncthreads();
}

// decode-table entry of symbol s whose remaining code bits are `bits`
__device__ __forceinline__ uint32_t ll_entry(const InfShared &S, int s, int bits)
{
	if (s < 256) return K_LIT | ((uint32_t)s << 8) | (uint32_t)bits;
	if (s == 256) return K_EOB | (uint32_t)bits;
	if (s - 257 >= 29) return 0;  // 286, 287: invalid
	return K_LEN | ((uint32_t)S.lbase[s - 257] << 8) | ((uint32_t)S.lext[s - 257] << 4) | (uint32_t)bits;
}
__device__ __forceinline__ uint32_t d_entry(const InfShared &S, int s, int bits)
{
	if (s >= 30) return 0;
	return K_LEN | ((uint32_t)S.dbase[s] << 8) | ((uint32_t)S.dext[s] << 4) | (uint32_t)bits;
}

// Two-level tables from S.lens[0..n), built by wave 0 (the other waves wait at the closing barrier);
// S.ok = 0: over-subscribed code or sub-table overflow.
template <bool DIST>
__device__ void build_tables(InfShared &S, int n)
{
	constexpr int FB = DIST ? D_BITS : LL_BITS, SUBCAP = DIST ? D_SUB : LL_SUB;
	uint32_t *tab = DIST ? S.d_tab : S.ll_tab, *sub = DIST ? S.d_sub : S.ll_sub;
	const int lane = threadIdx.x & 63;
	if (threadIdx.x < 16) S.cnt[threadIdx.x] = 0;
	for (int i = threadIdx.x; i < (1 << FB); i += NT) tab[i] = 0;
	for (int i = threadIdx.x; i < SUBCAP; i += NT) sub[i] = 0;
	__syncthreads();
	if (threadIdx.x < 64) {
		const uint64_t lt_mask = (1ull << lane) - 1ull;
		bool ok = true;
		for (int s = lane; s < n; s += 64) atomicAdd(&S.cnt[S.lens[s]], 1u);
		__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
		uint32_t next[16];  // canonical: first code of each length
		int left = 1;
		uint32_t c = 0;
		next[0] = 0;
#pragma unroll
		for (int len = 1; len <= 15; len++) {
			const int cn = (int)S.cnt[len];
			left = (left << 1) - cn;
			if (left < 0) ok = false;
			c = (c + (len > 1 ? S.cnt[len - 1] : 0u)) << 1;
			next[len] = c;
		}
		// code of every symbol = first code of its length + number of lower symbols of that length; LSB-first reversal
		for (int pass = 0; ok && pass < 2; pass++) {  // pass 0: sub-table widths of long codes; pass 1: entries
			uint32_t seen[16];
#pragma unroll
			for (int b = 1; b <= 15; b++) seen[b] = next[b];
			for (int s0 = 0; s0 < n; s0 += 64) {
				const int sidx = s0 + lane;
				const int l = sidx < n ? (int)S.lens[sidx] : 0;
				uint32_t code = 0;
#pragma unroll
				for (int b = 1; b <= 15; b++) {
					const uint64_t bal = __ballot(l == b);
					if (l == b) code = seen[b] + (uint32_t)__popcll(bal & lt_mask);
					seen[b] += (uint32_t)__popcll(bal);
				}
				const uint32_t r = l ? __brev(code) >> (32 - l) : 0u;
				const bool wide = pass == 1 && l != 0 && l <= FB - 6;  // 64 root entries or more
				if (l != 0) {
					if (pass == 0) {
						if (l > FB) atomicMax(&tab[r & ((1u << FB) - 1u)], (uint32_t)(l - FB));
					} else if (l <= FB) {
						// a code of l bits owns 2^(FB-l) root entries: up to 32 are written by the symbol's own lane, the
						// short codes (64 entries and more) by the whole wave below
						if (!wide) {
							const uint32_t e = DIST ? d_entry(S, sidx, l) : ll_entry(S, sidx, l);
							for (uint32_t k = r; k < (1u << FB); k += (1u << l)) tab[k] = e;
						}
					} else {
						const uint32_t p = tab[r & ((1u << FB) - 1u)];
						const uint32_t off = (p >> 8) & 0xFFFFu, kw = p & 15u;
						const uint32_t e = DIST ? d_entry(S, sidx, l - FB) : ll_entry(S, sidx, l - FB);
						for (uint32_t k = r >> FB; k < (1u << kw); k += (1u << (l - FB))) sub[off + k] = e;
					}
				}
				for (uint64_t wm = __ballot(wide); wm; wm &= wm - 1) {  // the short codes of this chunk, one at a time
					const int src = __ffsll((long long)wm) - 1;
					const int wl = __shfl(l, src, 64);
					const uint32_t wr = __shfl(r, src, 64);
					const int ws = s0 + src;
					const uint32_t e = DIST ? d_entry(S, ws, wl) : ll_entry(S, ws, wl);
					for (uint32_t k = wr + ((uint32_t)lane << wl); k < (1u << FB); k += (64u << wl)) tab[k] = e;
				}
			}
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			if (pass == 0) {  // allocate the sub-tables: every lane owns a contiguous stretch of the root table
				constexpr int PER = (1 << FB) / 64;
				uint32_t sum = 0;
				for (int i = 0; i < PER; i++) { const uint32_t kw = tab[lane * PER + i]; if (kw) sum += 1u << kw; }
				uint32_t inc = sum;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
				if (__shfl(inc, 63, 64) > (uint32_t)SUBCAP) ok = false;
				uint32_t off = inc - sum;
				for (int i = 0; ok && i < PER; i++) {
					const uint32_t kw = tab[lane * PER + i];
					if (kw) { tab[lane * PER + i] = 0x80000000u | (off << 8) | kw; off += 1u << kw; }
				}
				__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			}
		}
		if (lane == 0) S.ok = ok ? 1 : 0;
	}
	__syncthreads();
}

// per-lane bit reader over the staged input; bit positions are 32-bit offsets from the round's origin dword
struct LaneBits { uint64_t buf; int cnt; uint32_t next; };  // next = dword index relative to br.src

__device__ __forceinline__ void lane_init(const InfShared &S, LaneBits &lb, uint32_t org_dword, uint32_t rel_bit)
{
	const uint32_t *in32 = reinterpret_cast<const uint32_t *>(S.inbuf);
	const uint32_t idx = org_dword + (rel_bit >> 5), sh = rel_bit & 31u;
	const uint64_t w = (uint64_t)in32[idx & (INF_IN / 4 - 1)] | ((uint64_t)in32[(idx + 1) & (INF_IN / 4 - 1)] << 32);
	lb.buf = w >> sh; lb.cnt = 64 - (int)sh; lb.next = idx + 2;
}
__device__ __forceinline__ void lane_refill(const InfShared &S, LaneBits &lb)
{
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
template <bool EMIT>
__device__ __forceinline__ void walk_segment(InfShared &S, uint32_t org_dword, uint32_t start, uint32_t end, uint32_t &land,
                                             uint32_t &nbytes, uint32_t &nmatch, uint32_t &flags, uint32_t o, uint32_t mi)
{
	LaneBits lb;
	lane_init(S, lb, org_dword, start);
	nbytes = 0; nmatch = 0; flags = 0;
	while (lane_pos(lb, org_dword) < end) {
		if (nbytes >= (uint32_t)LANE_OUT_CAP) { flags = SEG_CUT; break; }  // keeps a round inside the output ring
		lane_refill(S, lb);
		uint32_t lo = (uint32_t)lb.buf;
		uint32_t e = S.ll_tab[lo & ((1u << LL_BITS) - 1u)];
		int used = 0;
		if (e >> 31) { used = LL_BITS; lo >>= LL_BITS; e = S.ll_sub[((e >> 8) & 0xFFFFu) + (lo & ((1u << (e & 15u)) - 1u))]; }
		if (e == 0) { flags = SEG_BAD; break; }
		const uint32_t cb = e & 15u, xb = (e >> 4) & 15u;
		const uint32_t kind = e & (3u << 24);
		const uint32_t val = ((e >> 8) & 0xFFFFu) + ((lo >> cb) & ((1u << xb) - 1u));  // code <= 15, extra <= 5 bits: inside lo
		used += (int)(cb + xb);
		lb.buf >>= used; lb.cnt -= used;
		if (kind == K_LIT) {
			if (EMIT) S.ring[(o + nbytes) & INF_RMASK] = (uint8_t)val;
			nbytes++;
		} else if (kind == K_EOB) {
			flags = SEG_EOB;
			break;
		} else {
			lane_refill(S, lb);
			uint32_t dlo = (uint32_t)lb.buf;
			uint32_t de = S.d_tab[dlo & ((1u << D_BITS) - 1u)];
			int dused = 0;
			if (de >> 31) { dused = D_BITS; dlo >>= D_BITS; de = S.d_sub[((de >> 8) & 0xFFFFu) + (dlo & ((1u << (de & 15u)) - 1u))]; }
			if (de == 0) { flags = SEG_BAD; break; }
			const uint32_t dcb = de & 15u, dxb = (de >> 4) & 15u;
			const uint32_t dist = ((de >> 8) & 0xFFFFu) + ((dlo >> dcb) & ((1u << dxb) - 1u));  // code <= 15, extra <= 13 bits
			dused += (int)(dcb + dxb);
			lb.buf >>= dused; lb.cnt -= dused;
			if (EMIT) {
				if (dist > o + nbytes) { flags = SEG_BAD; break; }  // distance too far back
				S.mlist[mi + nmatch] = make_uint2(o + nbytes, val | ((dist - 1u) << 9));
			}
			nbytes += val; nmatch++;
		}
	}
	land = lane_pos(lb, org_dword);
}

// The code-length sequence of a dynamic block header (<= 316 entries, run-length coded with the 19-symbol code)
// is the same kind of chain as the block body, so wave 0 decodes it the same way: 64 segments of CL_SEG bits,
// restarts until no start moves, prefix sum of the entry counts, second walk that writes S.lens.  A "repeat the
// previous length" symbol (16) writes markers that a scan resolves afterwards.
constexpr int CL_SEG = 64;
constexpr uint32_t CL_BAD = 1, CL_OVER = 2;
constexpr uint8_t CL_PREV = 0xFF;  // marker: same as the entry before (lengths are <= 15)

template <bool EMIT>
__device__ __forceinline__ void cl_walk(InfShared &S, uint32_t org_dword, uint32_t start, uint32_t end, uint32_t limit,
                                        uint32_t idx_base, uint32_t &land, uint32_t &cnt, uint32_t &flags)
{
	LaneBits lb;
	lane_init(S, lb, org_dword, start);
	cnt = 0; flags = 0;
	while (lane_pos(lb, org_dword) < end) {
		if (EMIT && cnt == limit) break;  // the sequence is complete: the block body starts here
		lane_refill(S, lb);
		const uint32_t lo = (uint32_t)lb.buf;
		const uint32_t ce = S.cl_tab[lo & ((1u << CL_BITS) - 1u)];
		if (ce == 0) { flags = CL_BAD; break; }
		const uint32_t sym = ce >> 3, cb = ce & 7u;
		const uint32_t xb = sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u;
		const uint32_t xv = (lo >> cb) & ((1u << xb) - 1u);
		const uint32_t rep = sym < 16 ? 1u : sym == 18 ? 11u + xv : 3u + xv;
		if (EMIT) {
			if (rep > limit - cnt) { flags = CL_OVER; break; }  // repeat beyond the last entry
			const uint8_t v = sym < 16 ? (uint8_t)sym : sym == 16 ? CL_PREV : (uint8_t)0;
			for (uint32_t t = 0; t < rep; t++) S.lens[idx_base + cnt + t] = v;
		}
		lb.buf >>= (cb + xb); lb.cnt -= (int)(cb + xb);
		cnt += rep;
	}
	land = lane_pos(lb, org_dword);
}

// smallest lane index (0..255) whose predicate is set, NT if none; all lanes call it (one barrier inside, and
// what was written to LDS before the call is visible to everybody after it)
__device__ __forceinline__ int first_lane_with(InfShared &S, bool pred, int slot)
{
	const uint64_t bal = __ballot(pred);
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) S.wred[slot * NWV + wave] = bal ? (uint32_t)(wave * 64 + __ffsll((long long)bal) - 1) : (uint32_t)NT;
	__syncthreads();
	uint32_t r = (uint32_t)NT;
#pragma unroll
	for (int w = 0; w < NWV; w++) r = min(r, S.wred[slot * NWV + w]);
	return (int)r;
}

__global__ void __launch_bounds__(NT) inflate_kernel(InflateArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
	InfShared &S = *reinterpret_cast<InfShared *>(smem_raw);
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
	while (!err && !last) {
		refill(S, br);
		last = getbits(br, 1) != 0;
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
			for (int i = 0; i < ncode; i++) {
				refill(S, br);
				const uint32_t v = getbits(br, 3);
				if (tid == 0) S.lens[c_clorder[i]] = (uint8_t)v;
			}
			__syncthreads();
			build_cl(S);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
			// the code lengths of both alphabets, run-length coded; S.lens is reused after cl is built
			{
				while (br.bytepos + 2048 > br.staged_end) stage_chunk(S, br);  // the sequence is < 700 bytes
				const uint32_t total = (uint32_t)(nlen + ndist);
				const uint64_t b0 = br.bytepos * 8u - (uint64_t)br.cnt;
				if (wave == 0) {  // the other waves wait at the barrier below
					uint32_t idx = 0, lerr = 0;
					uint64_t bp = b0;
					while (idx < total && !lerr) {
						const uint32_t org_dword = (uint32_t)(bp >> 5), org_bit = (uint32_t)bp & 31u;
						const uint32_t nominal = org_bit + (uint32_t)lane * CL_SEG, seg_end = nominal + CL_SEG;
						uint32_t start = nominal, land, cnt, fl;
						cl_walk<false>(S, org_dword, start, seg_end, 0, 0, land, cnt, fl);
						int first;
						for (;;) {  // restart from where the previous lane really landed until nothing moves
							const uint32_t pl = __shfl_up(land, 1, 64);
							const uint64_t flagged = __ballot(fl != 0);
							first = flagged ? __ffsll((long long)flagged) - 1 : 64;
							const bool moved = lane > 0 && lane <= first && pl != start;
							if (!__ballot(moved)) break;
							if (moved) { start = pl; cl_walk<false>(S, org_dword, start, seg_end, 0, 0, land, cnt, fl); }
						}
						uint32_t inc = cnt;  // entries up to and including this lane
#pragma unroll
						for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
						const uint32_t remaining = total - idx, base = inc - cnt;
						const bool mine = lane <= first && base < remaining;
						uint32_t eland = land, ecnt = 0, efl = 0;
						if (mine) cl_walk<true>(S, org_dword, start, seg_end, remaining - base, idx + base, eland, ecnt, efl);
						if (__ballot(mine && efl != 0)) { lerr = 1; break; }
						const uint64_t done = __ballot(mine && base + ecnt >= remaining);  // the lane that wrote the last entry
						if (done) {
							const int le = __ffsll((long long)done) - 1;
							idx = total;
							bp = (uint64_t)org_dword * 32u + __shfl(eland, le, 64);
						} else {  // 64 segments were not enough (or an invalid code came first, caught above)
							if (first < 64) { lerr = 1; break; }
							idx += __shfl(inc, 63, 64);
							bp = (uint64_t)org_dword * 32u + __shfl(land, 63, 64);
						}
					}
					__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
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
			if (S.lens[256] == 0) { err = CCT_ST_ZLIB; break; }  // no end-of-block code
			// distance lengths follow the literal/length lengths: move them to their own array slot first
			uint8_t dl = 0;
			if (tid < ndist) dl = S.lens[nlen + tid];
			__syncthreads();
			build_tables<false>(S, nlen);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
			if (tid < 32) S.lens[tid] = tid < ndist ? dl : 0;
			__syncthreads();
			build_tables<true>(S, ndist);
			if (!S.ok) { err = CCT_ST_ZLIB; break; }
		}
		// ---- symbols of this block, in speculative rounds (see the header comment)
		uint64_t bitpos = br.bytepos * 8u - (uint64_t)br.cnt;  // true position of the next symbol
		for (bool block_done = false; !block_done && !err;) {
			__syncthreads();
			if (pos - flushed >= (uint32_t)INF_FLUSH) flush_to(pos & ~(uint32_t)(INF_FLUSH - 1));
			if ((bitpos >> 3) > br.nbytes + 16) { err |= CCT_ST_ZLIB; break; }  // decoding zeros past the end
			const uint64_t need_end = ((bitpos + (uint64_t)NT * SEG_BITS) >> 3) + 32;
			while (need_end > br.staged_end) stage_chunk(S, br);
			const uint32_t org_dword = (uint32_t)(bitpos >> 5);      // lanes address bits relative to this dword
			const uint32_t org_bit = (uint32_t)bitpos & 31u;
			const uint32_t seg_end = org_bit + (uint32_t)(tid + 1) * SEG_BITS;
			uint32_t start = org_bit + (uint32_t)tid * SEG_BITS, land, nb, nm, fl;
			walk_segment<false>(S, org_dword, start, seg_end, land, nb, nm, fl, 0, 0);
			for (;;) {  // restart from where the previous lane really landed until nothing moves
				__syncthreads();
				S.land[tid] = land;
				const int first = first_lane_with(S, fl != 0, 0);
				const uint32_t pl = tid ? S.land[tid - 1] : start;
				const bool moved = tid > 0 && tid <= first && pl != start;
				if (!__syncthreads_or(moved)) break;
				if (moved) { start = pl; walk_segment<false>(S, org_dword, start, seg_end, land, nb, nm, fl, 0, 0); }
			}
			// the chain is true up to the first lane that saw end-of-block or an invalid code; cut the round
			// where the ring or the copy list would overflow (lane 0 always fits)
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
			if (mine) {
				uint32_t l2, b2, m2;
				walk_segment<true>(S, org_dword, start, seg_end, l2, b2, m2, efl, o, mi);
			}
			if (tid == lastl) { S.rres[0] = cb; S.rres[1] = cm; S.rres[2] = land; S.rres[3] = fl; }
			if (__syncthreads_or(mine && (efl & SEG_BAD))) { err |= CCT_ST_ZLIB; break; }
			const uint32_t round_bytes = S.rres[0], nmatch_round = S.rres[1], round_land = S.rres[2], lfl = S.rres[3];
			// LZ77 copies in stream order: a copy whose source ends before the first unresolved destination
			// does not depend on the others of its batch and is done by one lane; a dependent one by everybody
			for (uint32_t k0 = 0; k0 < nmatch_round;) {
				const uint32_t k = k0 + (uint32_t)tid;
				uint2 m = make_uint2(0, 0);
				if (k < nmatch_round) m = S.mlist[k];
				const uint32_t len = m.y & 511u, dist = (m.y >> 9) + 1u;
				const uint32_t first_dst = S.mlist[k0].x;
				const bool indep = k < nmatch_round && len <= 16u && m.x - dist + len <= first_dst;  // long copies: everybody
				const int nind = first_lane_with(S, !indep, 1);  // leading independent copies
				if (tid < nind) {
					for (uint32_t i = 0; i < len; i++) S.ring[(m.x + i) & INF_RMASK] = S.ring[(m.x - dist + i) & INF_RMASK];
				}
				__syncthreads();
				k0 += (uint32_t)nind;
				if (nind < NT && k0 < nmatch_round) {  // the next one depends on something just written
					const uint2 dm = S.mlist[k0];
					const uint32_t dlen = dm.y & 511u, ddist = (dm.y >> 9) + 1u;
					if (ddist >= dlen) {
						for (uint32_t i = tid; i < dlen; i += NT) S.ring[(dm.x + i) & INF_RMASK] = S.ring[(dm.x - ddist + i) & INF_RMASK];
					} else {  // overlapping: the output repeats its first ddist bytes, which are already in place
						for (uint32_t i = tid; i < dlen; i += NT) S.ring[(dm.x + i) & INF_RMASK] = S.ring[(dm.x - ddist + i % ddist) & INF_RMASK];
					}
					__syncthreads();
					k0++;
				}
			}
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
	}
}

}  // namespace

hipError_t launch_inflate(const InflateArgs &a, int n, hipStream_t st)
{
	const size_t lds = sizeof(InfShared);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(inflate_kernel, dim3(n), dim3(NT), lds, st, a);
	return hipGetLastError();
}

}  // namespace cct
