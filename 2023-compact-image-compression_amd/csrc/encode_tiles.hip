// CompaCT encode, stage (i), tile-staged fast path for block_size 16 on shapes whose traversal
// decomposes into aligned 64x64 tiles (every power-of-two square: 4096 consecutive positions of
// the generalized Hilbert curve are one aligned 64x64 square; csrc/api.cpp verifies this from
// the table instead of assuming it).
//
// One workgroup (1024 lanes) per slice, two streaming passes, no inter-workgroup traffic.
// Unit of work = a "super-tile": two consecutive tiles = 8192 traversal positions = 16 KiB.
//
//   HBM -> VGPR   every lane loads ONE 16-byte row chunk per super-tile (128-B row segments, fully
//                 coalesced); four super-tiles stay in flight in four named registers.
//   VGPR -> LDS   ds_write_b128 into a raster image of the tile whose 16-B column chunks are
//                 XOR-swizzled by row, so that the later 2-byte gathers spread over the banks.
//   LDS gather    lane = half a block (8 pixels): one ds_read_b128 of the pattern table (LDS byte
//                 offset of each traversal position inside a tile, swizzle included), eight
//                 ds_read_u16; the 8 pixels stay in registers and also go back to LDS in traversal
//                 order (one ds_write_b128) for the look-ahead consumers.
//   pass 1        |delta| > 64 flags per half block, pair-reduced with one shuffle; difficult blocks
//                 (cluster.py:51-59) are compacted in order with ballots; one WAVE per difficult
//                 block evaluates its 63 mesh candidates (cluster.py:122-158), one candidate per
//                 lane, with packed 16-bit compares, and ballots the fit mask.
//   resolve       greedy first fit (cluster.py:79-190) per island of difficult blocks, one lane each.
//   pass 2        token sizes per half block -> wave scan -> byte staging -> aligned 16-byte stores.
//
// The loops are software-pipelined so that one iteration needs two workgroup barriers:
//   pass 1:  P1 gather(s) | P2 analyse(s), append(s-1), masks(s-2), stage(s+1) -> LDS, load(s+5)
//   pass 2:  P1 gather(s), write tokens(s-2) | P2 size tokens(s-1), flush(s-2), stage(s+1), load(s+5)
// Barriers are `s_waitcnt lgkmcnt(0); s_barrier` (LDS only), so the global loads stay in flight
// across them; __syncthreads() is used only where HBM-resident lists/roles must be published.
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// LDS pointers carry their address space in the type: a generic pointer would turn every access
// into a flat_* instruction, whose completion forces vmcnt(0) and drains the prefetch ring.
#define LDS(T) __attribute__((address_space(3))) T

constexpr int STP = 8192;                 // pixels per super-tile
constexpr int TPX = 4096;                 // pixels per tile
constexpr int TT = 1024;                  // threads
constexpr int NW = TT / 64;

// LDS carve (bytes)
constexpr int L_RASTER = 0;                       // 2 x 16384
constexpr int L_DLIN = 32768;                     // 3 x 16384 (traversal-ordered pixels)
constexpr int L_PAT = L_DLIN + 3 * 16384;         // 4 x 8192 pattern tables
constexpr int L_LIST = L_PAT + 4 * 8192;          // 20480: difficult list (pass 1) / token staging (pass 2)
constexpr int L_MISC = L_LIST + 20480;
constexpr int MISC_WCNT = 0;                      // [2][16] u32
constexpr int MISC_WTOT = 128;                    // [2][16] u32
constexpr int MISC_CTR = 256;                     // status, n_full, n_jump
constexpr int MISC_TORG = 320;                    // u32[TILE_MAX_TILES]
constexpr int MISC_TORI = MISC_TORG + 4 * TILE_MAX_TILES;
constexpr int L_ROLE = L_MISC + ((MISC_TORI + TILE_MAX_TILES + 15) & ~15);

__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// cross-lane moves as DPP modifiers (VALU latency) instead of ds_bpermute (LDS latency)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)  // lanes without a source read 0
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1 across the whole wave
constexpr int DPP_QUAD_SWAP1 = 0xB1;   // quad_perm [1,0,3,2]: lane i <- lane i^1
constexpr int DPP_QUAD_EVEN = 0xA0;    // quad_perm [0,0,2,2]: lane i <- lane i&~1

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	v += dpp0<0x111>(v);  // row_shr:1
	v += dpp0<0x112>(v);  // row_shr:2
	v += dpp0<0x114>(v);  // row_shr:4
	v += dpp0<0x118>(v);  // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
	return v;
}

// base / total of 16 per-wave counts held in LDS: lanes 0..15 scan them, everyone reads the result
__device__ __forceinline__ void wave_counts_prefix(const LDS(uint32_t) *cnt16, int lane, int wave_s, uint32_t &base, uint32_t &tot)
{
	uint32_t v = (lane < NW) ? cnt16[lane] : 0u;
	v += dpp0<0x111>(v);
	v += dpp0<0x112>(v);
	v += dpp0<0x114>(v);
	v += dpp0<0x118>(v);
	tot = (uint32_t)__builtin_amdgcn_readlane((int)v, NW - 1);
	const uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)v, wave_s > 0 ? wave_s - 1 : 0);
	base = wave_s > 0 ? below : 0u;
}

struct DiffListT {
	LDS(uint32_t) *l_idx; LDS(uint8_t) *l_cur; LDS(uint64_t) *l_mask;
	uint32_t *g_idx; uint8_t *g_cur; uint64_t *g_mask;
	__device__ __forceinline__ void set(uint32_t e, uint32_t idx, uint32_t cur) const
	{
		if (e < ENC_LIST_CAP) { l_idx[e] = idx; l_cur[e] = (uint8_t)cur; }
		else { g_idx[e - ENC_LIST_CAP] = idx; g_cur[e - ENC_LIST_CAP] = (uint8_t)cur; }
	}
	__device__ __forceinline__ uint32_t idx(uint32_t e) const
	{
		uint32_t v;
		if (e < ENC_LIST_CAP) v = l_idx[e]; else v = g_idx[e - ENC_LIST_CAP];
		return v;
	}
	__device__ __forceinline__ uint32_t cur(uint32_t e) const
	{
		uint32_t v;
		if (e < ENC_LIST_CAP) v = l_cur[e]; else v = g_cur[e - ENC_LIST_CAP];
		return v;
	}
	__device__ __forceinline__ uint64_t mask(uint32_t e) const
	{
		uint64_t v;
		if (e < ENC_LIST_CAP) v = l_mask[e]; else v = g_mask[e - ENC_LIST_CAP];
		return v;
	}
	__device__ __forceinline__ void set_mask(uint32_t e, uint64_t m) const
	{
		if (e < ENC_LIST_CAP) l_mask[e] = m; else g_mask[e - ENC_LIST_CAP] = m;
	}
};

// pixel i (0..7) of a packed half block
__device__ __forceinline__ int px_of(const u32x4 &v, int i)
{
	const uint32_t w = (i < 2) ? v.x : (i < 4) ? v.y : (i < 6) ? v.z : v.w;
	return (int)((i & 1) ? (w >> 16) : (w & 0xFFFFu));
}

// packed 16-bit helpers (VOP3P): per-half a-b, logical >>15, add
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ uint32_t pk_neg_count(uint32_t acc, uint32_t x)
{
	// acc.halves += (x.halves < 0)
	// the shift count must come from a register holding 15 in BOTH halves: an inline constant
	// only reaches the low half of a packed operand
	uint32_t s;
	const uint32_t fifteen = 0x000F000Fu;
	asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(s) : "v"(fifteen), "v"(x));
	uint32_t r;
	asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(acc), "v"(s));
	return r;
}

template <bool SGN, bool ROLE_LDS>
__global__ void __launch_bounds__(TT) encode_tiles_kernel(TileEncArgs ta)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const EncArgs &a = ta.e;
	LDS(uint8_t) *raster = (LDS(uint8_t) *)(smem + L_RASTER);
	LDS(uint16_t) *dlin = (LDS(uint16_t) *)(smem + L_DLIN);
	LDS(uint8_t) *pat = (LDS(uint8_t) *)(smem + L_PAT);
	LDS(uint8_t) *stg = (LDS(uint8_t) *)(smem + L_LIST);
	LDS(uint64_t) *l_mask = (LDS(uint64_t) *)(smem + L_LIST);
	LDS(uint32_t) *l_idx = (LDS(uint32_t) *)(smem + L_LIST + ENC_LIST_CAP * 8);
	LDS(uint8_t) *l_cur = (LDS(uint8_t) *)(smem + L_LIST + ENC_LIST_CAP * 12);
	LDS(uint32_t) *wcnt = (LDS(uint32_t) *)(smem + L_MISC + MISC_WCNT);
	LDS(uint32_t) *wtot = (LDS(uint32_t) *)(smem + L_MISC + MISC_WTOT);
	LDS(uint32_t) *ctr = (LDS(uint32_t) *)(smem + L_MISC + MISC_CTR);
	LDS(uint32_t) *torg = (LDS(uint32_t) *)(smem + L_MISC + MISC_TORG);
	LDS(uint8_t) *tori = (LDS(uint8_t) *)(smem + L_MISC + MISC_TORI);
	LDS(uint8_t) *role_l = (LDS(uint8_t) *)(smem + L_ROLE);

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wave_s = __builtin_amdgcn_readfirstlane(wave);
	const int hb = tid & 1;  // which half of its block this lane holds
	const int sl = blockIdx.x;
	const int N = a.N, NB = a.NB;
	const int NS = N / STP;
	const uint32_t dbg = a.dbg_skip;
	const bool seg = (a.flags & CCT_FLAG_SEGMENTATION) != 0 && !(dbg & 1u);
	const uint16_t *img = a.images + (size_t)sl * N;
	uint8_t *role_g = ROLE_LDS ? nullptr : a.ws_role + (size_t)sl * NB;
	auto role_get = [&](int b) -> int { return ROLE_LDS ? (int)role_l[b] : (int)role_g[b]; };
	auto role_set = [&](int b, int v) { if (ROLE_LDS) role_l[b] = (uint8_t)v; else role_g[b] = (uint8_t)v; };
	DiffListT dl{l_idx, l_cur, l_mask,
	             a.ws_lidx + (size_t)sl * NB, a.ws_lcur + (size_t)sl * NB, a.ws_lmask + (size_t)sl * NB};

	// ---- one-time LDS tables: patterns, tile origins / orientations
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(ta.patterns);
		LDS(u32x4) *dst = (LDS(u32x4) *)(smem + L_PAT);
		const int n16 = ta.n_orient * (TPX * 2 / 16);
		for (int i = tid; i < n16; i += TT) dst[i] = src[i];
		for (int i = tid; i < ta.n_tiles; i += TT) { torg[i] = ta.tile_org[i]; tori[i] = ta.tile_orient[i]; }
		if (tid < 4) ctr[tid] = 0;
	}
	__syncthreads();

	// lane's share of a super-tile load: tile j = tid>>9, 16-byte chunk c = tid&511 of that tile
	const int ld_j = tid >> 9, ld_c = tid & 511;
	const int ld_row = ld_c >> 3, ld_col = (ld_c & 7) ^ (ld_row & 7);  // XOR swizzle on the SOURCE column
	const uint16_t *ld_base = img + (size_t)ld_row * ta.row_pitch + (size_t)ld_col * 8;
	auto load_st = [&](int s) -> u32x4 {
		return *reinterpret_cast<const u32x4 *>(ld_base + torg[2 * s + ld_j]);
	};
	auto stage_to_lds = [&](int s, const u32x4 &v) {
		*(LDS(u32x4) *)(raster + (s & 1) * 16384 + tid * 16) = v;
	};
	// gather this lane's 8 traversal-consecutive pixels of super-tile s; also store them linearly
	const int g_j = tid >> 9;                 // tile inside the super-tile
	const int g_k = (tid * 8) & (TPX - 1);    // first traversal position inside that tile
	auto gather = [&](int s, int slot) -> u32x4 {
		LDS(uint8_t) *rt = raster + (s & 1) * 16384 + g_j * 8192;
		const u32x4 pe = *(LDS(u32x4) *)(pat + (int)tori[2 * s + g_j] * 8192 + g_k * 2);
		const uint32_t p0 = *(LDS(uint16_t) *)(rt + (pe.x & 0xFFFFu));
		const uint32_t p1 = *(LDS(uint16_t) *)(rt + (pe.x >> 16));
		const uint32_t p2 = *(LDS(uint16_t) *)(rt + (pe.y & 0xFFFFu));
		const uint32_t p3 = *(LDS(uint16_t) *)(rt + (pe.y >> 16));
		const uint32_t p4 = *(LDS(uint16_t) *)(rt + (pe.z & 0xFFFFu));
		const uint32_t p5 = *(LDS(uint16_t) *)(rt + (pe.z >> 16));
		const uint32_t p6 = *(LDS(uint16_t) *)(rt + (pe.w & 0xFFFFu));
		const uint32_t p7 = *(LDS(uint16_t) *)(rt + (pe.w >> 16));
		u32x4 out;
		out.x = p0 | (p1 << 16); out.y = p2 | (p3 << 16);
		out.z = p4 | (p5 << 16); out.w = p6 | (p7 << 16);
		*(LDS(u32x4) *)(dlin + slot * STP + tid * 8) = out;
		return out;
	};
	auto SX = [&](int v) -> int { return SGN ? (int)(int16_t)(uint16_t)v : v; };
	auto large = [&](int d) -> uint32_t { return ((uint32_t)(d + 64) > 128u) ? 1u : 0u; };  // |d| > 64

	uint32_t ndiff = 0;
	// ================================================================== pass 1
	if (seg) {
		u32x4 r0 = load_st(0), r1 = r0, r2 = r0, r3 = r0;
		if (1 < NS) r1 = load_st(1);
		if (2 < NS) r2 = load_st(2);
		if (3 < NS) r3 = load_st(3);
		stage_to_lds(0, r0);
		if (4 < NS) r0 = load_st(4);

		uint32_t p_diff = 0, p_cur = 0, p_rank = 0;  // this lane's block of super-tile s-1
		uint32_t start_prev = 0;                      // first list record of super-tile s-2
		bool heavy = false;                           // records spilled to HBM: use full barriers

		auto p1_iter = [&](const int s, u32x4 &rnext) __attribute__((always_inline)) {
			const int m = s % 3;                     // ring slot of super-tile s
			const int m1 = (m + 2) % 3;              // slot of s-1
			const int m2 = (m + 1) % 3;              // slot of s-2
			if (heavy) __syncthreads(); else lds_barrier();
			u32x4 own = {0, 0, 0, 0};
			if (s < NS && !(dbg & 8u)) own = gather(s, m);
			if (heavy) __syncthreads(); else lds_barrier();
			// ---- (a) analyse super-tile s (cluster.py:30-59)
			uint32_t c_diff = 0, c_cur = 0;
			if (s < NS && !(dbg & 16u)) {
				int px[8];
#pragma unroll
				for (int i = 0; i < 8; i++) px[i] = SX(px_of(own, i));
				int prev = (int)dpp0<DPP_WAVE_SHR1>((uint32_t)px[7]);
				if (lane == 0) {
					if (tid > 0) prev = SX((int)dlin[m * STP + tid * 8 - 1]);
					else if (s > 0) prev = SX((int)dlin[m1 * STP + STP - 1]);
					else prev = px[0];
				}
				const uint32_t f0 = large(px[0] - prev);
				uint32_t cnt = 0;
#pragma unroll
				for (int i = 1; i < 8; i++) cnt += large(px[i] - px[i - 1]);
				if (hb) cnt += f0;  // the transition between the two halves is an inner one
				const uint32_t chg = cnt + dpp0<DPP_QUAD_SWAP1>(cnt);
				const uint32_t enter = dpp0<DPP_QUAD_EVEN>(f0);
				c_diff = (chg >= 8u && hb == 0) ? 1u : 0u;                    // cluster.py:58
				c_cur = chg + ((s > 0 || tid > 0) ? enter : 0u);              // cluster.py:110 (block 0: Q4)
				if (hb == 0) role_set(s * 512 + (tid >> 1), 0);
			}
			const uint64_t bal = __ballot(c_diff != 0);
			const uint32_t c_rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
			if (lane == 0) wcnt[(s & 1) * 16 + wave] = (uint32_t)__popcll(bal);
			// ---- (b) append records of super-tile s-1 (counts published one iteration ago)
			const uint32_t start_cur = ndiff;
			if (s >= 1 && s <= NS) {
				uint32_t base, tot;
				wave_counts_prefix(wcnt + ((s - 1) & 1) * 16, lane, wave_s, base, tot);
				if (p_diff) dl.set(ndiff + base + p_rank, (uint32_t)((s - 1) * 512 + (tid >> 1)), p_cur);
				ndiff += tot;
				if (ndiff > ENC_LIST_CAP) heavy = true;
			}
			// ---- (c) candidate masks of super-tile s-2: its look-ahead lies in s-2 and s-1
			if (s >= 2 && !(dbg & 2u)) {
				for (uint32_t e = start_prev + wave; e < start_cur; e += NW) {
					const int i = (int)dl.idx(e);
					const uint32_t cur = dl.cur(e);
					const int p = i + lane;
					const bool valid = lane >= 1 && p < NB;
					// block i lives in s-2; candidate p in s-2 or s-1
					LDS(uint16_t) *ap = dlin + m2 * STP + ((i * 16) & (STP - 1));
					const int pk = valid ? p * 16 : i * 16;
					LDS(uint16_t) *bp = dlin + (((pk >> 13) == s - 2) ? m2 : m1) * STP + (pk & (STP - 1));
					const u32x4 a0 = *(LDS(u32x4) *)(ap), a1 = *(LDS(u32x4) *)(ap + 8);
					const u32x4 b0 = *(LDS(u32x4) *)(bp), b1 = *(LDS(u32x4) *)(bp + 8);
					const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
					const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
					uint32_t hi_or = 0;
#pragma unroll
					for (int q = 0; q < 8; q++) hi_or |= aw[q] | bw[q];
					const bool small = !SGN && !__any((hi_or & 0xC000C000u) != 0);
					uint32_t up;
					if (small) {
						// up = #(B[t] - A[t] >= 65) + #(A[t+1] - B[t] >= 65), all values < 16384 so the
						// packed 16-bit differences cannot overflow.  Count the NEGATIVE results instead.
						uint32_t neg = 0;
#pragma unroll
						for (int q = 0; q < 8; q++) {
							const uint32_t lo = aw[q] + 0x00410041u;                       // A[t] + 65 (no carry: < 16384)
							neg = pk_neg_count(neg, pk_sub(bw[q], lo));                    // B[t] - (A[t]+65) < 0
							// A[t+1] pairs: (A[2q+1], A[2q+2]); the last pair's high half has no successor
							const uint32_t an = (q < 7) ? __builtin_amdgcn_alignbit(aw[q + 1], aw[q], 16) : (aw[7] >> 16);
							uint32_t hi = pk_sub(an, 0x00410041u);                         // A[t+1] - 65
							if (q == 7) hi |= 0xFFFF0000u;  // t = 15 has no successor: -1 - B[15] is always negative
							neg = pk_neg_count(neg, pk_sub(hi, bw[q]));                    // (A[t+1]-65) - B[t] < 0
						}
						up = 32u - ((neg & 0xFFFFu) + (neg >> 16));
					} else {
						up = 0;
						int bprev = 0;
#pragma unroll
						for (int t = 0; t < 16; t++) {
							const int av = SX((int)((aw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu));
							const int bv = SX((int)((bw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu));
							if (t > 0) up += (av - bprev >= 65) ? 1u : 0u;
							up += (bv - av >= 65) ? 1u : 0u;
							bprev = bv;
						}
					}
					// cluster.py:153,158: up + 1 < current_delta - 2 (uint32); block 0 wraps: always fits (Q4)
					const bool fit = valid && ((i == 0) ? true : ((up + 1u) < (cur - 2u)));
					const uint64_t mk = __ballot(fit);
					if (lane == 0) dl.set_mask(e, mk);
				}
			}
			start_prev = start_cur;
			p_diff = c_diff; p_cur = c_cur; p_rank = c_rank;
			// ---- (d) next super-tile: registers -> LDS raster image, refill the register
			if (s + 1 < NS) stage_to_lds(s + 1, rnext);
			if (s + 5 < NS) rnext = load_st(s + 5);
		};

		for (int sb = 0; sb < NS + 2; sb += 4) {
			p1_iter(sb + 0, r1);
			p1_iter(sb + 1, r2);
			p1_iter(sb + 2, r3);
			p1_iter(sb + 3, r0);
		}
		__syncthreads();

		// -------------------------------------------------------------- resolve
		for (uint32_t e0 = tid; e0 < ndiff; e0 += TT) {
			const uint32_t i0 = dl.idx(e0);
			if (e0 > 0 && i0 - dl.idx(e0 - 1) <= 63u) continue;  // not the head of an island
			uint64_t cw = 0;
			uint32_t base = i0, e = e0, i = i0;
			for (;;) {
				const uint32_t sh = i - base;
				cw = (sh >= 64u) ? 0ull : (cw >> sh);
				base = i;
				if (!(cw & 1ull)) {
					const uint64_t avail = dl.mask(e) & ~cw & ~1ull;
					if (avail) {
						const int j = __ffsll((long long)avail) - 1;
						role_set((int)i, j);
						role_set((int)i + j, ROLE_PARTNER);
						cw |= 1ull << j;
					}
				}
				if (++e >= ndiff) break;
				const uint32_t inext = dl.idx(e);
				if (inext - i > 63u) break;
				i = inext;
			}
		}
		__syncthreads();
	}

	// ================================================================== pass 2
	uint8_t *out = a.payload + (size_t)sl * a.stride;
	uint32_t out_pos = 0, carry = 0;
	bool cap_hit = false;
	uint32_t my_full = 0, my_jump = 0;
	bool my_q7 = false;
	if (!(dbg & 4u)) {
		u32x4 r0 = load_st(0), r1 = r0, r2 = r0, r3 = r0;
		if (1 < NS) r1 = load_st(1);
		if (2 < NS) r2 = load_st(2);
		if (3 < NS) r3 = load_st(3);
		stage_to_lds(0, r0);
		if (4 < NS) r0 = load_st(4);

		// token state of this lane's half block of super-tile s-1, carried from P2 to the next P1
		u32x4 g_own = {0, 0, 0, 0};  // pixels gathered in this iteration (super-tile s)
		u32x4 t_own = {0, 0, 0, 0}, t_par = {0, 0, 0, 0};
		int t_prev = 0, t_r = 0;
		uint32_t t_nbytes = 0, t_excl = 0;
		uint32_t keep_byte = 0;  // remainder byte carried from a flush to the next token write

		// one token: first byte always stored, second byte only for a full delta (core.py:316-323)
		auto put = [&](LDS(uint8_t) *&w, int d) __attribute__((always_inline)) {
			const bool two = (uint32_t)(d + 63) > 127u;  // d < -63 || d > 64
			w[0] = two ? (uint8_t)(0xE0 | ((d >> 8) & 0x0F)) : (uint8_t)(d & 0x7F);
			if (two) w[1] = (uint8_t)(d & 0xFF);
			w += two ? 2 : 1;
		};
		auto cnt2 = [&](int d, uint32_t &n2) __attribute__((always_inline)) {
			const bool two = (uint32_t)(d + 63) > 127u;
			n2 += two ? 1u : 0u;
			if (two) my_q7 |= (uint32_t)(d + 2047) > 4095u;  // outside [-2047, 2048] (Q7)
		};

		auto p2_iter = [&](const int s, u32x4 &rnext) __attribute__((always_inline)) {
			const int m = s % 3;
			const int m1 = (m + 2) % 3;
			const int m2 = (m + 1) % 3;
			lds_barrier();
			const u32x4 prev_own = g_own;  // pixels of super-tile s-1, gathered one iteration ago
			if (s < NS && !(dbg & 8u)) g_own = gather(s, m);
			// ---- write the tokens of super-tile s-2 (sized one iteration ago)
			if (s >= 2 && s <= NS + 1 && !(dbg & 32u)) {
				if ((uint32_t)tid < carry) stg[tid] = (uint8_t)keep_byte;
				uint32_t wbase, wsum;
				wave_counts_prefix(wtot + ((s - 2) & 1) * 16, lane, wave_s, wbase, wsum);
				if (t_nbytes) {
					LDS(uint8_t) *w = stg + carry + wbase + t_excl;
					int prev = t_prev;
					if (t_r == 0) {
#pragma unroll
						for (int i = 0; i < 8; i++) {
							const int v = px_of(t_own, i);
							put(w, v - prev);
							prev = v;
						}
					} else {
						if (hb == 0) *w++ = (uint8_t)(0x80 | t_r);  // jump tag, core.py:290-294
#pragma unroll
						for (int i = 0; i < 8; i++) {
							const int va = px_of(t_own, i), vb = px_of(t_par, i);
							put(w, va - prev);
							put(w, vb - va);
							prev = vb;
						}
					}
				}
				if (s == NS + 1) {  // last flush: stage the EOF byte (core.py:329-330) and zero the pad
					uint32_t total = carry + wsum;
					if (a.eof >= 0) { if (tid == 0) stg[total] = (uint8_t)a.eof; total += 1; }
					if (tid >= 1 && tid <= 15) stg[total + tid - 1] = 0;
				}
			}
			lds_barrier();
			// ---- flush super-tile s-2
			if (s >= 2 && s <= NS + 1) {
				uint32_t wbase, bytes;
				wave_counts_prefix(wtot + ((s - 2) & 1) * 16, lane, wave_s, wbase, bytes);
				const bool last = (s == NS + 1);
				uint32_t total = carry + bytes;
				if (last && a.eof >= 0) total += 1;
				const uint32_t nflush = last ? ((total + 15u) & ~15u) : (total & ~15u);
				if ((size_t)out_pos + nflush > a.stride) cap_hit = true;
				if (!cap_hit && !(dbg & 128u)) {
					LDS(u32x4) *src = (LDS(u32x4) *)stg;
					u32x4 *dst = reinterpret_cast<u32x4 *>(out + out_pos);
					for (uint32_t u = tid; u < nflush / 16u; u += TT) dst[u] = src[u];
				}
				const uint32_t rem = last ? 0u : (total - nflush);
				if ((uint32_t)tid < rem) keep_byte = stg[nflush + tid];
				if (last && tid == 0) a.sizes[sl] = cap_hit ? 0u : (out_pos + total);
				out_pos += nflush;
				carry = rem;
			}
			// ---- size the tokens of super-tile s-1
			if (s >= 1 && s <= NS && !(dbg & 64u)) {
				const int e = s - 1;
				const int b = e * 512 + (tid >> 1);
				t_own = prev_own;
				t_r = seg ? role_get(b) : 0;
				t_nbytes = 0;
				// predecessor pixel: usually the previous lane's last pixel
				int prev = (int)dpp0<DPP_WAVE_SHR1>((uint32_t)px_of(t_own, 7));
				if (lane == 0) {
					if (tid > 0) prev = (int)dlin[m1 * STP + tid * 8 - 1];
					else if (e > 0) prev = (int)dlin[m2 * STP + STP - 1];
					else prev = 0;
				}
				if (t_r != ROLE_PARTNER) {
					if (hb == 0 && b > 0 && seg) {
						int q = b - 1;
						int rq = role_get(q);
						if (rq != 0) {  // previous block meshed: last pixel of the previous GROUP
							while (rq == ROLE_PARTNER) { q--; rq = role_get(q); }
							const int k = (q + rq) * 16 + 15;
							const int st = k >> 13;
							prev = (int)dlin[((st == e) ? m1 : (st < e) ? m2 : m) * STP + (k & (STP - 1))];
						}
					}
					uint32_t n2 = 0;
					if (t_r == 0) {
						t_prev = prev;
#pragma unroll
						for (int i = 0; i < 8; i++) {
							const int v = px_of(t_own, i);
							cnt2(v - prev, n2);
							prev = v;
						}
						t_nbytes = 8 + n2;
					} else {
						const int kp = (b + t_r) * 16 + hb * 8;
						LDS(uint16_t) *pp = dlin + (((kp >> 13) == e) ? m1 : m) * STP + (kp & (STP - 1));
						t_par = *(LDS(u32x4) *)(pp);
						if (hb == 1) prev = (int)pp[-1];  // B[7] precedes A[8]
						t_prev = prev;
#pragma unroll
						for (int i = 0; i < 8; i++) {
							const int va = px_of(t_own, i), vb = px_of(t_par, i);
							cnt2(va - prev, n2);
							cnt2(vb - va, n2);
							prev = vb;
						}
						t_nbytes = 16 + n2 + (hb == 0 ? 1u : 0u);
						if (hb == 0) my_jump += 1;
					}
					my_full += n2;
				}
				if (a.roles_out && hb == 0) a.roles_out[(size_t)sl * NB + b] = (uint8_t)t_r;
				const uint32_t inc = wave_incl_scan(t_nbytes);
				t_excl = inc - t_nbytes;
				if (lane == 63) wtot[(e & 1) * 16 + wave] = inc;
			}
			if (s + 1 < NS) stage_to_lds(s + 1, rnext);
			if (s + 5 < NS) rnext = load_st(s + 5);
		};

		for (int sb = 0; sb < NS + 2; sb += 4) {
			p2_iter(sb + 0, r1);
			p2_iter(sb + 1, r2);
			p2_iter(sb + 2, r3);
			p2_iter(sb + 3, r0);
		}
	}
	// ---- statistics / status
	if (my_q7) atomicOr((uint32_t *)&ctr[0], CCT_ST_Q7);
	if (a.stats) {
		if (my_full) atomicAdd((uint32_t *)&ctr[1], my_full);
		if (my_jump) atomicAdd((uint32_t *)&ctr[2], my_jump);
	}
	__syncthreads();
	if (tid == 0) {
		a.status[sl] = ctr[0] | (cap_hit ? CCT_ST_CAP : 0u);
		if (a.stats) {
			uint32_t *st = a.stats + (size_t)sl * 4;
			st[0] = (uint32_t)N - ctr[1];
			st[1] = ctr[1];
			st[2] = ctr[2];
			st[3] = ndiff;
		}
	}
}

}  // namespace

size_t enc_tiles_lds_bytes(int NB, bool *role_in_lds)
{
	const size_t avail = 160 * 1024;
	const bool in_lds = (size_t)L_ROLE + (size_t)NB + 16 <= avail;
	if (role_in_lds) *role_in_lds = in_lds;
	return (size_t)L_ROLE + (in_lds ? (((size_t)NB + 15) & ~(size_t)15) : 0);
}

hipError_t launch_encode_tiles(const TileEncArgs &ta, int n, hipStream_t s)
{
	bool in_lds;
	const size_t lds = enc_tiles_lds_bytes(ta.e.NB, &in_lds);
	const bool sg = (ta.e.flags & CCT_FLAG_SIGNED_SEG) != 0;
	void (*k)(TileEncArgs) = in_lds ? (sg ? encode_tiles_kernel<true, true> : encode_tiles_kernel<false, true>)
	                                : (sg ? encode_tiles_kernel<true, false> : encode_tiles_kernel<false, false>);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k, dim3(n), dim3(TT), lds, s, ta);
	return hipGetLastError();
}

}  // namespace cct
