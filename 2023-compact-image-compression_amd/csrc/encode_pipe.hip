// CompaCT encode, stage (i) as a tile-parallel pipeline of three kernels (block_size 16, traversal made of aligned
// 64x64 tiles whose 4x4-pixel blocks are traversal blocks: every power-of-two square up to 1024x1024).
//
//   K1 pipe_analyse_kernel   grid = slices x groups of tiles, 128 lanes = one tile per step
//        HBM -> VGPR   16 bytes per lane and row: a lane owns a PAIR of horizontally adjacent 4x4 blocks (8x4 pixels),
//                      a wave-instruction reads eight full 128-byte lines
//        in registers  the 16 pixels of a block are put in traversal order with v_perm_b32 (selectors per block
//                      orientation) -- no per-pixel LDS gather
//        VGPR -> LDS   traversal-ordered tile image (32 bytes per block), three tiles deep for the mesh look-ahead
//        analysis      packed 16-bit deltas, |delta| > 64 counts (cluster.py:30-59), token bytes of the block if it is
//                      emitted alone (core.py:316-323), candidate fit masks of the difficult blocks (cluster.py:122-158)
//   K2 pipe_resolve_kernel   one workgroup per slice: greedy first fit over the islands of difficult blocks
//                      (cluster.py:79-190), the token bytes of the meshed pairs, the predecessor pixel of every block that
//                      follows a meshed block, payload offset of every tile
//   K3 pipe_pack_kernel      grid = slices x tiles: same front end, tokens of every block formed four pixels at a time
//                      with byte permutes through a 16-entry table, OR-ed into a zeroed LDS image of the tile's payload
//                      bytes at their final offsets, flushed with aligned 16-byte stores
//
// Probed on the MI355X (tools/microbench/lds_probe.hip): ds_write_b8, ds_write_b32 and ds_or_b32 all cost ~4.5 cycles per
// wave-instruction, LDS stores at addresses that are not multiples of the access size ~79 cycles, and 8-byte-per-lane
// block-shaped global reads reach 4.5 TB/s where 16-byte-per-lane full lines reach 6.4 TB/s.
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS(T) __attribute__((address_space(3))) T

constexpr int PW = 128;             // lanes per workgroup of K1 / K3: one 64x64 tile, a pair of blocks per lane
constexpr int TILE_BYTES = 8192;
constexpr int STG_BYTES = 10496;    // payload bytes of one tile: 15 (alignment) + 193 * 32.5 + 63 * 65 + EOF + pad
constexpr int PAIR_CAP = 256;       // leaders per tile

// ---- packed 16-bit arithmetic (two pixels per instruction) --------------------------------------------
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ uint32_t pk_lshr(uint32_t sh_both_halves, uint32_t x)
{
	uint32_t r;
	asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh_both_halves), "v"(x));
	return r;
}
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
	return __builtin_amdgcn_perm(hi, lo, sel);  // byte i of the result = byte sel[i] of {hi:lo}; 0x0c = 0x00
}
// LDS atomics (ds_add / ds_or): the address space stays in the pointer type, so no flat_atomic is emitted
template <class T>
__device__ __forceinline__ T lds_add(LDS(T) *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or(LDS(uint32_t) *p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	v += dpp0<0x111>(v);
	v += dpp0<0x112>(v);
	v += dpp0<0x114>(v);
	v += dpp0<0x118>(v);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
	return v;
}

// ---- front end shared by K1 and K3 ---------------------------------------------------------------------
// Rows 0..3 of a 4x4 block as (columns 0-1, columns 2-3) dwords -> its 16 pixels in traversal order.
// ot: the block orientation's table (eight selectors, then the quadrant choice bits).  A 4x4 block is walked quadrant
// by quadrant (2x2 pixels each); quarter 0 is the top-left or the bottom-right quadrant, quarter 2 the other one,
// quarter 1 the bottom-left or the top-right one, quarter 3 the other (api.cpp verifies this for every orientation).
__device__ __forceinline__ void permute_block(uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1, uint32_t l2, uint32_t h2,
                                              uint32_t l3, uint32_t h3, const LDS(uint32_t) *ot, uint32_t d[8])
{
	const u32x4 s0 = *(const LDS(u32x4) *)ot, s1 = *(const LDS(u32x4) *)(ot + 4);
	const uint32_t cb = ot[8];
	const bool c0 = (cb & 1u) != 0, c1 = (cb & 2u) != 0;
	// raster quadrants as (top dword, bottom dword): TL = (l0,l1)  TR = (h0,h1)  BL = (l2,l3)  BR = (h2,h3)
	const uint32_t t0 = c0 ? h2 : l0, b0 = c0 ? h3 : l1;
	const uint32_t t2 = c0 ? l0 : h2, b2 = c0 ? l1 : h3;
	const uint32_t t1 = c1 ? h0 : l2, b1 = c1 ? h1 : l3;
	const uint32_t t3 = c1 ? l2 : h0, b3 = c1 ? l3 : h1;
	d[0] = perm(b0, t0, s0.x); d[1] = perm(b0, t0, s0.y);
	d[2] = perm(b1, t1, s0.z); d[3] = perm(b1, t1, s0.w);
	d[4] = perm(b2, t2, s1.x); d[5] = perm(b2, t2, s1.y);
	d[6] = perm(b3, t3, s1.z); d[7] = perm(b3, t3, s1.w);
}

// packed deltas x[j] = (D[2j] - D[2j-1], D[2j+1] - D[2j]) mod 2^16 of 16 traversal-ordered pixels after pixel pv
__device__ __forceinline__ void deltas16(const uint32_t d[8], uint32_t pv, uint32_t x[8])
{
	x[0] = pk_sub(d[0], (d[0] << 16) | (pv & 0xFFFFu));
#pragma unroll
	for (int j = 1; j < 8; j++) x[j] = pk_sub(d[j], __builtin_amdgcn_alignbit(d[j], d[j - 1], 16));
}

__device__ __forceinline__ int px16(const uint32_t d[8], int i) { return (int)((d[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu); }

// two-byte token: delta outside [-63, 64] (core.py:316); large: |delta| > 64 (cluster.py:37-38)
__device__ __forceinline__ bool tok_two(int dlt) { return (uint32_t)(dlt + 63) > 127u; }
__device__ __forceinline__ bool seg_large(int dlt) { return (uint32_t)(dlt + 64) > 128u; }

// ---- K1 ---------------------------------------------------------------------------------------------------
constexpr int K1_DLIN = 0;                        // 3 x 8192
constexpr int K1_RTAB = 3 * TILE_BYTES;           // TILE_MAX_ORIENT x 512
constexpr int K1_OTAB = K1_RTAB + TILE_MAX_ORIENT * 512;  // 256
constexpr int K1_LST = K1_OTAB + 256;             // 3 x 2 x 128 u16
constexpr int K1_MISC = K1_LST + 3 * 2 * 128 * 2; // lcnt[3][2], halo flag
constexpr int K1_LDS = K1_MISC + 64;

// token bytes and segmentation counts of one block; packed path: every pixel of the wave's blocks < 0x4000
template <bool SGN>
__device__ __forceinline__ void analyse_block(const uint32_t d[8], uint32_t pv, bool wide, bool first_of_slice,
                                              uint32_t &n2, uint32_t &chg, uint32_t &enter)
{
	if (!wide) {
		uint32_t x[8];
		deltas16(d, pv, x);
		const uint32_t K63 = 0x003F003Fu, ONE = 0x00010001u, SEVEN = 0x00070007u;
		uint32_t accw = 0, acce = 0, w0 = 0, e0 = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint32_t z = pk_add(x[j], K63);               // delta + 63: two-byte iff not in [0, 127]
			const uint32_t w = pk_min_u(pk_lshr(SEVEN, z), ONE);
			const uint32_t e = pk_min_u(pk_add(z, ONE), ONE);   // 0 iff delta == -64 (two bytes, but not "large")
			if (j == 0) { w0 = w & 1u; e0 = e & 1u; }
			accw = pk_add(accw, w);
			acce = pk_add(acce, e);
		}
		const uint32_t sw = (accw & 0xFFFFu) + (accw >> 16), se = (acce & 0xFFFFu) + (acce >> 16);
		n2 = sw;
		enter = w0 & e0;
		chg = (sw - w0) - ((16u - se) - (1u - e0));
	} else {
		n2 = 0; chg = 0; enter = 0;
		int pu = (int)(pv & 0xFFFFu);
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const int v = px16(d, i);
			n2 += tok_two(v - pu) ? 1u : 0u;
			const int ds = SGN ? ((int)(int16_t)v - (int)(int16_t)pu) : (v - pu);
			const uint32_t lg = seg_large(ds) ? 1u : 0u;
			if (i == 0) enter = lg; else chg += lg;
			pu = v;
		}
	}
	if (first_of_slice) enter = 0;  // P[0] = 0: the first pixel has no entering transition (cluster.py:33)
}

// fit mask of difficult block A against the 63 candidates (one per lane; lane j looks at block i + j): cluster.py:122-158
template <bool SGN>
__device__ __forceinline__ uint64_t fit_mask(const LDS(uint16_t) *ap, const LDS(uint16_t) *bp, bool valid, uint32_t cur, bool block0)
{
	const u32x4 a0 = *(const LDS(u32x4) *)(ap), a1 = *(const LDS(u32x4) *)(ap + 8);
	const u32x4 b0 = *(const LDS(u32x4) *)(bp), b1 = *(const LDS(u32x4) *)(bp + 8);
	const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
	const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
	uint32_t hi_or = 0;
#pragma unroll
	for (int q = 0; q < 8; q++) hi_or |= aw[q] | bw[q];
	const bool small = !SGN && !__any((hi_or & 0xC000C000u) != 0);
	uint32_t up;
	if (small) {
		// up = #(B[t] - A[t] >= 65) + #(A[t+1] - B[t] >= 65); all values < 16384, so packed 16-bit differences are
		// exact: count the NEGATIVE results of B[t] - (A[t] + 65) and (A[t+1] - 65) - B[t]
		const uint32_t FIFTEEN = 0x000F000Fu, K65 = 0x00410041u;
		uint32_t neg = 0;
#pragma unroll
		for (int q = 0; q < 8; q++) {
			neg = pk_add(neg, pk_lshr(FIFTEEN, pk_sub(bw[q], pk_add(aw[q], K65))));
			const uint32_t an = (q < 7) ? __builtin_amdgcn_alignbit(aw[q + 1], aw[q], 16) : (aw[7] >> 16);
			uint32_t hi = pk_sub(an, K65);
			if (q == 7) hi |= 0xFFFF0000u;  // t = 15 has no successor: -1 - B[15] is always negative
			neg = pk_add(neg, pk_lshr(FIFTEEN, pk_sub(hi, bw[q])));
		}
		up = 32u - ((neg & 0xFFFFu) + (neg >> 16));
	} else {
		up = 0;
		int bprev = 0;
#pragma unroll
		for (int t = 0; t < 16; t++) {
			int av = (int)((aw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu), bv = (int)((bw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu);
			if (SGN) { av = (int)(int16_t)av; bv = (int)(int16_t)bv; }
			if (t > 0) up += (av - bprev >= 65) ? 1u : 0u;
			up += (bv - av >= 65) ? 1u : 0u;
			bprev = bv;
		}
	}
	// cluster.py:153,158: up + 1 < current_delta - 2 in uint32; block 0 wraps: it always fits
	const bool fit = valid && (block0 ? true : ((up + 1u) < (cur - 2u)));
	return __ballot(fit);
}

template <bool SGN>
__global__ void __launch_bounds__(PW) pipe_analyse_kernel(PipeArgs a, int tpw)
{
	__shared__ __attribute__((aligned(16))) uint8_t smem[K1_LDS];
	LDS(uint8_t) *dlin = (LDS(uint8_t) *)(smem + K1_DLIN);
	LDS(uint32_t) *rtab = (LDS(uint32_t) *)(smem + K1_RTAB);
	LDS(uint32_t) *otab = (LDS(uint32_t) *)(smem + K1_OTAB);
	LDS(uint16_t) *lst = (LDS(uint16_t) *)(smem + K1_LST);
	LDS(uint32_t) *lcnt = (LDS(uint32_t) *)(smem + K1_MISC);
	LDS(uint32_t) *halo_flag = lcnt + 8;

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int NT = a.n_tiles, NB = a.e.NB, N = a.e.N;
	const int wps = (NT + tpw - 1) / tpw;
	const int sl = blockIdx.x / wps;
	const int t0 = (blockIdx.x % wps) * tpw;
	const int nT = min(tpw, NT - t0);
	const bool seg = (a.e.flags & CCT_FLAG_SEGMENTATION) != 0;
	const uint16_t *img = a.e.images + (size_t)sl * N;
	uint8_t *ssz = a.ssz + (size_t)sl * NB;
	uint64_t *gmask = a.mask + (size_t)sl * NB;
	const int pitch = a.row_pitch;

	// lane's region of a tile: block row by (0..15), block pair bxp (0..7)
	const int by = tid >> 3, bxp = tid & 7;
	const size_t reg_off = (size_t)(by * 4) * pitch + bxp * 8;
	auto load_tile = [&](int tile, u32x4 r[4]) {
		const uint16_t *p = img + a.tile_org[tile] + reg_off;
#pragma unroll
		for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);
	};
	u32x4 r[4];
	load_tile(t0, r);
	// pixel before the first tile of this workgroup (the slice starts at 0, core.py:278)
	uint32_t first_prev = 0;
	if (t0 > 0) first_prev = img[a.tile_org[t0 - 1] + a.tile_last[a.tile_orient[t0 - 1]]];
	{
		const int nr = a.n_orient * 128;
		for (int i = tid; i < nr; i += PW) rtab[i] = reinterpret_cast<const uint32_t *>(a.rtab)[i];
		if (tid < 64) otab[tid] = a.otab[tid];
		if (tid < 9) lcnt[tid] = 0;
	}
	__syncthreads();

	uint32_t dA[8], dB[8];   // the lane's two blocks of the current tile, traversal order
	int kA = 0, kB = 0;
	bool halo = false;
	for (int s = 0; s <= nT; s++) {
		const int tile = t0 + s;
		const bool body = s < nT;
		const bool have = body || halo;
		LDS(uint8_t) *slot = dlin + (s % 3) * TILE_BYTES;
		if (have) {
			const uint32_t e2 = rtab[(int)a.tile_orient[tile] * 128 + tid];  // entries of the left and the right block
			kA = (int)(e2 & 0xFFu); kB = (int)((e2 >> 16) & 0xFFu);
			permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, otab + ((e2 >> 8) & 3u) * 16, dA);
			permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, otab + ((e2 >> 24) & 3u) * 16, dB);
			*(LDS(u32x4) *)(slot + kA * 32) = (u32x4){dA[0], dA[1], dA[2], dA[3]};
			*(LDS(u32x4) *)(slot + kA * 32 + 16) = (u32x4){dA[4], dA[5], dA[6], dA[7]};
			*(LDS(u32x4) *)(slot + kB * 32) = (u32x4){dB[0], dB[1], dB[2], dB[3]};
			*(LDS(u32x4) *)(slot + kB * 32 + 16) = (u32x4){dB[4], dB[5], dB[6], dB[7]};
		}
		if (s + 1 < nT) load_tile(tile + 1, r);  // in flight across the analysis below
		lds_barrier();                           // lgkmcnt(0) only: the prefetch stays in flight
		if (body) {
			// ---- analysis of the lane's two blocks (cluster.py:30-59, core.py:316-323)
			uint32_t orall = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) orall |= dA[j] | dB[j];
			const bool wide = SGN || __any((orall & 0xC000C000u) != 0);
			LDS(uint8_t) *pslot = dlin + ((s + 2) % 3) * TILE_BYTES;
			uint32_t ndiff_lane[2] = {0, 0}, cur_lane[2] = {0, 0};
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int k = h ? kB : kA;
				const uint32_t *d = h ? dB : dA;
				uint32_t pv;
				if (k > 0) pv = *(const LDS(uint16_t) *)(slot + k * 32 - 2);
				else if (s > 0) pv = *(const LDS(uint16_t) *)(pslot + TILE_BYTES - 2);
				else pv = first_prev;
				const bool first_of_slice = (tile == 0 && k == 0);
				uint32_t n2, chg, enter;
				analyse_block<SGN>(d, pv, wide, first_of_slice, n2, chg, enter);
				const bool difficult = seg && chg >= 8u;                      // cluster.py:58
				ssz[tile * 256 + k] = (uint8_t)((16u + n2) | (difficult ? 0x80u : 0u));
				ndiff_lane[h] = difficult ? 1u : 0u;
				cur_lane[h] = chg + enter;                                    // cluster.py:110
			}
			// difficult blocks of this tile: per-wave lists, order irrelevant for the masks
			const uint64_t balA = __ballot(ndiff_lane[0] != 0), balB = __ballot(ndiff_lane[1] != 0);
			const uint32_t nA = (uint32_t)__popcll(balA);
			LDS(uint16_t) *mylst = lst + ((s % 3) * 2 + wave) * 128;
			const uint64_t below = (1ull << lane) - 1ull;
			if (ndiff_lane[0]) mylst[__popcll(balA & below)] = (uint16_t)(kA | (cur_lane[0] << 8));
			if (ndiff_lane[1]) mylst[nA + __popcll(balB & below)] = (uint16_t)(kB | (cur_lane[1] << 8));
			if (lane == 0) lcnt[(s % 3) * 2 + wave] = nA + (uint32_t)__popcll(balB);
			if (s == nT - 1 && t0 + nT < NT) {
				// look-ahead of the last tile reaches into the next workgroup's first tile: fetch it only if needed
				const bool needA = ndiff_lane[0] && kA >= 193, needB = ndiff_lane[1] && kB >= 193;
				if (__any(needA || needB) && lane == 0) *halo_flag = 1;
			}
		}
		if (s == nT - 1) {
			__syncthreads();
			halo = (*halo_flag != 0);
			if (halo) load_tile(t0 + nT, r);
		}
		// ---- candidate masks of the previous tile (its list was completed before this step's barrier)
		if (s >= 1) {
			const int ps = s - 1;
			LDS(uint8_t) *cs = dlin + (ps % 3) * TILE_BYTES;
			const uint32_t c0 = lcnt[(ps % 3) * 2], c1 = lcnt[(ps % 3) * 2 + 1];
			const LDS(uint16_t) *l0 = lst + (ps % 3) * 2 * 128;
			const int tb = (t0 + ps) * 256;
			for (uint32_t e = wave; e < c0 + c1; e += 2) {
				const uint32_t ent = e < c0 ? l0[e] : l0[128 + e - c0];
				const int i = (int)(ent & 0xFFu);
				const uint32_t cur = ent >> 8;
				const int p = i + lane;
				const bool valid = lane >= 1 && tb + p < NB;
				const int pk = valid ? p : i;
				const LDS(uint16_t) *ap = (const LDS(uint16_t) *)(cs + i * 32);
				const LDS(uint16_t) *bp = (const LDS(uint16_t) *)((pk < 256 ? cs : slot) + (pk & 255) * 32);
				const uint64_t mk = fit_mask<SGN>(ap, bp, valid, cur, tb + i == 0);
				if (lane == 0) gmask[tb + i] = mk;
			}
		}
	}
}

// ---- K2 ---------------------------------------------------------------------------------------------------
constexpr int K2T = 256;
constexpr int K2_CAP = 4096;                      // difficult-list records kept in LDS
constexpr int K2_ROLE = 0;                        // PIPE_MAX_NB bytes
constexpr int K2_IDX = PIPE_MAX_NB;               // K2_CAP u32
constexpr int K2_MASK = K2_IDX + K2_CAP * 4;      // K2_CAP u64
constexpr int K2_TSUM = K2_MASK + K2_CAP * 8;     // TILE_MAX_TILES/4 = 256 i32 (sum of single sizes + corrections)
constexpr int K2_MISC = K2_TSUM + 256 * 4;
constexpr int K2_LDS = K2_MISC + 128;

__device__ __forceinline__ uint32_t wg_incl_scan256(uint32_t v, LDS(uint32_t) *scratch, int tid, uint32_t &total)
{
	const uint32_t inc = wave_incl_scan(v);
	if ((tid & 63) == 63) scratch[tid >> 6] = inc;
	__syncthreads();
	const uint32_t w0 = scratch[0], w1 = scratch[1], w2 = scratch[2], w3 = scratch[3];
	const int w = tid >> 6;
	const uint32_t base = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
	total = w0 + w1 + w2 + w3;
	__syncthreads();
	return base + inc;
}

__global__ void __launch_bounds__(K2T) pipe_resolve_kernel(PipeArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem2[];
	LDS(uint8_t) *role = (LDS(uint8_t) *)(smem2 + K2_ROLE);
	LDS(uint32_t) *l_idx = (LDS(uint32_t) *)(smem2 + K2_IDX);
	LDS(uint64_t) *l_mask = (LDS(uint64_t) *)(smem2 + K2_MASK);
	LDS(int32_t) *tsum = (LDS(int32_t) *)(smem2 + K2_TSUM);
	LDS(uint32_t) *misc = (LDS(uint32_t) *)(smem2 + K2_MISC);  // [0..3] scan scratch, [4] pair counter, [5] status

	const int tid = threadIdx.x;
	const int sl = blockIdx.x;
	const int NB = a.e.NB, N = a.e.N, NT = a.n_tiles;
	const bool seg = (a.e.flags & CCT_FLAG_SEGMENTATION) != 0;
	const uint16_t *img = a.e.images + (size_t)sl * N;
	const int32_t *O = a.e.lut;
	const uint8_t *ssz = a.ssz + (size_t)sl * NB;
	const uint64_t *gmask = a.mask + (size_t)sl * NB;
	uint32_t *spec = a.spec + (size_t)sl * NB;
	uint32_t *g_idx = a.spill_idx + (size_t)sl * NB;
	uint8_t *rec = a.pairrec + (size_t)sl * (NB / 2) * PIPE_PAIR_REC;
	auto PX = [&](int pos) -> int { return (int)img[O[pos]]; };
	auto idx_of = [&](uint32_t e) -> uint32_t { uint32_t v; if (e < K2_CAP) v = l_idx[e]; else v = g_idx[e]; return v; };
	auto mask_of = [&](uint32_t e, uint32_t i) -> uint64_t { uint64_t v; if (e < K2_CAP) v = l_mask[e]; else v = gmask[i]; return v; };

	for (int i = tid; i < NB / 16; i += K2T) *(LDS(u32x4) *)(role + i * 16) = (u32x4){0, 0, 0, 0};
	for (int i = tid; i < NT; i += K2T) tsum[i] = 0;
	if (tid < 8) misc[tid] = 0;
	__syncthreads();

	// ---- ordered list of difficult blocks + sum of the single sizes per tile
	const int n16 = NB / 16;                              // 16 blocks per chunk, 16 chunks per tile
	const int cpl = (n16 + K2T - 1) / K2T;                // chunks per lane (contiguous ranges keep the order)
	uint32_t cnt = 0;
	for (int c = tid * cpl; c < min(n16, (tid + 1) * cpl); c++) {
		const u32x4 v = *reinterpret_cast<const u32x4 *>(ssz + (size_t)c * 16);
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		uint32_t sum = 0;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			cnt += (uint32_t)__popc(w[q] & 0x80808080u);
			const uint32_t sz = w[q] & 0x7F7F7F7Fu;
			sum += (sz & 0xFFu) + ((sz >> 8) & 0xFFu) + ((sz >> 16) & 0xFFu) + (sz >> 24);
		}
		lds_add(&tsum[c >> 4], (int32_t)sum);
	}
	uint32_t ndiff;
	uint32_t pos = wg_incl_scan256(cnt, misc, tid, ndiff) - cnt;
	if (seg && ndiff) {
		for (int c = tid * cpl; c < min(n16, (tid + 1) * cpl); c++) {
			const u32x4 v = *reinterpret_cast<const u32x4 *>(ssz + (size_t)c * 16);
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				uint32_t m = w[q] & 0x80808080u;
				while (m) {
					const int bit = __ffs((int)m) - 1;
					m &= m - 1;
					const uint32_t b = (uint32_t)(c * 16 + q * 4 + (bit >> 3));
					if (pos < K2_CAP) { l_idx[pos] = b; l_mask[pos] = gmask[b]; } else g_idx[pos] = b;
					pos++;
				}
			}
		}
	}
	__syncthreads();

	// ---- resolve: greedy first fit (cluster.py:79-190), one lane per island of difficult blocks
	if (seg) {
		for (uint32_t e0 = tid; e0 < ndiff; e0 += K2T) {
			const uint32_t i0 = idx_of(e0);
			if (e0 > 0 && i0 - idx_of(e0 - 1) <= 63u) continue;  // not the head of an island
			uint64_t cw = 0;                                    // blocks at base + bit already consumed as partners
			uint32_t base = i0, e = e0, i = i0;
			for (;;) {
				const uint32_t sh = i - base;
				cw = (sh >= 64u) ? 0ull : (cw >> sh);
				base = i;
				if (!(cw & 1ull)) {
					const uint64_t avail = mask_of(e, i) & ~cw & ~1ull;
					if (avail) {
						const int j = __ffsll((long long)avail) - 1;
						role[i] = (uint8_t)j;
						role[i + j] = ROLE_PARTNER;
						cw |= 1ull << j;
					}
				}
				if (++e >= ndiff) break;
				const uint32_t inext = idx_of(e);
				if (inext - i > 63u) break;
				i = inext;
			}
		}
	}
	__syncthreads();

	// ---- meshed pairs: their token bytes (core.py:281-323 along the interleaved order of cluster.py:173-174), and the
	// blocks that follow a meshed block: their predecessor pixel is the last pixel of the previous GROUP
	// last pixel written before block b's group (b > 0, b not a partner)
	auto true_prev = [&](int b) -> int {
		int q = b - 1;
		int rq = role[q];
		if (rq != 0) {
			while (rq == ROLE_PARTNER) { q--; rq = role[q]; }
			if (rq != 0) q += rq;  // a pair ends with its partner's last pixel
		}
		return PX(q * 16 + 15);
	};
	uint32_t q7 = 0;
	if (seg) {
		for (uint32_t e = tid; e < ndiff; e += K2T) {
			const int i = (int)idx_of(e);
			const int r = role[i];
			if (r == 0 || r == ROLE_PARTNER) continue;
			const int p = i + r;
			const uint32_t slot = lds_add(&misc[4], 1u);
			uint8_t *out = rec + (size_t)slot * PIPE_PAIR_REC;
			int prev = i > 0 ? true_prev(i) : 0;
			int va[16], vb[16];
#pragma unroll
			for (int t = 0; t < 16; t++) { va[t] = PX(i * 16 + t); vb[t] = PX(p * 16 + t); }
			int n = 0;
			out[n++] = (uint8_t)(0x80 | r);  // core.py:290-294
			auto put = [&](int dlt) {
				if (tok_two(dlt)) {
					out[n++] = (uint8_t)(0xE0 | ((dlt >> 8) & 0x0F));
					out[n++] = (uint8_t)(dlt & 0xFF);
					if ((uint32_t)(dlt + 2047) > 4095u) q7 = 1;  // outside [-2047, 2048]
				} else out[n++] = (uint8_t)(dlt & 0x7F);
			};
#pragma unroll
			for (int t = 0; t < 16; t++) { put(va[t] - prev); put(vb[t] - va[t]); prev = vb[t]; }
			spec[i] = (slot << 8) | (uint32_t)n;
			lds_add(&tsum[i >> 8], n - (int)(ssz[i] & 0x7F));
			lds_add(&tsum[p >> 8], -(int)(ssz[p] & 0x7F));
			// blocks after the leader and after the partner, if emitted alone
			for (int h = 0; h < 2; h++) {
				const int b = (h ? p : i) + 1;
				if (b >= NB || role[b] != 0) continue;
				const int tp = true_prev(b), dp = PX(b * 16 - 1), v0 = PX(b * 16);
				spec[b] = (uint32_t)tp;
				const int corr = (int)tok_two(v0 - tp) - (int)tok_two(v0 - dp);
				if (corr) lds_add(&tsum[b >> 8], corr);
			}
		}
	}
	if (q7) lds_or(&misc[5], CCT_ST_Q7);
	__syncthreads();

	// ---- roles to HBM (K3 and the caller), tile offsets, slice size, statistics
	uint8_t *groles = a.roles + (size_t)sl * NB;
	uint8_t *oroles = a.e.roles_out ? a.e.roles_out + (size_t)sl * NB : nullptr;
	for (int i = tid; i < NB / 16; i += K2T) {
		const u32x4 v = *(const LDS(u32x4) *)(role + i * 16);
		*reinterpret_cast<u32x4 *>(groles + (size_t)i * 16) = v;
		if (oroles) *reinterpret_cast<u32x4 *>(oroles + (size_t)i * 16) = v;
	}
	uint32_t *toff = a.toff + (size_t)sl * (NT + 1);
	uint32_t total;
	{
		const uint32_t mine = tid < NT ? (uint32_t)tsum[tid] : 0u;
		const uint32_t inc = wg_incl_scan256(mine, misc, tid, total);
		if (tid < NT) toff[tid] = inc - mine;
		if (tid == 0) toff[NT] = total;
	}
	if (tid == 0) {
		const uint32_t njump = misc[4];
		const uint32_t size = total + (a.e.eof >= 0 ? 1u : 0u);
		const bool cap = (size_t)((size + 15u) & ~15u) > a.e.stride;
		a.e.sizes[sl] = cap ? 0u : size;
		a.e.status[sl] = misc[5] | (cap ? CCT_ST_CAP : 0u);
		if (a.e.stats) {
			uint32_t *st = a.e.stats + (size_t)sl * 4;
			const uint32_t nfull = total - (uint32_t)N - njump;
			st[0] = (uint32_t)N - nfull; st[1] = nfull; st[2] = njump; st[3] = seg ? ndiff : 0u;
		}
	}
}

// ---- K3 ---------------------------------------------------------------------------------------------------
constexpr int K3_DLIN = 0;                        // 8192
constexpr int K3_STG = TILE_BYTES;                // STG_BYTES
constexpr int K3_TTAB = K3_STG + STG_BYTES;       // 16 entries x 128 bytes (the index arrives as mask-sum << 7)
constexpr int K3_RTAB = K3_TTAB + 2048;           // 512
constexpr int K3_OTAB = K3_RTAB + 512;            // 256
constexpr int K3_PAIR = K3_OTAB + 256;            // PAIR_CAP x 8
constexpr int K3_MISC = K3_PAIR + PAIR_CAP * 8;   // [0..1] wave totals, [2] pair count, [3] status
constexpr int K3_LDS = K3_MISC + 64;

// two-byte masks of the four 4-pixel groups of a block: bit 7 of byte p of m[g] <=> pixel 4g+p takes two bytes
__device__ __forceinline__ uint32_t group_masks(const uint32_t x[8], uint32_t m[4])
{
	const uint32_t K64 = 0x00400040u, K63 = 0x003F003Fu;
	uint32_t n2 = 0;
#pragma unroll
	for (int g = 0; g < 4; g++) {
		const uint32_t wa = pk_sub(K64, x[2 * g]) | pk_add(x[2 * g], K63);         // sign: delta > 64 | delta < -63
		const uint32_t wb = pk_sub(K64, x[2 * g + 1]) | pk_add(x[2 * g + 1], K63);
		m[g] = perm(wb, wa, 0x07050301u) & 0x80808080u;
		n2 += (uint32_t)__popc(m[g]);
	}
	return n2;
}
// the same from exact 17-bit differences (pixels >= 0x4000 present)
__device__ __forceinline__ uint32_t group_masks_wide(const uint32_t d[8], uint32_t pv, uint32_t m[4], bool &q7)
{
	uint32_t n2 = 0;
	int pu = (int)(pv & 0xFFFFu);
#pragma unroll
	for (int g = 0; g < 4; g++) {
		uint32_t mg = 0;
#pragma unroll
		for (int p = 0; p < 4; p++) {
			const int v = px16(d, 4 * g + p), dlt = v - pu;
			if (tok_two(dlt)) { mg |= 0x80u << (8 * p); n2++; q7 |= (uint32_t)(dlt + 2047) > 4095u; }
			pu = v;
		}
		m[g] = mg;
	}
	return n2;
}

// tokens of one block OR-ed into the (zeroed) payload image at byte offset o
__device__ __forceinline__ void emit_block(const uint32_t x[8], const uint32_t m[4], uint32_t o, LDS(uint8_t) *stg,
                                           const LDS(uint8_t) *ttab)
{
#pragma unroll
	for (int g = 0; g < 4; g++) {
		const uint32_t xa = x[2 * g], xb = x[2 * g + 1];
		const uint32_t pd = perm(xb, xa, 0x06040200u);                               // low bytes of the four deltas
		const uint32_t P = (pd & 0x7F7F7F7Fu) | (pd & m[g]);                         // short: 7 bits; full: second byte
		const uint32_t X = (perm(xb, xa, 0x07050301u) & 0x0F0F0F0Fu) | 0xE0E0E0E0u;  // full: first byte
		const uint32_t ti = __builtin_amdgcn_udot4(m[g], 0x08040201u, 0u, false);    // mask as 4 bits << 7
		const u32x4 te = *(const LDS(u32x4) *)(ttab + ti);
		const uint32_t lo = perm(X, P, te.x), hi = perm(X, P, te.y);
		// shift the 4..8 bytes to the byte phase of o and merge
		const uint32_t s8 = (o & 3u) * 8u;
		const uint64_t v01 = ((uint64_t)hi << 32 | lo) << s8;
		const uint32_t d2 = (uint32_t)(((uint64_t)hi << s8) >> 32);
		LDS(uint32_t) *w = (LDS(uint32_t) *)(stg + (o & ~3u));
		lds_or(w, (uint32_t)v01);
		lds_or(w + 1, (uint32_t)(v01 >> 32));
		lds_or(w + 2, d2);
		o += te.z;
	}
}

__global__ void __launch_bounds__(PW) pipe_pack_kernel(PipeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint8_t smem[K3_LDS];
	LDS(uint8_t) *dlin = (LDS(uint8_t) *)(smem + K3_DLIN);
	LDS(uint8_t) *stg = (LDS(uint8_t) *)(smem + K3_STG);
	LDS(uint8_t) *ttab = (LDS(uint8_t) *)(smem + K3_TTAB);
	LDS(uint32_t) *rtab = (LDS(uint32_t) *)(smem + K3_RTAB);
	LDS(uint32_t) *otab = (LDS(uint32_t) *)(smem + K3_OTAB);
	LDS(uint32_t) *pairs = (LDS(uint32_t) *)(smem + K3_PAIR);
	LDS(uint32_t) *misc = (LDS(uint32_t) *)(smem + K3_MISC);

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int NT = a.n_tiles, NB = a.e.NB, N = a.e.N;
	const int sl = blockIdx.x / NT, tile = blockIdx.x % NT;
	const uint16_t *img = a.e.images + (size_t)sl * N;
	const int pitch = a.row_pitch;
	const int to = a.tile_orient[tile];

	// ---- loads first: pixel rows, roles, offsets
	const int by = tid >> 3, bxp = tid & 7;
	const uint16_t *p = img + a.tile_org[tile] + (size_t)(by * 4) * pitch + bxp * 8;
	u32x4 r[4];
#pragma unroll
	for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);
	const int b0 = tile * 256 + 2 * tid;  // the lane's two consecutive traversal blocks
	const uint8_t *roles = a.roles + (size_t)sl * NB;
	const uint32_t rr = *reinterpret_cast<const uint16_t *>(roles + b0);
	const uint32_t role0 = rr & 0xFFu, role1 = rr >> 8;
	const uint32_t rolep = b0 > 0 ? roles[b0 - 1] : 0u;
	const uint32_t *toff = a.toff + (size_t)sl * (NT + 1);
	const uint32_t off_t = toff[tile], off_n = toff[tile + 1];
	const uint32_t *spec = a.spec + (size_t)sl * NB;
	uint32_t tile_prev = 0;
	if (tid == 0 && tile > 0) tile_prev = img[a.tile_org[tile - 1] + a.tile_last[a.tile_orient[tile - 1]]];
	// special blocks: leaders carry their record and size, blocks after a meshed block their predecessor pixel
	const bool lead0 = role0 >= 1 && role0 <= 63, lead1 = role1 >= 1 && role1 <= 63;
	const bool succ0 = role0 == 0 && rolep != 0, succ1 = role1 == 0 && role0 != 0;
	uint32_t sp0 = 0, sp1 = 0;
	if (lead0 || succ0) sp0 = spec[b0];
	if (lead1 || succ1) sp1 = spec[b0 + 1];

	// ---- tables and the zeroed payload image
	rtab[tid] = reinterpret_cast<const uint32_t *>(a.rtab)[to * 128 + tid];
	if (tid < 64) otab[tid] = a.otab[tid];
	if (tid < 16) *(LDS(u32x4) *)(ttab + tid * 128) = reinterpret_cast<const u32x4 *>(a.ttab)[tid];
	for (int i = tid; i < STG_BYTES / 16; i += PW) *(LDS(u32x4) *)(stg + i * 16) = (u32x4){0, 0, 0, 0};
	if (tid < 4) misc[tid] = 0;
	__syncthreads();

	// ---- traversal order through LDS
	{
		const uint32_t e2 = rtab[tid];
		const int kA = (int)(e2 & 0xFFu), kB = (int)((e2 >> 16) & 0xFFu);
		uint32_t dA[8], dB[8];
		permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, otab + ((e2 >> 8) & 3u) * 16, dA);
		permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, otab + ((e2 >> 24) & 3u) * 16, dB);
		*(LDS(u32x4) *)(dlin + kA * 32) = (u32x4){dA[0], dA[1], dA[2], dA[3]};
		*(LDS(u32x4) *)(dlin + kA * 32 + 16) = (u32x4){dA[4], dA[5], dA[6], dA[7]};
		*(LDS(u32x4) *)(dlin + kB * 32) = (u32x4){dB[0], dB[1], dB[2], dB[3]};
		*(LDS(u32x4) *)(dlin + kB * 32 + 16) = (u32x4){dB[4], dB[5], dB[6], dB[7]};
	}
	__syncthreads();
	uint32_t d[16];
	{
		const LDS(u32x4) *src = (const LDS(u32x4) *)(dlin + tid * 64);
		const u32x4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
		d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
		d[8] = v2.x; d[9] = v2.y; d[10] = v2.z; d[11] = v2.w; d[12] = v3.x; d[13] = v3.y; d[14] = v3.z; d[15] = v3.w;
	}
	uint32_t pv0 = tid > 0 ? (uint32_t) * (const LDS(uint16_t) *)(dlin + tid * 64 - 2) : tile_prev;
	if (succ0) pv0 = sp0;
	uint32_t pv1 = d[7] >> 16;
	if (succ1) pv1 = sp1;

	// ---- token sizes
	uint32_t x[16], m[8];
	deltas16(d, pv0, x);
	deltas16(d + 8, pv1, x + 8);
	uint32_t orall = 0;
#pragma unroll
	for (int j = 0; j < 16; j++) orall |= d[j];
	orall |= pv0 | pv1;
	bool q7 = false;
	uint32_t n20, n21;
	if (!__any((orall & 0xC000C000u) != 0)) {
		n20 = group_masks(x, m);
		n21 = group_masks(x + 8, m + 4);
		if (__any((orall & 0xF800F800u) != 0)) {  // a delta outside [-2047, 2048] needs a pixel >= 2048
			const uint32_t K2048 = 0x08000800u, K2047 = 0x07FF07FFu;
			uint32_t bad0 = 0, bad1 = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				bad0 |= pk_sub(K2048, x[j]) | pk_add(x[j], K2047);
				bad1 |= pk_sub(K2048, x[8 + j]) | pk_add(x[8 + j], K2047);
			}
			q7 = (role0 == 0 && (bad0 & 0x80008000u)) || (role1 == 0 && (bad1 & 0x80008000u));
		}
	} else {
		bool qa = false, qb = false;
		n20 = group_masks_wide(d, pv0, m, qa);
		n21 = group_masks_wide(d + 8, pv1, m + 4, qb);
		q7 = (role0 == 0 && qa) || (role1 == 0 && qb);
	}
	const uint32_t sz0 = role0 == 0 ? 16u + n20 : (lead0 ? (sp0 & 0xFFu) : 0u);
	const uint32_t sz1 = role1 == 0 ? 16u + n21 : (lead1 ? (sp1 & 0xFFu) : 0u);
	const uint32_t inc = wave_incl_scan(sz0 + sz1);
	if (lane == 63) misc[wave] = inc;
	__syncthreads();
	const uint32_t head = off_t & 15u;
	const uint32_t tot = misc[0] + misc[1];
	uint32_t o0 = head + (wave ? misc[0] : 0u) + inc - (sz0 + sz1);
	const uint32_t o1 = o0 + sz0;

	// ---- tokens into the payload image
	if (role0 == 0) emit_block(x, m, o0, stg, ttab);
	if (role1 == 0) emit_block(x + 8, m + 4, o1, stg, ttab);
	if (lead0) { const uint32_t e = lds_add(&misc[2], 1u); pairs[2 * e] = o0; pairs[2 * e + 1] = sp0; }
	if (lead1) { const uint32_t e = lds_add(&misc[2], 1u); pairs[2 * e] = o1; pairs[2 * e + 1] = sp1; }
	uint32_t st = q7 ? CCT_ST_Q7 : 0u;
	if (tid == 0 && tot != off_n - off_t) st |= CCT_ST_INTERNAL;
	if (st) lds_or(&misc[3], st);
	__syncthreads();
	// ---- meshed pairs: K2 left their bytes in HBM records
	{
		const uint32_t npair = misc[2];
		const uint8_t *rec = a.pairrec + (size_t)sl * (NB / 2) * PIPE_PAIR_REC;
		for (uint32_t e = wave; e < npair; e += 2) {
			const uint32_t o = pairs[2 * e], sp = pairs[2 * e + 1];
			const uint32_t n = sp & 0xFFu;
			const uint8_t *src = rec + (size_t)(sp >> 8) * PIPE_PAIR_REC;
			for (uint32_t j = lane; j < n; j += 64) stg[o + j] = src[j];
		}
		if (tile == NT - 1 && a.e.eof >= 0 && tid == 0) stg[head + tot] = (uint8_t)a.e.eof;  // core.py:329-330
	}
	__syncthreads();
	// ---- flush: whole 16-byte chunks with one store, the two ends shared with the neighbouring tiles byte by byte
	{
		const bool last = tile == NT - 1;
		const uint32_t end = head + tot + ((last && a.e.eof >= 0) ? 1u : 0u);
		const size_t base = (size_t)(off_t & ~15u);
		const bool room = base + ((end + 15u) & ~15u) <= a.e.stride;
		uint8_t *out = a.e.payload + (size_t)sl * a.e.stride + base;
		const uint32_t c_first = head ? 1u : 0u;                        // chunk 0 is partial when head > 0
		const uint32_t c_end = last ? (end + 15u) / 16u : end / 16u;    // the last tile owns its padding
		if (room) {
			for (uint32_t c = c_first + tid; c < c_end; c += PW)
				*reinterpret_cast<u32x4 *>(out + (size_t)c * 16) = *(const LDS(u32x4) *)(stg + c * 16);
			if (head && tid < 16 && (uint32_t)tid >= head && (uint32_t)tid < end) out[tid] = stg[tid];
			if (!last && tid >= 16 && tid < 32) {
				const uint32_t i = c_end * 16u + (uint32_t)(tid - 16);
				if (i < end && (i >= 16u || !head)) out[i] = stg[i];
			}
		}
		if (tid == 0 && misc[3]) atomicOr(a.e.status + sl, misc[3] | (room ? 0u : CCT_ST_CAP));
	}
}

}  // namespace

hipError_t launch_encode_pipe(const PipeArgs &pa, int n, hipStream_t s)
{
	const int NT = pa.n_tiles;
	const int tpw = NT >= 16 ? 4 : (NT >= 4 ? 2 : 1);  // tiles per K1 workgroup
	const int wps = (NT + tpw - 1) / tpw;
	const bool sg = (pa.e.flags & CCT_FLAG_SIGNED_SEG) != 0;
	if (sg) hipLaunchKernelGGL(pipe_analyse_kernel<true>, dim3(n * wps), dim3(PW), 0, s, pa, tpw);
	else hipLaunchKernelGGL(pipe_analyse_kernel<false>, dim3(n * wps), dim3(PW), 0, s, pa, tpw);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	static bool attr_set = false;
	if (!attr_set) {
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(pipe_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, K2_LDS);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	hipLaunchKernelGGL(pipe_resolve_kernel, dim3(n), dim3(K2T), K2_LDS, s, pa);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(pipe_pack_kernel, dim3(n * NT), dim3(PW), 0, s, pa);
	return hipGetLastError();
}

}  // namespace cct
