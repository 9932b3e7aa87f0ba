// CompaCT encode, stage (i) as a tile-parallel pipeline of four kernels (block_size 16, traversal made of aligned
// 64x64 tiles whose 4x4-pixel blocks are traversal blocks: every power-of-two square up to 1024x1024).
//
//   K1a pipe_analyse_kernel  grid = slices x groups of tiles, 128 lanes = one tile per step, no LDS traffic, no barrier
//        HBM -> VGPR   16 bytes per lane and row: a lane owns a PAIR of horizontally adjacent 4x4 blocks (8x4 pixels),
//                      a wave-instruction reads eight full 128-byte lines; the next tile's rows are in flight meanwhile
//        in registers  the 16 pixels of a block are put in traversal order with v_perm_b32 (selectors per block
//                      orientation) -- no per-pixel gather
//        analysis      packed 16-bit deltas, |delta| > 64 counts (cluster.py:30-59), token bytes of the block if it is
//                      emitted alone (core.py:316-323) -> one byte per block (size | difficult << 7)
//   K1b pipe_masks_kernel    one wave per tile: candidate fit masks of its difficult blocks (cluster.py:122-158), one
//                      candidate per lane, the difficult block's thresholds in SGPRs
//   K2  pipe_resolve_kernel  one workgroup per slice: greedy first fit over the islands of difficult blocks
//                      (cluster.py:79-190), the token bytes of the meshed pairs, the predecessor pixel of every block that
//                      follows a meshed block, payload offset of every tile
//   K3  pipe_pack_kernel     grid = slices x tiles: same front end, tokens of every block formed four pixels at a time
//                      with byte permutes through a 16-entry table, OR-ed into a zeroed LDS image of the tile's payload
//                      bytes at their final offsets, flushed with aligned 16-byte stores
//
// Probed on the MI355X (tools/microbench/lds_probe.hip): ds_write_b8, ds_write_b32 and ds_or_b32 all cost ~4.5 cycles per
// wave-instruction, LDS stores at addresses that are not multiples of the access size ~79 cycles, and 8-byte-per-lane
// block-shaped global reads reach 4.5 TB/s where 16-byte-per-lane full lines reach 6.4 TB/s.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS(T) __attribute__((address_space(3))) T

constexpr int PW = 128;             // lanes per workgroup of K1 / K3: one 64x64 tile, a pair of blocks per lane
#define TILE_ORG(a, t) ((a).tiles.orgo[t] & 0xFFFFFFu)
#define TILE_ORIENT(a, t) ((int)((a).tiles.orgo[t] >> 24))

// ---- packed 16-bit arithmetic (two pixels per instruction) --------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)  // v_pk_sub_i16; operands may be scalar registers
{
	return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)  // v_pk_add_u16
{
	return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
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

// ---- diagnostic build only (CCT_PIPE_STAMPS=1): cycles per phase, summed per wave, written to a buffer of their own
struct Stamps {
	uint64_t acc[8];
	uint64_t last, rt0;
};
template <bool ON>
__device__ __forceinline__ void stamp_init(Stamps &st)
{
	if (ON) { for (int i = 0; i < 8; i++) st.acc[i] = 0; st.rt0 = __builtin_amdgcn_s_memrealtime(); st.last = __builtin_amdgcn_s_memtime(); }
}
template <bool ON>
__device__ __forceinline__ void stamp(Stamps &st, int phase)
{
	if (ON) {
		__builtin_amdgcn_sched_barrier(0);
		const uint64_t t = __builtin_amdgcn_s_memtime();
		st.acc[phase] += t - st.last;
		st.last = t;
		__builtin_amdgcn_sched_barrier(0);
	}
}
template <bool ON>
__device__ __forceinline__ void stamp_store(const Stamps &st, uint64_t *buf, int slot)
{
	if (ON && (threadIdx.x & 63) == 0 && buf) {
		uint64_t all = 0;
		for (int i = 0; i < 7; i++) { buf[(size_t)slot * 8 + i] = st.acc[i]; all += st.acc[i]; }
		// slot 7: shader cycles per 100 MHz tick, x 1000 (the clock the wave ran at, in units of 100 kHz)
		const uint64_t rt = __builtin_amdgcn_s_memrealtime() - st.rt0;
		buf[(size_t)slot * 8 + 7] = rt ? (all + st.acc[7]) * 1000 / rt : 0;
	}
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

// ---- K1a ---------------------------------------------------------------------------------------------------
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

template <bool SGN, bool STAMP>
__global__ void __launch_bounds__(PW) pipe_analyse_kernel(PipeArgs a, int tpw, uint64_t *stamps)
{
	Stamps st;
	stamp_init<STAMP>(st);
	__shared__ __attribute__((aligned(16))) uint32_t otab[64];
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
	const u32x4 *ptab = reinterpret_cast<const u32x4 *>(a.ptab);

	// lane's region of a tile: block row by (0..15), block pair bxp (0..7)
	const int by = tid >> 3, bxp = tid & 7;
	const size_t reg_off = (size_t)(by * 4) * pitch + bxp * 8;
	auto load_rows = [&](int tile, u32x4 r[4]) {
		const uint16_t *p = img + TILE_ORG(a, tile) + reg_off;
#pragma unroll
		for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);
	};
	// pixels before the lane's two blocks (traversal order); a block that opens its tile follows the previous tile's last pixel
	auto load_prev = [&](int tile, const u32x4 &ent, uint32_t &pa, uint32_t &pb) {
		const uint32_t org = TILE_ORG(a, tile);
		const uint32_t porg = tile > 0 ? TILE_ORG(a, tile - 1) + a.tiles.last[TILE_ORIENT(a, tile - 1)] : 0u;
		pa = img[ent.y != 0xFFFFFFFFu ? org + ent.y : porg];
		pb = img[ent.z != 0xFFFFFFFFu ? org + ent.z : porg];
	};
	u32x4 r[4], ent;
	load_rows(t0, r);
	ent = ptab[TILE_ORIENT(a, t0) * PW + tid];
	if (tid < 64) otab[tid] = a.otab[tid];
	uint32_t pvA, pvB;
	load_prev(t0, ent, pvA, pvB);
	__syncthreads();
	stamp<STAMP>(st, 0);

	for (int s = 0; s < nT; s++) {
		const int tile = t0 + s;
		uint32_t dA[8], dB[8];
		const uint32_t e2 = ent.x;
		const int kA = (int)(e2 & 0xFFu), kB = (int)((e2 >> 16) & 0xFFu);
		permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, (const LDS(uint32_t) *)otab + ((e2 >> 8) & 3u) * 16, dA);
		permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, (const LDS(uint32_t) *)otab + ((e2 >> 24) & 3u) * 16, dB);
		const uint32_t cpA = pvA, cpB = pvB;
		if (s + 1 < nT) {  // the next tile's rows are in flight across the analysis below
			load_rows(tile + 1, r);
			ent = ptab[TILE_ORIENT(a, tile + 1) * PW + tid];
		}
		stamp<STAMP>(st, 1);
		uint32_t orall = cpA | cpB;
#pragma unroll
		for (int j = 0; j < 8; j++) orall |= dA[j] | dB[j];
		const bool wide = SGN || __any((orall & 0xC000C000u) != 0);
		bool any_diff = false;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int k = h ? kB : kA;
			const uint32_t *d = h ? dB : dA;
			const bool first_of_slice = (tile == 0 && k == 0);
			const uint32_t pv = first_of_slice ? 0u : (h ? cpB : cpA);  // the slice starts from pixel value 0 (core.py:278)
			uint32_t n2, chg, enter;
			analyse_block<SGN>(d, pv, wide, first_of_slice, n2, chg, enter);
			const bool difficult = seg && chg >= 8u;                      // cluster.py:58
			const int b = tile * 256 + k;
			ssz[b] = (uint8_t)((16u + n2) | (difficult ? 0x80u : 0u));
			if (difficult) gmask[b] = (uint64_t)(chg + enter);            // cluster.py:110; the mask kernel replaces it with the mask
			any_diff |= difficult;
		}
		// tiles the mask kernel has to look at: listed once per slice (the workgroup's two waves may both ask)
		if (__any(any_diff) && lane == 0 && atomicOr(a.tflag + (size_t)sl * NT + tile, 1u) == 0u)
			a.tlist[(size_t)sl * NT + atomicAdd(a.tcount + sl, 1u)] = (uint32_t)tile;
		stamp<STAMP>(st, 2);
		if (s + 1 < nT) load_prev(tile + 1, ent, pvA, pvB);
		stamp<STAMP>(st, 3);
	}
	if (STAMP) { st.acc[5] = st.rt0; st.acc[6] = __builtin_amdgcn_s_memrealtime(); }
	stamp_store<STAMP>(st, stamps, blockIdx.x * 2 + wave);
}

// ---- K1b --------------------------------------------------------------------------------------------------
// MQ workgroups (4 waves) per slice walk the slice's tiles that have difficult blocks (K1a lists them): few, long-lived
// waves, because launching a wave costs ~2 ns chip-wide (measured: kernels of 32768 short waves take >= 65 us whatever
// they do).  A tile's 256 blocks and the first 64 blocks of the next tile -- every candidate a difficult block of this tile can have -- are
// fetched once (lane = traversal block: four 8-byte row segments, cache hits: K1a has just streamed them), put in
// traversal order and kept in LDS (32 bytes per block).  The waves then take the tile's difficult blocks round-robin:
// lane j compares block i + j with block i, whose terms go to SGPRs, and counts the positive jumps of the interleaved
// order A0 B0 A1 B1 ... (cluster.py:131-153).
constexpr int MW = 8, MQ = 8;          // waves per workgroup, workgroups per slice
constexpr int K1B_DLIN = 0;            // 320 blocks x 32 bytes
constexpr int K1B_BTAB = 320 * 32;
constexpr int K1B_OTAB = K1B_BTAB;
constexpr int K1B_LIST = K1B_OTAB + 256;   // 256 u16: block | cur << 8
constexpr int K1B_MISC = K1B_LIST + 512;
constexpr int K1B_LDS = K1B_MISC + 32;

template <bool SGN, bool STAMP>
__global__ void __launch_bounds__(64 * MW) pipe_masks_kernel(PipeArgs a, uint64_t *stamps)
{
	Stamps st;
	stamp_init<STAMP>(st);
	__shared__ __attribute__((aligned(16))) uint8_t smem[K1B_LDS];
	LDS(uint8_t) *dlin = (LDS(uint8_t) *)(smem + K1B_DLIN);
	LDS(uint32_t) *otab = (LDS(uint32_t) *)(smem + K1B_OTAB);
	LDS(uint16_t) *lst = (LDS(uint16_t) *)(smem + K1B_LIST);
	LDS(uint32_t) *misc = (LDS(uint32_t) *)(smem + K1B_MISC);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int NT = a.n_tiles, NB = a.e.NB, N = a.e.N;
	const int sl = blockIdx.x / MQ, q = blockIdx.x % MQ;
	const int ntl = (int)a.tcount[sl];
	if (q >= ntl) return;
	const uint16_t *img = a.e.images + (size_t)sl * N;
	uint64_t *gmask = a.mask + (size_t)sl * NB;
	const int pitch = a.row_pitch;
	if (tid < 64) otab[tid] = a.otab[tid];
	for (int ti = q; ti < ntl; ti += MQ) {
	const int tile = (int)a.tlist[(size_t)sl * NT + ti];  // 32-bit entries: a wave-uniform read becomes a scalar (dword-aligned) load
	const bool has_next = tile + 1 < NT;
	const int to0 = TILE_ORIENT(a, tile), to1 = has_next ? TILE_ORIENT(a, tile + 1) : to0;
	const uint32_t org0 = TILE_ORG(a, tile), org1 = has_next ? TILE_ORG(a, tile + 1) : org0;
	// ---- this lane's block of the tile (and, for the first wave, of the next tile's first quarter)
	// lanes 0..255: the tile's blocks; lanes 256..319: the next tile's first 64 blocks; the rest idle here
	const bool own = tid < 256, nxt = tid >= 256 && tid < 320 && has_next;
	uint32_t bt0 = 0;
	uint2 r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
	if (own || nxt) {
		bt0 = a.btab[(own ? to0 : to1) * 256 + (tid & 255)];
		const uint16_t *p0 = img + (own ? org0 : org1) + (bt0 & 0xFFFFFFu);
		r0 = *reinterpret_cast<const uint2 *>(p0); r1 = *reinterpret_cast<const uint2 *>(p0 + pitch);
		r2 = *reinterpret_cast<const uint2 *>(p0 + 2 * (size_t)pitch); r3 = *reinterpret_cast<const uint2 *>(p0 + 3 * (size_t)pitch);
	}
	// the tile's difficult blocks: one byte per lane; cur waits in the mask slot
	const uint8_t sz = own ? a.ssz[(size_t)sl * NB + tile * 256 + tid] : (uint8_t)0;
	const bool diff = (sz & 0x80u) != 0;
	uint32_t cur = 0;
	if (diff) cur = (uint32_t)gmask[tile * 256 + tid];
	const uint64_t bal = __ballot(diff);
	if (lane == 0) misc[wave] = (uint32_t)__popcll(bal);
	__syncthreads();
	if (tid < 320) {
		uint32_t d[8];
		permute_block(r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y, (const LDS(uint32_t) *)otab + (bt0 >> 24) * 16, d);
		*(LDS(u32x4) *)(dlin + tid * 32) = (u32x4){d[0], d[1], d[2], d[3]};
		*(LDS(u32x4) *)(dlin + tid * 32 + 16) = (u32x4){d[4], d[5], d[6], d[7]};
		uint32_t pos = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
		for (int v = 0; v < 4; v++) if (v < wave) pos += misc[v];
		if (diff) lst[pos] = (uint16_t)(tid | (cur << 8));
	}
	const int total = (int)(misc[0] + misc[1] + misc[2] + misc[3]);
	stamp<STAMP>(st, 0);
	__syncthreads();
	stamp<STAMP>(st, 1);
	// two entries per step: their LDS reads and scalar terms overlap
	auto load_entry = [&](int e, uint32_t bw[8], uint32_t av[8], int &b, uint32_t &ecur, bool &valid) {
		const uint32_t ent = lst[e];
		const int i = (int)(ent & 0xFFu);
		ecur = ent >> 8;
		b = tile * 256 + i;
		valid = lane >= 1 && b + lane < NB;
		const int c = valid ? i + lane : i;
		const LDS(u32x4) *ap = (const LDS(u32x4) *)(dlin + i * 32), *bp = (const LDS(u32x4) *)(dlin + c * 32);
		const u32x4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
		bw[0] = b0.x; bw[1] = b0.y; bw[2] = b0.z; bw[3] = b0.w; bw[4] = b1.x; bw[5] = b1.y; bw[6] = b1.z; bw[7] = b1.w;
		av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
	};
	auto eval_entry = [&](const uint32_t bw[8], const uint32_t av[8], int b, uint32_t ecur, bool valid) {
		uint32_t aw[8];
#pragma unroll
		for (int j = 0; j < 8; j++) aw[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)av[j]);  // block A is wave-uniform
		uint32_t hi_or = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) hi_or |= bw[j] | aw[j];
		const bool small = !SGN && !__any((hi_or & 0xC000C000u) != 0);
		uint32_t up;
		if (small) {
			// up = #(B[t] - A[t] >= 65) + #(A[t+1] - B[t] >= 65); all values < 16384, so packed 16-bit differences
			// are exact: count the NEGATIVE results of B[t] - (A[t] + 65) and A[t+1] - (B[t] + 65).  A's terms are
			// scalar (no carry between the halves below 0x4000); t = 15 has no successor: 0 - (B + 65) < 0
			const uint32_t FIFTEEN = 0x000F000Fu, K65 = 0x00410041u;
			uint32_t neg = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const uint32_t a65 = aw[j] + K65;
				const uint32_t an = (j < 7) ? ((aw[j] >> 16) | (aw[j + 1] << 16)) : (aw[7] >> 16);
				neg = pk_add(neg, pk_lshr(FIFTEEN, pk_sub(bw[j], a65)));
				neg = pk_add(neg, pk_lshr(FIFTEEN, pk_sub(an, pk_add(bw[j], K65))));
			}
			up = 32u - ((neg & 0xFFFFu) + (neg >> 16));
		} else {
			up = 0;
			int bprev = 0;
#pragma unroll
			for (int t = 0; t < 16; t++) {
				int avv = (int)((aw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu), bv = (int)((bw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu);
				if (SGN) { avv = (int)(int16_t)avv; bv = (int)(int16_t)bv; }
				if (t > 0) up += (avv - bprev >= 65) ? 1u : 0u;
				up += (bv - avv >= 65) ? 1u : 0u;
				bprev = bv;
			}
		}
		// cluster.py:153,158: up + 1 < current_delta - 2 in uint32; block 0 wraps: it always fits
		const bool fit = valid && (b == 0 ? true : ((up + 1u) < (ecur - 2u)));
		const uint64_t mk = __ballot(fit);
		if (lane == 0) gmask[b] = mk;
	};
	for (int e = wave; e < total; e += 2 * MW) {
		uint32_t bw0[8], av0[8], bw1[8], av1[8];
		int b0i, b1i = 0;
		uint32_t c0, c1 = 0;
		bool v0, v1 = false;
		const bool two = e + MW < total;
		load_entry(e, bw0, av0, b0i, c0, v0);
		if (two) load_entry(e + MW, bw1, av1, b1i, c1, v1);
		eval_entry(bw0, av0, b0i, c0, v0);
		if (two) eval_entry(bw1, av1, b1i, c1, v1);
	}
	stamp<STAMP>(st, 2);
	__syncthreads();  // the staged tile and the list are reused
	}
	if (STAMP) { st.acc[5] = st.rt0; st.acc[6] = __builtin_amdgcn_s_memrealtime(); }
	stamp_store<STAMP>(st, stamps, blockIdx.x * MW + wave);
}

// ---- K2 ---------------------------------------------------------------------------------------------------
constexpr int K2T = 256;
constexpr int K2_CAP = 4096;                      // difficult-list records kept in LDS
constexpr int K2_ROLE = 0;                        // PIPE_MAX_NB bytes
constexpr int K2_IDX = PIPE_MAX_NB;               // K2_CAP u32
constexpr int K2_MASK = K2_IDX + K2_CAP * 4;      // K2_CAP u64
constexpr int K2_TSUM = K2_MASK + K2_CAP * 8;     // 512 i32 (sum of single sizes + corrections per half tile)
constexpr int K2_MISC = K2_TSUM + 512 * 4;
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

template <bool STAMP>
__global__ void __launch_bounds__(K2T) pipe_resolve_kernel(PipeArgs a, uint64_t *stamps)
{
	Stamps st;
	stamp_init<STAMP>(st);
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
	for (int i = tid; i < 2 * NT; i += K2T) tsum[i] = 0;
	if (tid < 8) misc[tid] = 0;
	__syncthreads();
	stamp<STAMP>(st, 0);

	// ---- ordered list of difficult blocks + sum of the single sizes per tile
	const int n16 = NB / 16;                              // 16 blocks per chunk, 16 chunks per tile
	const int cpl = (n16 + K2T - 1) / K2T;                // chunks per lane (contiguous ranges keep the order)
	uint32_t cnt = 0;
	// a lane's chunks, all requested before the first is used and kept for the second walk below (a rolled loop waits for
	// every load where it is issued: four memory round trips in a row, twice).  Up to 512x512 a lane owns at most four chunks.
	constexpr int CPL_REG = 4;
	const bool in_regs = cpl <= CPL_REG;
	const int c_lo = tid * cpl, c_hi = min(n16, (tid + 1) * cpl);
	u32x4 vr[CPL_REG];
	if (in_regs) {
#pragma unroll
		for (int k = 0; k < CPL_REG; k++) vr[k] = *reinterpret_cast<const u32x4 *>(ssz + (size_t)min(c_lo + k, n16 - 1) * 16);
	}
	auto chunk_of = [&](int c, int k) -> u32x4 { return in_regs ? vr[k] : *reinterpret_cast<const u32x4 *>(ssz + (size_t)c * 16); };
	auto count_chunk = [&](int c, const u32x4 &v) {
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		uint32_t sum = 0;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			cnt += (uint32_t)__popc(w[q] & 0x80808080u);
			sum = __builtin_amdgcn_udot4(w[q] & 0x7F7F7F7Fu, 0x01010101u, sum, false);
		}
		lds_add(&tsum[c >> 3], (int32_t)sum);
	};
	if (in_regs) {
#pragma unroll
		for (int k = 0; k < CPL_REG; k++) if (c_lo + k < c_hi) count_chunk(c_lo + k, vr[k]);
	} else for (int c = c_lo; c < c_hi; c++) count_chunk(c, chunk_of(c, 0));
	uint32_t ndiff;
	uint32_t pos = wg_incl_scan256(cnt, misc, tid, ndiff) - cnt;
	stamp<STAMP>(st, 1);
	if (seg && ndiff) {
		auto list_chunk = [&](int c, const u32x4 &v) {
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				uint32_t m = w[q] & 0x80808080u;
				while (m) {
					const int bit = __ffs((int)m) - 1;
					m &= m - 1;
					const uint32_t b = (uint32_t)(c * 16 + q * 4 + (bit >> 3));
					if (pos < K2_CAP) l_idx[pos] = b; else g_idx[pos] = b;
					pos++;
				}
			}
		};
		if (in_regs) {
#pragma unroll
			for (int k = 0; k < CPL_REG; k++) if (c_lo + k < c_hi) list_chunk(c_lo + k, vr[k]);
		} else for (int c = c_lo; c < c_hi; c++) list_chunk(c, chunk_of(c, 0));
		__syncthreads();
		// all masks with independent loads, eight in flight per lane
		const uint32_t nm = min(ndiff, (uint32_t)K2_CAP);
		for (uint32_t e0 = tid; e0 < nm; e0 += K2T * 8) {
			uint64_t mv[8];
#pragma unroll
			for (int u = 0; u < 8; u++) mv[u] = gmask[l_idx[min(e0 + (uint32_t)u * K2T, nm - 1)]];
#pragma unroll
			for (int u = 0; u < 8; u++) if (e0 + (uint32_t)u * K2T < nm) l_mask[e0 + (uint32_t)u * K2T] = mv[u];
		}
	}
	__syncthreads();
	stamp<STAMP>(st, 2);

	// ---- resolve: greedy first fit (cluster.py:79-190).  A block whose mask is empty can neither take a partner nor
	// change the state of the walk (being taken is decided by its leader), so only the others are walked: fewer, shorter
	// islands (two of them more than 63 blocks apart cannot influence each other), one lane per island
	uint32_t nwalk = 0;
	if (seg && ndiff) {
		const uint32_t chunk = (ndiff + K2T - 1) / K2T;  // contiguous entries per lane keep the order
		const uint32_t lo = min(ndiff, tid * chunk), hi = min(ndiff, lo + chunk);
		uint32_t nz = 0;
		for (uint32_t e = lo; e < hi; e++) nz += mask_of(e, idx_of(e)) != 0 ? 1u : 0u;
		uint32_t wpos = wg_incl_scan256(nz, misc, tid, nwalk) - nz;
		// compaction in place: entry e moves to wpos <= e; every lane reads its own range before anyone writes behind it
		uint32_t ki[8]; uint64_t km[8];
		const bool fits = chunk <= 8;
		if (fits) {
#pragma unroll
			for (int u = 0; u < 8; u++) { const uint32_t e = lo + u; ki[u] = e < hi ? idx_of(e) : 0u; km[u] = e < hi ? mask_of(e, ki[u]) : 0ull; }
		}
		__syncthreads();
		if (fits) {
#pragma unroll
			for (int u = 0; u < 8; u++) if (km[u] != 0) { l_idx[wpos] = ki[u]; l_mask[wpos] = km[u]; wpos++; }
		} else nwalk = ndiff;  // more than 2048 difficult blocks: walk them all (correct, slower)
		__syncthreads();
		for (uint32_t e0 = tid; e0 < nwalk; e0 += K2T) {
			const uint32_t i0 = idx_of(e0);
			if (e0 > 0 && i0 - idx_of(e0 - 1) <= 63u) continue;  // not the head of an island
			uint64_t cw = 0;                                    // blocks at i + bit already consumed as partners
			uint32_t e = e0, i = i0;
			uint64_t mk = mask_of(e0, i0);
			for (;;) {
				// the next record is fetched while this one is decided
				const bool more = e + 1 < nwalk;
				const uint32_t inext = more ? idx_of(e + 1) : 0u;
				const uint64_t mnext = more ? mask_of(e + 1, inext) : 0ull;
				if (!(cw & 1ull)) {
					const uint64_t avail = mk & ~cw & ~1ull;
					if (avail) {
						const int j = __ffsll((long long)avail) - 1;
						role[i] = (uint8_t)j;
						role[i + j] = ROLE_PARTNER;
						cw |= 1ull << j;
					}
				}
				if (!more || inext - i > 63u) break;
				cw >>= (inext - i);
				i = inext; mk = mnext; e++;
			}
		}
	}
	__syncthreads();
	stamp<STAMP>(st, 3);

	// ---- meshed pairs: their token bytes (core.py:281-323 along the interleaved order of cluster.py:173-174), and the
	// blocks that follow a meshed block: their predecessor pixel is the last pixel of the previous GROUP
	// traversal position of the last pixel written before block b's group (b > 0, b not a partner)
	auto true_prev_pos = [&](int b) -> int {
		int q = b - 1;
		int rq = role[q];
		if (rq != 0) {
			while (rq == ROLE_PARTNER) { q--; rq = role[q]; }
			if (rq != 0) q += rq;  // a pair ends with its partner's last pixel
		}
		return q * 16 + 15;
	};
	uint32_t q7 = 0;
	if (seg) {
		for (uint32_t e = tid; e < nwalk; e += K2T) {  // every leader has a non-empty mask
			const int i = (int)idx_of(e);
			const int r = role[i];
			if (r == 0 || r == ROLE_PARTNER) continue;
			const int p = i + r;
			const uint32_t slot = lds_add(&misc[4], 1u);
			uint8_t *out = rec + (size_t)slot * PIPE_PAIR_REC;
			// every position first, then every pixel: two round trips for the whole pair and its two followers
			const int bs[2] = {i + 1, p + 1};
			bool follow[2];
			int fpos[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				follow[h] = bs[h] < NB && role[bs[h] < NB ? bs[h] : 0] == 0;
				fpos[h] = follow[h] ? true_prev_pos(bs[h]) : 0;
			}
			int prev = i > 0 ? PX(true_prev_pos(i)) : 0;
			int va[16], vb[16];
#pragma unroll
			for (int t = 0; t < 16; t++) { va[t] = PX(i * 16 + t); vb[t] = PX(p * 16 + t); }
			int ftp[2], fdp[2], fv0[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int b = follow[h] ? bs[h] : 1;
				ftp[h] = PX(fpos[h]); fdp[h] = PX(b * 16 - 1); fv0[h] = PX(b * 16);
			}
			const int sz_i = ssz[i] & 0x7F, sz_p = ssz[p] & 0x7F;
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
			lds_add(&tsum[i >> 7], n - sz_i);
			lds_add(&tsum[p >> 7], -sz_p);
			// blocks after the leader and after the partner, if emitted alone
#pragma unroll
			for (int h = 0; h < 2; h++) {
				if (!follow[h]) continue;
				spec[bs[h]] = (uint32_t)ftp[h];
				const int corr = (int)tok_two(fv0[h] - ftp[h]) - (int)tok_two(fv0[h] - fdp[h]);
				if (corr) lds_add(&tsum[bs[h] >> 7], corr);
			}
		}
	}
	if (q7) lds_or(&misc[5], CCT_ST_Q7);
	__syncthreads();
	stamp<STAMP>(st, 4);

	// ---- roles to HBM (K3 and the caller), tile offsets, slice size, statistics
	uint8_t *groles = a.roles + (size_t)sl * NB;
	uint8_t *oroles = a.e.roles_out ? a.e.roles_out + (size_t)sl * NB : nullptr;
	for (int i = tid; i < NB / 16; i += K2T) {
		const u32x4 v = *(const LDS(u32x4) *)(role + i * 16);
		*reinterpret_cast<u32x4 *>(groles + (size_t)i * 16) = v;
		if (oroles) *reinterpret_cast<u32x4 *>(oroles + (size_t)i * 16) = v;
	}
	uint32_t *toff = a.toff + (size_t)sl * (2 * NT + 1);  // payload offset of every half tile (128 blocks)
	uint32_t total;
	{
		const uint32_t m0 = tid < NT ? (uint32_t)tsum[2 * tid] : 0u, m1 = tid < NT ? (uint32_t)tsum[2 * tid + 1] : 0u;
		const uint32_t inc = wg_incl_scan256(m0 + m1, misc, tid, total);
		if (tid < NT) { toff[2 * tid] = inc - m0 - m1; toff[2 * tid + 1] = inc - m1; }
		if (tid == 0) toff[2 * NT] = total;
	}
	if (tid == 0) {
		const uint32_t njump = misc[4];
		const uint32_t size = total + (a.e.eof >= 0 ? 1u : 0u);
		const bool cap = (size_t)((size + 15u) & ~15u) > a.e.stride;
		a.e.sizes[sl] = cap ? 0u : size;
		a.e.status[sl] = misc[5] | (cap ? CCT_ST_CAP : 0u);
		if (a.e.stats) {
			uint32_t *sts = a.e.stats + (size_t)sl * 4;
			const uint32_t nfull = total - (uint32_t)N - njump;
			sts[0] = (uint32_t)N - nfull; sts[1] = nfull; sts[2] = njump; sts[3] = seg ? ndiff : 0u;
		}
	}
	stamp<STAMP>(st, 5);
	stamp_store<STAMP>(st, stamps, blockIdx.x * 4 + (tid >> 6));
}

// ---- K3 ---------------------------------------------------------------------------------------------------
// Measured (tools/sweep_pipe.py, CCT_K3_EXTRA_LDS): whatever a workgroup declares, about 96 KB of LDS per CU are in use at a
// time, so LDS per wave decides the residency.  The payload image therefore holds what real slices need (a half tile of
// 2048 pixels at up to 1.45 bytes per pixel); a half tile that needs more (up to 6.2 KB: dense meshes of noise) is written
// byte by byte to HBM instead (emit_block_bytes): exact, slow, rare.
constexpr int HSTG_BYTES = 3072;                  // payload image of a half tile
constexpr int HSTG_FIT = HSTG_BYTES - 32;         // head + tokens + EOF must stay below (the ORs touch up to 11 bytes more)
constexpr int K3_STG = 0;                         // HSTG_BYTES
constexpr int K3_TTAB = HSTG_BYTES;               // 16 entries x 16 bytes
constexpr int K3_OTAB = K3_TTAB + 256;            // 256
constexpr int K3_SZL = K3_OTAB + 256;             // 128 token sizes of the half tile's blocks, traversal order
constexpr int K3_OFFL = K3_SZL + 128;             // 128 u16: their offsets
constexpr int K3_PAIR = K3_OFFL + 256;            // 128 x 8: leaders (offset, record) -- shares its bytes with ...
constexpr int K3_SPEC = K3_PAIR;                  // ... 128 u32 spec words, which are in registers before the first leader is listed
constexpr int K3_LAST = K3_PAIR + 128 * 8;        // 129 u16
constexpr int K3_ROLE = K3_LAST + 272 + 1;        // 130 bytes from an ODD address: [1 + 2 lane] pairs are then 2-byte aligned
constexpr int K3_LDS = K3_ROLE + 143;

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
		const u32x4 te = *(const LDS(u32x4) *)(ttab + (ti >> 3));
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

// tokens of one block straight to HBM, one byte at a time (oversized half tiles only)
__device__ __forceinline__ void emit_block_bytes(const uint32_t d[8], uint32_t pv, uint8_t *out)
{
	int pu = (int)(pv & 0xFFFFu);
	for (int i = 0; i < 16; i++) {
		const int v = px16(d, i), dlt = v - pu;
		if (tok_two(dlt)) { *out++ = (uint8_t)(0xE0 | ((dlt >> 8) & 0x0F)); *out++ = (uint8_t)(dlt & 0xFF); }
		else *out++ = (uint8_t)(dlt & 0x7F);
		pu = v;
	}
}

// One wave per half tile (128 traversal blocks = a 64x32 or a 32x64 pixel region; lane = a pair of blocks as before).
// No workgroup barrier: sizes, offsets and the payload image are private to the wave, LDS accesses of a wave execute in
// order, and wave_fence() keeps the compiler from moving them across the phase boundaries.  Every global load is
// issued up front: pixel rows, the half tile's roles and spec words (coalesced, into LDS) and the one pixel before the
// half tile; pixels before the other blocks are exchanged through LDS after the permute.
__device__ __forceinline__ void wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <bool STAMP>
__global__ void __launch_bounds__(64) pipe_pack_kernel(PipeArgs a, uint64_t *stamps)
{
	Stamps st;
	stamp_init<STAMP>(st);
	__shared__ __attribute__((aligned(16))) uint8_t smem[K3_LDS];
	LDS(uint8_t) *stg = (LDS(uint8_t) *)(smem + K3_STG);
	LDS(uint8_t) *ttab = (LDS(uint8_t) *)(smem + K3_TTAB);
	LDS(uint32_t) *otab = (LDS(uint32_t) *)(smem + K3_OTAB);
	LDS(uint8_t) *szl = (LDS(uint8_t) *)(smem + K3_SZL);
	LDS(uint16_t) *offl = (LDS(uint16_t) *)(smem + K3_OFFL);
	LDS(uint32_t) *pairs = (LDS(uint32_t) *)(smem + K3_PAIR);
	LDS(uint8_t) *rl = (LDS(uint8_t) *)(smem + K3_ROLE);       // [0] role of the block before the half tile, [1 + k] role of block k
	LDS(uint32_t) *specl = (LDS(uint32_t) *)(smem + K3_SPEC);
	LDS(uint16_t) *lastpx = (LDS(uint16_t) *)(smem + K3_LAST);  // [0] pixel before the half tile, [1 + k] last pixel of block k

	const int lane = threadIdx.x;
	const int NT = a.n_tiles, NB = a.e.NB, N = a.e.N;
	const int NH = 2 * NT;
	const int sl = blockIdx.x / NH, ht = blockIdx.x % NH;
	const int tile = ht >> 1, half = ht & 1;
	const int hb = ht * 128;  // first block of the half tile in the slice
	const uint16_t *img = a.e.images + (size_t)sl * N;
	const int pitch = a.row_pitch;
	const int to = TILE_ORIENT(a, tile);

	// ---- every global load, issued together
	const u32x4 ent = reinterpret_cast<const u32x4 *>(a.ptab2)[(to * 2 + half) * 64 + lane];
	const uint32_t org = TILE_ORG(a, tile);
	uint32_t before_addr = 0;  // the pixel before the half tile (none: the slice starts from pixel value 0, core.py:278)
	if (half) before_addr = org + a.tiles.mid[to];
	else if (tile > 0) before_addr = TILE_ORG(a, tile - 1) + a.tiles.last[TILE_ORIENT(a, tile - 1)];
	const uint32_t *toff = a.toff + (size_t)sl * (NH + 1);
	const uint32_t off_t = toff[ht], off_n = toff[ht + 1];
	const uint8_t *roles = a.roles + (size_t)sl * NB + hb;
	const uint32_t *spec = a.spec + (size_t)sl * NB + hb;
	const uint32_t role2 = *reinterpret_cast<const uint16_t *>(roles + 2 * lane);
	const uint2 spec2 = *reinterpret_cast<const uint2 *>(spec + 2 * lane);
	// lane 0: role of the block before the half tile | pixel before the half tile << 16.  Read by every lane (one address, one
	// transaction) from a clamped address: a load under `if (lane == 0)` is waited for on the spot, before the rows are requested
	const uint32_t edge_role = roles[hb > 0 ? -1 : 0], edge_px = img[before_addr];
	const uint32_t edge = (hb > 0 ? edge_role : 0u) | (ht > 0 ? edge_px << 16 : 0u);
	const uint32_t geom = a.tiles.geom[to * 2 + half];
	const uint32_t reg_off = (geom & 1u) ? (uint32_t)((lane >> 2) * 4 * pitch) + ((geom >> 8) + (uint32_t)(lane & 3)) * 8u
	                                     : ((geom >> 8) + (uint32_t)(lane >> 3)) * 4u * (uint32_t)pitch + (uint32_t)(lane & 7) * 8u;
	const uint16_t *p = img + org + reg_off;
	u32x4 r[4];
#pragma unroll
	for (int q = 0; q < 4; q++) r[q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);

	// ---- tables and the zeroed payload image
	otab[lane] = a.otab[lane];
	if (lane < 16) *(LDS(u32x4) *)(ttab + lane * 16) = reinterpret_cast<const u32x4 *>(a.ttab)[lane];
	for (int i = lane; i < HSTG_BYTES / 16; i += 64) *(LDS(u32x4) *)(stg + i * 16) = (u32x4){0, 0, 0, 0};
	*(LDS(uint16_t) *)(rl + 1 + 2 * lane) = (uint16_t)role2;  // rl[1 + 2 lane], rl[2 + 2 lane] at an even address: rl starts odd
	specl[2 * lane] = spec2.x;
	specl[2 * lane + 1] = spec2.y;
	if (lane == 0) { rl[0] = (uint8_t)edge; lastpx[0] = (uint16_t)(edge >> 16); }
	wave_fence();
	stamp<STAMP>(st, 0);

	// ---- traversal order inside the lane's two blocks; pixels before them through LDS
	const int hk = half * 128;
	const int kA = (int)(ent.x & 0xFFu) - hk, kB = (int)((ent.x >> 16) & 0xFFu) - hk;  // block inside the half tile
	uint32_t dA[8], dB[8], xA[8], xB[8], mA[4], mB[4];
	permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, otab + ((ent.x >> 8) & 3u) * 16, dA);
	permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, otab + ((ent.x >> 24) & 3u) * 16, dB);
	lastpx[1 + kA] = (uint16_t)(dA[7] >> 16);
	lastpx[1 + kB] = (uint16_t)(dB[7] >> 16);
	wave_fence();
	const uint32_t roleA = rl[1 + kA], roleB = rl[1 + kB], rolepA = rl[kA], rolepB = rl[kB];
	// special blocks: leaders carry their record and size, blocks after a meshed block their predecessor pixel
	const bool leadA = roleA >= 1 && roleA <= 63, leadB = roleB >= 1 && roleB <= 63;
	const bool succA = roleA == 0 && rolepA != 0, succB = roleB == 0 && rolepB != 0;
	const uint32_t spA = specl[kA], spB = specl[kB];
	const uint32_t pvA = succA ? spA : lastpx[kA], pvB = succB ? spB : lastpx[kB];
	deltas16(dA, pvA, xA);
	deltas16(dB, pvB, xB);
	uint32_t orall = pvA | pvB;
#pragma unroll
	for (int j = 0; j < 8; j++) orall |= dA[j] | dB[j];
	bool q7 = false;
	uint32_t n2A, n2B;
	if (!__any((orall & 0xC000C000u) != 0)) {
		n2A = group_masks(xA, mA);
		n2B = group_masks(xB, mB);
		if (__any((orall & 0xF800F800u) != 0)) {  // a delta outside [-2047, 2048] needs a pixel >= 2048
			const uint32_t K2048 = 0x08000800u, K2047 = 0x07FF07FFu;
			uint32_t bad0 = 0, bad1 = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				bad0 |= pk_sub(K2048, xA[j]) | pk_add(xA[j], K2047);
				bad1 |= pk_sub(K2048, xB[j]) | pk_add(xB[j], K2047);
			}
			q7 = (roleA == 0 && (bad0 & 0x80008000u)) || (roleB == 0 && (bad1 & 0x80008000u));
		}
	} else {
		bool qa = false, qb = false;
		n2A = group_masks_wide(dA, pvA, mA, qa);
		n2B = group_masks_wide(dB, pvB, mB, qb);
		q7 = (roleA == 0 && qa) || (roleB == 0 && qb);
	}
	const uint32_t szA = roleA == 0 ? 16u + n2A : (leadA ? (spA & 0xFFu) : 0u);
	const uint32_t szB = roleB == 0 ? 16u + n2B : (leadB ? (spB & 0xFFu) : 0u);
	szl[kA] = (uint8_t)szA;
	szl[kB] = (uint8_t)szB;
	wave_fence();
	stamp<STAMP>(st, 1);
	// ---- offsets: scan of the 128 sizes, two blocks per lane
	uint32_t tot;
	{
		const uint32_t v = *(const LDS(uint16_t) *)(szl + lane * 2);
		const uint32_t s0 = v & 0xFFu, s1 = v >> 8;
		const uint32_t inc = wave_incl_scan(s0 + s1);
		tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
		const uint32_t e0 = inc - s0 - s1;
		*(LDS(uint32_t) *)(offl + lane * 2) = e0 | ((e0 + s0) << 16);
	}
	wave_fence();
	const uint32_t head = off_t & 15u;
	const uint32_t oA = head + offl[kA], oB = head + offl[kB];
	stamp<STAMP>(st, 2);

	const bool last = ht == NH - 1;
	uint32_t stat = q7 ? CCT_ST_Q7 : 0u;
	if (tot != off_n - off_t) stat |= CCT_ST_INTERNAL;
	if (head + tot + 1u > (uint32_t)HSTG_FIT) {
		// ---- oversized half tile: every lane writes its blocks (and its pair records) byte by byte
		const uint32_t end = tot + ((last && a.e.eof >= 0) ? 1u : 0u);
		const bool room = (size_t)off_t + ((end + 15u) & ~15u) + 16u <= a.e.stride;
		uint8_t *out = a.e.payload + (size_t)sl * a.e.stride + off_t;
		if (room) {
			const uint8_t *rec = a.pairrec + (size_t)sl * (NB / 2) * PIPE_PAIR_REC;
			if (roleA == 0) emit_block_bytes(dA, pvA, out + (oA - head));
			if (roleB == 0) emit_block_bytes(dB, pvB, out + (oB - head));
			if (leadA) { const uint8_t *src = rec + (size_t)(spA >> 8) * PIPE_PAIR_REC; for (uint32_t j = 0; j < (spA & 0xFFu); j++) out[oA - head + j] = src[j]; }
			if (leadB) { const uint8_t *src = rec + (size_t)(spB >> 8) * PIPE_PAIR_REC; for (uint32_t j = 0; j < (spB & 0xFFu); j++) out[oB - head + j] = src[j]; }
			if (last && lane < 16) {  // EOF (core.py:329-330) and the zero padding of the slice's last 16 bytes
				const uint32_t i = tot + (uint32_t)lane;
				if (lane == 0 && a.e.eof >= 0) out[i] = (uint8_t)a.e.eof;
				else if (i >= end && ((size_t)off_t + i) < (((size_t)off_t + end + 15u) & ~(size_t)15u)) out[i] = 0;
			}
		} else stat |= CCT_ST_CAP;
		if (__any(stat != 0)) { if (stat) atomicOr(a.e.status + sl, stat); }
		stamp_store<STAMP>(st, stamps, blockIdx.x);
		return;
	}
	// ---- tokens into the payload image
	if (roleA == 0) emit_block(xA, mA, oA, stg, ttab);
	if (roleB == 0) emit_block(xB, mB, oB, stg, ttab);
	// ---- meshed pairs: K2 left their bytes in HBM records; the wave copies them one after the other
	{
		const uint64_t lbA = __ballot(leadA), lbB = __ballot(leadB);
		const uint32_t nA = (uint32_t)__popcll(lbA);
		const uint64_t below = (1ull << lane) - 1ull;
		if (leadA) { const uint32_t e = (uint32_t)__popcll(lbA & below); pairs[2 * e] = oA; pairs[2 * e + 1] = spA; }
		if (leadB) { const uint32_t e = nA + (uint32_t)__popcll(lbB & below); pairs[2 * e] = oB; pairs[2 * e + 1] = spB; }
		const uint32_t npair = nA + (uint32_t)__popcll(lbB);
		wave_fence();  // the token ORs and the list are in LDS before record bytes are stored beside them
		if (npair) {
			const uint8_t *rec = a.pairrec + (size_t)sl * (NB / 2) * PIPE_PAIR_REC;
			for (uint32_t e = 0; e < npair; e++) {
				const uint32_t o = pairs[2 * e], sp = pairs[2 * e + 1];
				const uint32_t n = sp & 0xFFu;
				const uint8_t *src = rec + (size_t)(sp >> 8) * PIPE_PAIR_REC;
				for (uint32_t j = lane; j < n; j += 64) stg[o + j] = src[j];
			}
		}
	}
	if (last && a.e.eof >= 0 && lane == 0) stg[head + tot] = (uint8_t)a.e.eof;  // core.py:329-330
	wave_fence();
	stamp<STAMP>(st, 3);
	// ---- flush: whole 16-byte chunks with one store, the two ends shared with the neighbouring half tiles byte by byte
	{
		const uint32_t end = head + tot + ((last && a.e.eof >= 0) ? 1u : 0u);
		const size_t base = (size_t)(off_t & ~15u);
		const bool room = base + ((end + 15u) & ~15u) <= a.e.stride;
		uint8_t *out = a.e.payload + (size_t)sl * a.e.stride + base;
		const uint32_t c_first = head ? 1u : 0u;                        // chunk 0 is partial when head > 0
		const uint32_t c_end = last ? (end + 15u) / 16u : end / 16u;    // the last half tile owns its padding
		if (room) {
			for (uint32_t c = c_first + lane; c < c_end; c += 64)
				*reinterpret_cast<u32x4 *>(out + (size_t)c * 16) = *(const LDS(u32x4) *)(stg + c * 16);
			if (head && lane < 16 && (uint32_t)lane >= head && (uint32_t)lane < end) out[lane] = stg[lane];
			if (!last && lane >= 16 && lane < 32) {
				const uint32_t i = c_end * 16u + (uint32_t)(lane - 16);
				if (i < end && (i >= 16u || !head)) out[i] = stg[i];
			}
		} else stat |= CCT_ST_CAP;
	}
	if (__any(stat != 0)) { if (stat) atomicOr(a.e.status + sl, stat); }
	stamp<STAMP>(st, 4);
	if (STAMP) { st.acc[5] = st.rt0; st.acc[6] = __builtin_amdgcn_s_memrealtime(); }
	stamp_store<STAMP>(st, stamps, blockIdx.x);
}

}  // namespace

// diagnostic build: CCT_PIPE_STAMPS=1 runs the stamped instantiations once and prints the mean cycles per phase
static void report_stamps(const char *name, const uint64_t *d_buf, size_t nslots, const char *const *phase)
{
	std::vector<uint64_t> h(nslots * 8);
	if (hipMemcpy(h.data(), d_buf, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
	double sum[8] = {0};
	for (size_t i = 0; i < nslots; i++) for (int p = 0; p < 8; p++) sum[p] += (double)h[i * 8 + p];
	double all = 0;
	for (int p = 0; p < 7; p++) all += sum[p];
	fprintf(stderr, "[stamps] %s: %.0f shader cycles per wave at %.2f GHz:", name, all / nslots, sum[7] / nslots / 1000.0 * 0.1);
	for (int p = 0; p < 7; p++) if (phase[p]) fprintf(stderr, "  %s %.0f", phase[p], sum[p] / nslots);
	fprintf(stderr, "\n");
}

static void report_timeline(const char *name, const uint64_t *d_buf, size_t nslots)
{
	std::vector<uint64_t> h(nslots * 8);
	if (hipMemcpy(h.data(), d_buf, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
	uint64_t t0 = ~0ull;
	for (size_t i = 0; i < nslots; i++) if (h[i * 8 + 5]) t0 = std::min(t0, h[i * 8 + 5]);
	std::vector<double> st_, en_, lf_;
	double life = 0;
	for (size_t i = 0; i < nslots; i++) {
		if (!h[i * 8 + 5]) continue;  // a wave that left before its first stamp
		st_.push_back((h[i * 8 + 5] - t0) * 0.01); en_.push_back((h[i * 8 + 6] - t0) * 0.01);
		lf_.push_back(en_.back() - st_.back()); life += lf_.back();
	}
	if (st_.empty()) return;
	std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end()); std::sort(lf_.begin(), lf_.end());
	const size_t m = st_.size();
	fprintf(stderr, "[timeline] %s: life (us) p50 %.1f p90 %.1f p99 %.1f max %.1f\n", name, lf_[m / 2], lf_[m * 9 / 10], lf_[m * 99 / 100], lf_[m - 1]);
	fprintf(stderr, "[timeline] %s: %zu waves, mean life %.2f us, mean concurrency %.0f waves | starts (us) p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f | ends p0 %.1f p50 %.1f p100 %.1f\n",
	        name, m, life / m, life / en_[m - 1], st_[0], st_[m / 4], st_[m / 2], st_[3 * m / 4], st_[m - 1], en_[0], en_[m / 2], en_[m - 1]);
}

hipError_t launch_encode_pipe(const PipeArgs &pa, int n, hipStream_t s, const PipeTune *tune)
{
	const int NT = pa.n_tiles;
	int tpw = NT >= 64 ? 8 : (NT >= 16 ? 4 : 1);  // tiles per K1a workgroup
	if (tune && tune->tpw > 0) tpw = std::min(tune->tpw, NT);
	const int wps = (NT + tpw - 1) / tpw;
	const bool sg = (pa.e.flags & CCT_FLAG_SIGNED_SEG) != 0;
	const bool seg = (pa.e.flags & CCT_FLAG_SEGMENTATION) != 0;
	static const bool stamps_on = getenv("CCT_PIPE_STAMPS") != nullptr;
	uint64_t *d_st = nullptr;
	const size_t slots1 = (size_t)n * wps * 2, slots3 = (size_t)n * NT * 2;
	if (stamps_on) {
		if (hipMalloc(&d_st, std::max(std::max(slots1, slots3), (size_t)n * NT * MW) * 64) != hipSuccess) return hipErrorOutOfMemory;
		(void)hipMemset(d_st, 0, std::max(std::max(slots1, slots3), (size_t)n * NT * MW) * 64);
	}
	if (stamps_on) {
		int b1 = 0, b2 = 0, b3 = 0, b4 = 0;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, (pipe_analyse_kernel<false, false>), PW, 0);
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2, (pipe_masks_kernel<false, false>), 64 * MW, 0);
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b3, pipe_resolve_kernel<false>, K2T, K2_LDS);
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4, pipe_pack_kernel<false>, 64, 0);
		fprintf(stderr, "[occupancy API] workgroups per CU: analyse %d (x2 waves)  masks %d (x%d waves)  resolve %d  pack %d (x1 wave)\n", b1, b2, MW, b3, b4);
	}
	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	const bool timing = tune && tune->times_us;
	if (timing) for (auto &e : ev) if (hipEventCreate(&e) != hipSuccess) return hipErrorUnknown;
	if (timing) (void)hipEventRecord(ev[0], s);
	{
		hipError_t e0 = hipMemsetAsync(pa.tflag, 0, (size_t)n * (NT + 1) * sizeof(uint32_t), s);  // flags, then the per-slice counts
		if (e0 != hipSuccess) return e0;
	}
	if (stamps_on) {
		if (sg) hipLaunchKernelGGL((pipe_analyse_kernel<true, true>), dim3(n * wps), dim3(PW), 0, s, pa, tpw, d_st);
		else hipLaunchKernelGGL((pipe_analyse_kernel<false, true>), dim3(n * wps), dim3(PW), 0, s, pa, tpw, d_st);
		(void)hipStreamSynchronize(s);
		static const char *const ph1[8] = {"setup", "perm+prefetch", "analysis", "prev-px", nullptr, nullptr, nullptr, nullptr};
		report_stamps("K1a analyse", d_st, slots1, ph1);
		report_timeline("K1a analyse", d_st, slots1);
	} else if (sg) hipLaunchKernelGGL((pipe_analyse_kernel<true, false>), dim3(n * wps), dim3(PW), 0, s, pa, tpw, (uint64_t *)nullptr);
	else hipLaunchKernelGGL((pipe_analyse_kernel<false, false>), dim3(n * wps), dim3(PW), 0, s, pa, tpw, (uint64_t *)nullptr);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	if (timing) (void)hipEventRecord(ev[1], s);
	if (seg) {
		if (stamps_on) {
			(void)hipMemsetAsync(d_st, 0, (size_t)n * MQ * MW * 64, s);
			hipLaunchKernelGGL((pipe_masks_kernel<false, true>), dim3(n * MQ), dim3(64 * MW), 0, s, pa, d_st);
			(void)hipStreamSynchronize(s);
			static const char *const ph[8] = {"fetch+stage", "sync", "entries", nullptr, nullptr, nullptr, nullptr, nullptr};
			report_stamps("K1b masks", d_st, (size_t)n * MQ * MW, ph);
			report_timeline("K1b masks", d_st, (size_t)n * MQ * MW);
		} else if (sg) hipLaunchKernelGGL((pipe_masks_kernel<true, false>), dim3(n * MQ), dim3(64 * MW), 0, s, pa, (uint64_t *)nullptr);
		else hipLaunchKernelGGL((pipe_masks_kernel<false, false>), dim3(n * MQ), dim3(64 * MW), 0, s, pa, (uint64_t *)nullptr);
		e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	if (timing) (void)hipEventRecord(ev[2], s);
	static bool attr_set = false;
	if (!attr_set) {
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(pipe_resolve_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, K2_LDS);
		if (e != hipSuccess) return e;
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(pipe_resolve_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, K2_LDS);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	if (stamps_on) {
		(void)hipMemsetAsync(d_st, 0, (size_t)n * 4 * 64, s);
		hipLaunchKernelGGL(pipe_resolve_kernel<true>, dim3(n), dim3(K2T), K2_LDS, s, pa, d_st);
		(void)hipStreamSynchronize(s);
		static const char *const ph2[8] = {"init", "count+scan", "list+masks", "resolve", "pairs", "roles+offsets", nullptr, nullptr};
		report_stamps("K2 resolve", d_st, (size_t)n * 4, ph2);
	} else hipLaunchKernelGGL(pipe_resolve_kernel<false>, dim3(n), dim3(K2T), K2_LDS, s, pa, (uint64_t *)nullptr);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	if (timing) (void)hipEventRecord(ev[3], s);
	if (stamps_on) {
		(void)hipMemsetAsync(d_st, 0, slots3 * 64, s);
		hipLaunchKernelGGL(pipe_pack_kernel<true>, dim3(n * NT * 2), dim3(64), 0, s, pa, d_st);
		(void)hipStreamSynchronize(s);
		static const char *const ph3[8] = {"loads+tables", "perm+sizes", "scan+offsets", "emit+pairs", "flush", nullptr, nullptr, nullptr};
		report_stamps("K3 pack", d_st, slots3, ph3);
		report_timeline("K3 pack", d_st, slots3);
		(void)hipFree(d_st);
	} else hipLaunchKernelGGL(pipe_pack_kernel<false>, dim3(n * NT * 2), dim3(64), 0, s, pa, (uint64_t *)nullptr);
	e = hipGetLastError();
	if (timing) {
		(void)hipEventRecord(ev[4], s);
		(void)hipEventSynchronize(ev[4]);
		for (int i = 0; i < 4; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); tune->times_us[i] = ms * 1e3f; }
		for (auto &x : ev) (void)hipEventDestroy(x);
	}
	return e;
}

}  // namespace cct
