// CompaCT encode, stage (i) -- traversal, segmentation / mesh, delta coding, tag-byte pack (core.py:212-330 with
// curve.py:45-138 and cluster.py:20-199 underneath) -- as ONE streaming kernel: every pixel is read from HBM once and
// every payload byte written once.  Same applicability as encode_pipe.hip (block_size 16, traversal made of aligned 64x64
// tiles whose 4x4-pixel blocks are traversal blocks: every power-of-two square up to 1024x1024).
//
//   grid = slices x groups; a group = SW consecutive tiles (SW x 256 traversal blocks); one 256-lane workgroup per group,
//   one wave per tile.  A workgroup draws its group index from a per-slice ticket, so every group it ever waits for is
//   held by a workgroup that is already running (no assumption about dispatch order).
//
//   HBM -> VGPR   a lane owns a pair of horizontally adjacent 4x4 blocks: four 16-byte row segments, every load of a wave
//                 is eight full 128-byte lines.  The 64 blocks after the group (the look-ahead of the mesh search) come in
//                 the same way.
//   VGPR -> LDS   v_perm_b32 puts a block's 16 pixels in traversal order; the group's blocks sit in LDS in traversal
//                 order, 32 bytes each.  From here on lane L of wave w owns blocks 256 w + 64 s + L (s = 0..3): scans
//                 along the traversal are DPP wave scans.
//   analysis      packed 16-bit deltas; per block the 16 two-byte-token bits (core.py:316-323); difficult blocks
//                 (cluster.py:30-59) exactly, but only in waves that have a block with eight or more two-byte tokens
//   masks         one wave step per difficult block of its own tile: lane j tests candidate block i + j against block i,
//                 whose thresholds are scalar operands (cluster.py:122-158); blocks that fit nobody are dropped here
//   resolve       greedy first fit (cluster.py:79-190), one lane per island of difficult blocks.  Only an island that
//                 starts in the group's first 63 blocks can depend on the previous group: it waits for that group's
//                 carry word {which of my first 63 blocks are taken}; everything else resolves at once, and a group
//                 publishes its own carry as soon as its tail is decided (in general before its own wait ends)
//   sizes         token bytes per block along the final order (a block after a meshed block follows the last pixel of
//                 the previous GROUP of the partition); meshed pairs are handled by one lane per pair
//   offsets       DPP scans; the bytes before the group come from the predecessors' published totals
//   pack          the LDS region that held the pixels is zeroed and becomes the group's payload image; tokens are formed
//                 four pixels at a time (16-entry selector table) and OR-ed in at their final byte offsets; aligned
//                 16-byte stores flush it (byte stores only at the two ends shared with the neighbouring groups)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define LDS(T) __attribute__((address_space(3))) T

constexpr int HALO = 64;                // look-ahead blocks after the group
constexpr int LCAP = 64;                // list entries per wave kept in LDS (the rest spill to HBM)
constexpr int MAX_SW = 4;

// LDS layout (bytes) of a workgroup of SW waves = SW tiles.  The pixel stage and, later, the payload image share the first
// region: the image of a group is at most 32 bytes per block + one byte per meshed pair + the pairs that reach into the
// look-ahead + head + EOF + what the last token's ORs touch.
template <int SW>
struct Geo {
	static constexpr int ST = 64 * SW;             // lanes
	static constexpr int NBG = 256 * SW;           // blocks of a full group
	static constexpr int PAIR_FAST = ST;           // meshed pairs per group handled in registers; beyond: records in HBM
	static constexpr int PAIR_MAX = (NBG + HALO) / 2;
	static constexpr int L_PIX = 0;                // [-1 .. NBG + HALO) x 32
	static constexpr int IMG_NEED = 32 * NBG + PAIR_MAX + 32 * 63 + 16 + 1 + 16;
	static constexpr int PIX_NEED = 32 * (1 + NBG + HALO);
	static constexpr int L_IMG_BYTES = ((IMG_NEED > PIX_NEED ? IMG_NEED : PIX_NEED) + 15) & ~15;
	static constexpr int L_ROLE = L_IMG_BYTES;               // 16 + NBG + HALO
	static constexpr int L_LMASK = L_ROLE + 16 + NBG + HALO; // SW x LCAP u64; later: u16 payload offset of every block
	static constexpr int L_LIDX = L_LMASK + SW * LCAP * 8;   // SW x LCAP u16
	static constexpr int L_PAIRS = L_LIDX + SW * LCAP * 2;   // PAIR_MAX u16
	static constexpr int L_OTAB = L_PAIRS + PAIR_MAX * 2;    // 64 dwords
	static constexpr int L_TTAB = L_OTAB + 256;              // 64 dwords
	static constexpr int L_MISC = L_TTAB + 256;              // 32 dwords, then the difficult-block ballots of the SW x 4 slots
	static constexpr int L_TOTAL = L_MISC + 128 + MAX_SW * 4 * 8;
	static_assert(NBG * 2 <= SW * LCAP * 8, "block offsets alias the mask list");
	static_assert(L_LMASK % 16 == 0 && L_OTAB % 16 == 0 && L_MISC % 16 == 0 && L_PAIRS % 2 == 0, "alignment");
};
static_assert(Geo<4>::L_TOTAL <= 40960, "four workgroups of four waves per CU");

enum : int {  // dwords of the misc area
	M_TICKET = 0, M_CNT = 1 /* MAX_SW */, M_NPAIRS = 5, M_WTOT = 6 /* MAX_SW */, M_HB = 10, M_WIDE = 11, M_CARRY_LO = 12, M_CARRY_HI = 13,
	M_LASTPX = 14, M_STATUS = 15, M_NDIFF = 16, M_BASE = 17, M_ACC_JUMP = 18, M_ACC_DIFF = 19
};
constexpr uint32_t WIDE_14 = 1u, WIDE_11 = 2u;   // a pixel >= 0x4000 / >= 0x0800 somewhere in the group

#define TILE_ORG(a, t) ((a).tiles.orgo[t] & 0xFFFFFFu)
#define TILE_ORIENT(a, t) ((int)((a).tiles.orgo[t] >> 24))

// ---- packed 16-bit arithmetic --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
template <class T>
__device__ __forceinline__ T lds_add(LDS(T) *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or(LDS(uint32_t) *p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_max(LDS(uint32_t) *p, uint32_t v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
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
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)  // a value every lane holds, as a scalar
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}

// Rows 0..3 of a 4x4 block as (columns 0-1, columns 2-3) dwords -> its 16 pixels in traversal order (see
// build_pipe_tables in api.cpp for the structure this relies on and verifies).
__device__ __forceinline__ void permute_block(uint32_t l0, uint32_t h0, uint32_t l1, uint32_t h1, uint32_t l2, uint32_t h2,
                                              uint32_t l3, uint32_t h3, const LDS(uint32_t) *ot, uint32_t d[8])
{
	const u32x4 s0 = *(const LDS(u32x4) *)ot, s1 = *(const LDS(u32x4) *)(ot + 4);
	const uint32_t cb = ot[8];
	const bool c0 = (cb & 1u) != 0, c1 = (cb & 2u) != 0;
	const uint32_t t0 = c0 ? h2 : l0, b0 = c0 ? h3 : l1;
	const uint32_t t2 = c0 ? l0 : h2, b2 = c0 ? l1 : h3;
	const uint32_t t1 = c1 ? h0 : l2, b1 = c1 ? h1 : l3;
	const uint32_t t3 = c1 ? l2 : h0, b3 = c1 ? l3 : h1;
	d[0] = perm(b0, t0, s0.x); d[1] = perm(b0, t0, s0.y);
	d[2] = perm(b1, t1, s0.z); d[3] = perm(b1, t1, s0.w);
	d[4] = perm(b2, t2, s1.x); d[5] = perm(b2, t2, s1.y);
	d[6] = perm(b3, t3, s1.z); d[7] = perm(b3, t3, s1.w);
}

// packed deltas x[j] = (D[2j] - D[2j-1], D[2j+1] - D[2j]) mod 2^16 of 2 NW traversal-ordered pixels after pixel pv
template <int NW>
__device__ __forceinline__ void deltas(const uint32_t d[NW], uint32_t pv, uint32_t x[NW])
{
	x[0] = pk_sub(d[0], (d[0] << 16) | (pv & 0xFFFFu));
#pragma unroll
	for (int j = 1; j < NW; j++) x[j] = pk_sub(d[j], __builtin_amdgcn_alignbit(d[j], d[j - 1], 16));
}
__device__ __forceinline__ int px16(const uint32_t *d, int i) { return (int)((d[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu); }
__device__ __forceinline__ bool tok_two(int dlt) { return (uint32_t)(dlt + 63) > 127u; }      // core.py:316
__device__ __forceinline__ bool seg_large(int dlt) { return (uint32_t)(dlt + 64) > 128u; }    // cluster.py:37-38
__device__ __forceinline__ bool out_of_q7(int dlt) { return (uint32_t)(dlt + 2047) > 4095u; }

// bit i <=> pixel i of the 16 takes two bytes (delta outside [-63, 64]); packed arithmetic: every pixel < 0x4000
__device__ __forceinline__ uint32_t two_byte_bits(const uint32_t x[8])
{
	const uint32_t K64 = 0x00400040u, K63 = 0x003F003Fu;
	uint32_t acc[2] = {0, 0};
#pragma unroll
	for (int g = 0; g < 4; g++) {
		const uint32_t wa = pk_sub(K64, x[2 * g]) | pk_add(x[2 * g], K63);          // sign: delta > 64 | delta < -63
		const uint32_t wb = pk_sub(K64, x[2 * g + 1]) | pk_add(x[2 * g + 1], K63);
		const uint32_t m = perm(wb, wa, 0x07050301u) & 0x80808080u;
		acc[g >> 1] = __builtin_amdgcn_udot4(m, (g & 1) ? 0x80402010u : 0x08040201u, acc[g >> 1], false);  // 128 x nibble (<< 4)
	}
	return (acc[0] >> 7) | ((acc[1] >> 7) << 8);
}
// a delta outside [-2047, 2048] among the 2 NW packed ones (pixels < 0x4000)
template <int NW>
__device__ __forceinline__ bool any_out_of_q7(const uint32_t x[NW])
{
	const uint32_t K2048 = 0x08000800u, K2047 = 0x07FF07FFu;
	uint32_t bad = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) bad |= pk_sub(K2048, x[j]) | pk_add(x[j], K2047);
	return (bad & 0x80008000u) != 0;
}
// number of halves equal to -64 among 16 packed deltas, and whether delta 0 is one
__device__ __forceinline__ uint32_t count_m64(const uint32_t x[8], uint32_t &first_is)
{
	uint32_t acc = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) acc = pk_add(acc, pk_min_u(x[j] ^ 0xFFC0FFC0u, 0x00010001u));   // 1 per half != -64
	first_is = (x[0] & 0xFFFFu) == 0xFFC0u ? 1u : 0u;
	return 16u - ((acc & 0xFFFFu) + (acc >> 16));
}

// tokens of 16 pixels (packed deltas x, two-byte bits mb) OR-ed into the zeroed payload image at byte offset o;
// ttab entry f: v_perm selectors of the up to 8 bytes, their number, the bits of the low bytes that are kept
__device__ __forceinline__ uint32_t emit16(const uint32_t x[8], uint32_t mb, uint32_t o, LDS(uint8_t) *img, const LDS(uint8_t) *ttab)
{
#pragma unroll
	for (int g = 0; g < 4; g++) {
		const uint32_t xa = x[2 * g], xb = x[2 * g + 1];
		const u32x4 te = *(const LDS(u32x4) *)(ttab + ((mb >> (4 * g)) & 15u) * 16u);
		const uint32_t P = perm(xb, xa, 0x06040200u) & te.w;                          // short: 7 bits; full: second byte
		const uint32_t X = (perm(xb, xa, 0x07050301u) & 0x0F0F0F0Fu) | 0xE0E0E0E0u;   // full: first byte
		const uint32_t lo = perm(X, P, te.x), hi = perm(X, P, te.y);
		const uint32_t s8 = (o & 3u) * 8u;
		const uint64_t v01 = ((uint64_t)hi << 32 | lo) << s8;
		const uint32_t d2 = (uint32_t)(((uint64_t)hi << s8) >> 32);
		LDS(uint32_t) *w = (LDS(uint32_t) *)(img + (o & ~3u));
		lds_or(w, (uint32_t)v01);
		lds_or(w + 1, (uint32_t)(v01 >> 32));
		lds_or(w + 2, d2);
		o += te.z;
	}
	return o;
}

// the same for 16 pixels that all take one byte (the common case: smooth tissue, air): four dwords of 7-bit deltas, moved to
// the byte phase of o by one v_perm each
__device__ __forceinline__ void emit16_short(const uint32_t x[8], uint32_t o, LDS(uint8_t) *img)
{
	uint32_t P[4];
#pragma unroll
	for (int g = 0; g < 4; g++) P[g] = perm(x[2 * g + 1], x[2 * g], 0x06040200u) & 0x7F7F7F7Fu;
	const uint32_t sel = 0x07060504u - (o & 3u) * 0x01010101u;   // byte i of a word: byte 4 + i - phase of {this dword : the one before}
	LDS(uint32_t) *w = (LDS(uint32_t) *)(img + (o & ~3u));
	lds_or(w, perm(P[0], 0u, sel));
	lds_or(w + 1, perm(P[1], P[0], sel));
	lds_or(w + 2, perm(P[2], P[1], sel));
	lds_or(w + 3, perm(P[3], P[2], sel));
	lds_or(w + 4, perm(0u, P[3], sel));
}

struct Hand {  // the three words a group publishes (8-byte agent-scope stores; bit 63 = valid; zeroed before every launch)
	uint64_t carry;    // bits 0..62: which of the NEXT group's first 63 blocks this group's leaders have taken
	uint64_t lastpx;   // bits 0..15: last pixel of this group's last emitted block group (the next group's predecessor pixel)
	uint64_t total;    // bits 0..23 payload bytes, 24..39 meshed pairs, 40..55 difficult blocks
	uint64_t pad;
};
constexpr uint64_t H_VALID = 1ull << 63;
__device__ __forceinline__ void hand_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t hand_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr uint32_t SPIN_LIMIT = 1u << 18;   // polls (a microsecond or more each) before a wait gives up with CCT_ST_INTERNAL

// diagnostic build only (CCT_STREAM_STAMPS=1): shader clock at every phase boundary, per wave, to a buffer of their own
#define STAMP(k) do { if (STAMPS) { __builtin_amdgcn_sched_barrier(0); st_[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
constexpr int N_STAMPS = 16;

template <int SW, bool SGN, bool STAMPS>
__global__ void __launch_bounds__(64 * SW, 4) stream_kernel(StreamArgs a, uint64_t *stamps)
{
	using G = Geo<SW>;
	constexpr int ST = G::ST, NBG = G::NBG, PAIR_FAST = G::PAIR_FAST;
	constexpr int L_PIX = G::L_PIX, L_IMG_BYTES = G::L_IMG_BYTES, L_ROLE = G::L_ROLE, L_LMASK = G::L_LMASK, L_LIDX = G::L_LIDX;
	constexpr int L_PAIRS = G::L_PAIRS, L_OTAB = G::L_OTAB, L_TTAB = G::L_TTAB, L_MISC = G::L_MISC, L_TOTAL = G::L_TOTAL;
	uint64_t st_[N_STAMPS] = {0};
	uint64_t rt0_ = 0;
	if (STAMPS) { rt0_ = __builtin_amdgcn_s_memrealtime(); }
	STAMP(0);
	__shared__ __attribute__((aligned(16))) uint8_t smem[L_TOTAL];
	LDS(uint8_t) *lds = (LDS(uint8_t) *)smem;
	LDS(uint8_t) *pix0 = lds + L_PIX + 32;                      // block b at pix0 + 32 b (b = -1: the pixel before the group)
	LDS(uint8_t) *roles = lds + L_ROLE + 16;                    // 0 alone, 1..63 leader (jump), >= 0x80 partner
	LDS(uint64_t) *lmask = (LDS(uint64_t) *)(lds + L_LMASK);
	LDS(uint16_t) *boff = (LDS(uint16_t) *)(lds + L_LMASK);     // after the resolve
	LDS(uint16_t) *lidx = (LDS(uint16_t) *)(lds + L_LIDX);
	LDS(uint64_t) *dbal = (LDS(uint64_t) *)(lds + L_MISC + 128);
	LDS(uint16_t) *pairs = (LDS(uint16_t) *)(lds + L_PAIRS);
	LDS(uint32_t) *otab = (LDS(uint32_t) *)(lds + L_OTAB);
	LDS(uint8_t) *ttab = lds + L_TTAB;
	LDS(uint32_t) *misc = (LDS(uint32_t) *)(lds + L_MISC);

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int NT = a.n_tiles, NB = a.e.NB, N = a.e.N, gps = a.gps, pitch = a.row_pitch;
	// consecutive workgroups belong to different slices: the groups of one slice start a whole round of slices apart, so a
	// group's predecessors are well ahead of it when it asks for their carry and totals
	const int sl = blockIdx.x % (gridDim.x / gps);
	const bool seg = (a.e.flags & CCT_FLAG_SEGMENTATION) != 0;
	const uint16_t *img = a.e.images + (size_t)sl * N;

	// ---- ticket, tables, cleared state.  The pixel loads do not wait for the ticket: they are issued for the group the
	// dispatch order suggests and repeated in the (never observed) case that the ticket says otherwise.
	uint32_t tick = 0;
	if (tid == 0) tick = atomicAdd(a.ticket + sl, 1u);
	struct Loaded { u32x4 r[2][4]; uint32_t e[2]; u32x4 hr[4]; uint2 he; uint32_t before; };
	auto issue_loads = [&](int gg, Loaded &L) {
		const int t0_ = gg * SW, ntg_ = min(SW, NT - t0_);
		if (wave == SW - 1) {
			// the look-ahead: the first 64 traversal blocks of the next tile = one 32x32-pixel quadrant (8 rows of 4 block pairs,
			// origin in the kernel arguments), and the pixel before the group
			L.he = make_uint2(0u, 0u);
			if (t0_ + ntg_ < NT) {
				const int to = TILE_ORIENT(a, t0_ + ntg_);
				L.he = reinterpret_cast<const uint2 *>(a.htab)[to * 32 + (lane & 31)];
				const uint16_t *p = img + TILE_ORG(a, t0_ + ntg_) + a.tiles.qorg[to] + (size_t)(((lane & 31) >> 2) * 4) * pitch + (lane & 3) * 8;
#pragma unroll
				for (int q = 0; q < 4; q++) L.hr[q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);
			}
			L.before = 0;
			if (t0_ > 0) L.before = img[TILE_ORG(a, t0_ - 1) + a.tiles.last[TILE_ORIENT(a, t0_ - 1)]];
		}
		if (wave < ntg_) {
			const int t = t0_ + wave;
			const uint32_t org = TILE_ORG(a, t);
			const int to = TILE_ORIENT(a, t);
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int rl = 64 * h + lane;
				const uint16_t *p = img + org + (size_t)((rl >> 3) * 4) * pitch + (rl & 7) * 8;
#pragma unroll
				for (int q = 0; q < 4; q++) L.r[h][q] = *reinterpret_cast<const u32x4 *>(p + (size_t)q * pitch);
				L.e[h] = a.ptab[(size_t)(to * 128 + rl) * 4];
			}
		}
	};
	const int g_guess = (int)((blockIdx.x / (gridDim.x / gps) + ((a.dbg & 4) ? 1u : 0u)) % (uint32_t)gps);   // (dbg 4: a wrong guess on purpose)
	// the tables are requested before the pixels (the loads of a wave return in order: what is needed first goes first)
	for (int i = tid; i < 128; i += ST) {   // (a workgroup of one wave makes two rounds)
		const uint32_t tabv = i < 64 ? a.otab[i] : a.ttab[i - 64];
		if (i < 64) otab[i] = tabv;
		else ((LDS(uint32_t) *)ttab)[i - 64] = tabv;
	}
	for (int i = tid; i < (16 + NBG + HALO) / 4; i += ST) ((LDS(uint32_t) *)(lds + L_ROLE))[i] = 0u;
	if (tid < 32) misc[tid] = 0u;
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the pixel loads stay in flight
	STAMP(1);
	// Everything up to the first word handed to another group runs on the guess; the ticket is looked at there (it has
	// arrived long before) and, should it differ, the group it names is processed instead.
	Hand *hand = reinterpret_cast<Hand *>(a.hand) + (size_t)sl * gps;
	int g = g_guess, t0, ntg, nbg, abs0;
	bool has_next, active, wide, big, carry_lane;
	uint64_t carry_v;
	const uint64_t *carry_src;
	uint32_t x[4][8], info[4];
	uint64_t bal[4];
	auto front = [&]() {  // pixels of group g -> LDS, analysis of the lane's blocks, ballots and ticket shared
	Loaded L;
	issue_loads(g, L);
	t0 = g * SW;
	ntg = min(SW, NT - t0);                 // tiles of this group
	nbg = ntg * 256;                        // its blocks
	abs0 = t0 * 256;                        // slice index of its first block
	has_next = t0 + ntg < NT;               // a look-ahead exists
	active = wave < ntg;
	// the predecessor's carry and last pixel: asked for now, looked at after the resolve (most groups have published both
	// long before); lanes 0 and 1 of wave 0
	carry_v = 0;
	carry_lane = wave == 0 && g > 0 && lane < 2 && !(a.dbg & 1);
	carry_src = lane == 0 ? &hand[g > 0 ? g - 1 : 0].carry : &hand[g > 0 ? g - 1 : 0].lastpx;
	if (carry_lane) carry_v = hand_load(carry_src);

	// ---- VGPR -> LDS (traversal order)
	{
		uint32_t orall = 0;
		if (active) {
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const u32x4 *r = L.r[h];
				const uint32_t e = L.e[h];
				uint32_t dA[8], dB[8];
				permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, otab + ((e >> 8) & 3u) * 16, dA);
				permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, otab + ((e >> 24) & 3u) * 16, dB);
				LDS(u32x4) *pa = (LDS(u32x4) *)(pix0 + 32 * (256 * wave + (int)(e & 0xFFu)));
				LDS(u32x4) *pb = (LDS(u32x4) *)(pix0 + 32 * (256 * wave + (int)((e >> 16) & 0xFFu)));
				pa[0] = (u32x4){dA[0], dA[1], dA[2], dA[3]}; pa[1] = (u32x4){dA[4], dA[5], dA[6], dA[7]};
				pb[0] = (u32x4){dB[0], dB[1], dB[2], dB[3]}; pb[1] = (u32x4){dB[4], dB[5], dB[6], dB[7]};
#pragma unroll
				for (int j = 0; j < 8; j++) orall |= dA[j] | dB[j];
			}
		}
		if (wave == SW - 1) {
			if (has_next) {
				const u32x4 *r = L.hr;
				uint32_t dA[8], dB[8];
				permute_block(r[0].x, r[0].y, r[1].x, r[1].y, r[2].x, r[2].y, r[3].x, r[3].y, otab + ((L.he.x >> 8) & 3u) * 16, dA);
				permute_block(r[0].z, r[0].w, r[1].z, r[1].w, r[2].z, r[2].w, r[3].z, r[3].w, otab + ((L.he.x >> 24) & 3u) * 16, dB);
				if (lane < 32) {
					LDS(u32x4) *pa = (LDS(u32x4) *)(pix0 + 32 * (nbg + (int)(L.he.x & 0xFFu)));
					LDS(u32x4) *pb = (LDS(u32x4) *)(pix0 + 32 * (nbg + (int)((L.he.x >> 16) & 0xFFu)));
					pa[0] = (u32x4){dA[0], dA[1], dA[2], dA[3]}; pa[1] = (u32x4){dA[4], dA[5], dA[6], dA[7]};
					pb[0] = (u32x4){dB[0], dB[1], dB[2], dB[3]}; pb[1] = (u32x4){dB[4], dB[5], dB[6], dB[7]};
				}
#pragma unroll
				for (int j = 0; j < 8; j++) orall |= dA[j] | dB[j];
			}
			if (lane == 0) *(LDS(uint32_t) *)(pix0 - 4) = L.before << 16;
			orall |= L.before;
		}
		const uint32_t wd = (__any((orall & 0xC000C000u) != 0) ? WIDE_14 : 0u) | (__any((orall & 0xF800F800u) != 0) ? WIDE_11 : 0u);
		if (wd && lane == 0) lds_or(&misc[M_WIDE], wd);
	}
	STAMP(2);
	__syncthreads();
	STAMP(3);
	const uint32_t gw = misc[M_WIDE];
	wide = SGN || (gw & WIDE_14) != 0;     // exact (unpacked) arithmetic for the whole group
	big = (gw & WIDE_11) != 0;             // a delta outside [-2047, 2048] needs a pixel >= 2048

	// ---- analysis of the lane's four blocks
	// info[s]: two-byte bits (16) | tokens with two bytes << 16 (5) | transitions incl. the entering one << 21 (5) |
	//          difficult << 26 | a delta out of the format's range among pixels 1..15 << 27 | the same for pixel 0 << 28
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const int b = 256 * wave + 64 * s + lane;
		bal[s] = 0; info[s] = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) x[s][j] = 0;
		if (!active) continue;
		const LDS(u32x4) *pp = (const LDS(u32x4) *)(pix0 + 32 * b);
		const u32x4 v0 = pp[0], v1 = pp[1];
		const uint32_t d[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
		const bool first = abs0 + b == 0;                                       // the slice starts from pixel value 0 (core.py:278)
		const uint32_t pv = first ? 0u : (uint32_t) * (const LDS(uint16_t) *)(pix0 + 32 * b - 2);
		deltas<8>(d, pv, x[s]);
		uint32_t mb, chg = 0, enter = 0;
		bool bad = false, bad0 = false;
		if (!wide) {
			mb = two_byte_bits(x[s]);
			if (__any(__popc(mb) >= 8)) {                                         // a difficult block has >= 8 two-byte tokens
				uint32_t f0;
				const uint32_t n64 = count_m64(x[s], f0);                           // delta -64: two bytes, but not "large" 
				chg = (uint32_t)__popc(mb >> 1) - (n64 - f0);
				enter = (mb & 1u) & (f0 ^ 1u);
			}
			if (big) {
				uint32_t rest[8];
#pragma unroll
				for (int j = 0; j < 8; j++) rest[j] = x[s][j];
				rest[0] &= 0xFFFF0000u;
				bad = any_out_of_q7<8>(rest);
				bad0 = out_of_q7((int)(int16_t)(x[s][0] & 0xFFFFu));
			}
		} else {
			mb = 0;
			int pu = (int)pv;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const int v = px16(d, i), dlt = v - pu;
				if (tok_two(dlt)) { mb |= 1u << i; if (i == 0) bad0 = out_of_q7(dlt); else bad |= out_of_q7(dlt); }
				const int ds = SGN ? ((int)(int16_t)v - (int)(int16_t)pu) : dlt;
				const uint32_t lg = seg_large(ds) ? 1u : 0u;
				if (i == 0) enter = lg; else chg += lg;
				pu = v;
			}
		}
		if (first) enter = 0;                                                   // P[0] = 0 (cluster.py:33)
		const bool difficult = seg && chg >= 8u;                                // cluster.py:58
		info[s] = mb | ((uint32_t)__popc(mb) << 16) | ((chg + enter) << 21) | (difficult ? 1u << 26 : 0u) | (bad ? 1u << 27 : 0u) |
		          (bad0 ? 1u << 28 : 0u);
		bal[s] = __ballot(difficult);
	}

	// the ballots of the difficult blocks are shared (the mask phase splits them among the waves), and with them the ticket
	if (lane == 0) {
#pragma unroll
		for (int s = 0; s < 4; s++) dbal[wave * 4 + s] = bal[s];
	}
	if (tid == 0) misc[M_TICKET] = tick;
	__syncthreads();
	};
	front();
	if ((int)misc[M_TICKET] != g) {
		// (never observed) the dispatch order and the ticket disagree: start over as the group the ticket names
		g = (int)misc[M_TICKET];
		__syncthreads();
		if (tid < 32) misc[tid] = 0u;
		__syncthreads();
		front();
	}
	if (carry_lane && !(carry_v & H_VALID)) carry_v = hand_load(carry_src);   // asked again: the answer arrives during the mask phase
	// No difficult block among the group's last 64 blocks: none of them can lead a pair, so nothing of the next group is
	// taken and the group ends with its last block emitted alone -- the next group can be told right away
	const bool last_group = g == gps - 1;
	bool carry_out_done = false;
	if (wave == SW - 1 && !last_group && bal[3] == 0) {
		if (lane == 0) {
			hand_store(&hand[g].carry, H_VALID);
			hand_store(&hand[g].lastpx, (uint64_t) * (const LDS(uint16_t) *)(pix0 + 32 * (nbg - 1) + 30) | H_VALID);
		}
		carry_out_done = true;
	}
	STAMP(4);
	// ---- candidate fit masks of the group's difficult blocks (cluster.py:110-158).  The waves share the work by ordinal:
	// wave v of the ntg active ones takes the v-th quarter of the group's difficult blocks, in order (a tile full of bone
	// would otherwise keep the other three waiting), so the per-wave lists concatenate to the ordered list the resolve walks.
	// Entries with an empty mask are dropped: such a block can neither take a partner nor change the state of the walk.
	uint32_t cnt = 0;
	if (active && seg) {
		LDS(uint16_t) *my_idx = lidx + wave * LCAP;
		LDS(uint64_t) *my_mask = lmask + wave * LCAP;
		uint16_t *sp_idx = a.spill_idx + (size_t)sl * NB + abs0 + 256 * wave;
		uint64_t *sp_mask = a.spill_mask + (size_t)sl * NB + abs0 + 256 * wave;
		// lane q < 4 ntg holds the ballot of slot q; ordinals by a wave scan
		const uint64_t myb = lane < 4 * ntg ? dbal[lane] : 0ull;
		const uint32_t myc = (uint32_t)__popcll(myb);
		const uint32_t inc = wave_incl_scan(myc);
		const uint32_t D = rdlane(inc, 63);
		const uint32_t lo = (uint32_t)wave * D / (uint32_t)ntg, hi = (uint32_t)(wave + 1) * D / (uint32_t)ntg;
		const uint64_t nonempty = __ballot(myc != 0);
		const uint32_t mlo = (uint32_t)myb, mhi = (uint32_t)(myb >> 32);
		auto slot_mask = [&](int q) -> uint64_t { return (uint64_t)rdlane(mlo, q) | (uint64_t)rdlane(mhi, q) << 32; };
		int sq = 0;
		uint64_t bm = 0;
		if (hi > lo) {   // seek ordinal lo: the first slot whose inclusive count exceeds it
			sq = __builtin_ctzll(__ballot(inc > lo));
			bm = slot_mask(sq);
			for (uint32_t skip = lo - (rdlane(inc, sq) - rdlane(myc, sq)); skip; skip--) bm &= bm - 1;
		}
		auto next_block = [&]() -> int {  // the next difficult block of the wave's range, group index (wave-uniform)
			if (bm == 0) {
				sq = __builtin_ctzll(nonempty & ~((2ull << sq) - 1ull));
				bm = slot_mask(sq);
			}
			const int la = __builtin_ctzll(bm);
			bm &= bm - 1;
			return 64 * sq + la;
		};
		struct Cand { u32x4 a0, a1, w0, w1; uint32_t pt, pp; };
		auto request = [&](int ba, Cand &c) {  // every lane: block A (one address: a broadcast read), its candidate B = A + lane,
			const LDS(uint8_t) *pA = pix0 + 32 * ba;  // and pixel t = lane & 15 of A with its predecessor
			c.a0 = ((const LDS(u32x4) *)pA)[0]; c.a1 = ((const LDS(u32x4) *)pA)[1];
			const LDS(u32x4) *pq = (const LDS(u32x4) *)(pA + 32 * lane);
			c.w0 = pq[0]; c.w1 = pq[1];
			c.pt = *(const LDS(uint16_t) *)(pA + 2 * (lane & 15)); c.pp = *(const LDS(uint16_t) *)(pA + 2 * (lane & 15) - 2);
		};
		Cand cn;
		int ba_next = 0;
		if (hi > lo) { ba_next = next_block(); request(ba_next, cn); }
		for (uint32_t todo = hi - lo; todo; todo--) {
			const int ba = ba_next;
			const Cand c = cn;
			if (todo > 1) { ba_next = next_block(); request(ba_next, cn); }   // in flight while this block is evaluated
			const bool a_first = abs0 + ba == 0;
			const u32x4 a0 = c.a0, a1 = c.a1, w0 = c.w0, w1 = c.w1;
			// transitions of A including the entering one (cluster.py:110): lane t looks at pixel t
			int pt = (int)c.pt, pp_ = (int)c.pp;
			if (SGN) { pt = (int)(int16_t)pt; pp_ = (int)(int16_t)pp_; }
			const uint32_t cur = (uint32_t)__popcll(__ballot(lane < 16 && !(a_first && lane == 0) && seg_large(pt - pp_)));
			const uint32_t av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
			const uint32_t bw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
			const bool valid = lane >= 1 && abs0 + ba + lane < NB;
			uint32_t up;
			if (!wide) {
				//   B[t] - A[t] >= 65      <=>  bit 15 of B[t] - (A[t] + 65 + 0x8000)
				//   A[t+1] - B[t] >= 65    <=>  bit 15 of (A[t+1] - 65 + 0x8000) - B[t]       (t = 15 has no successor: never)
				uint32_t acc = 0;
#pragma unroll
				for (int j = 0; j < 8; j++) {
					const uint32_t an = j < 7 ? __builtin_amdgcn_alignbit(av[j + 1], av[j], 16) : (av[7] >> 16);
					const uint32_t r1 = pk_sub(bw[j], pk_add(av[j], 0x80418041u));
					const uint32_t r2 = pk_sub(pk_add(an, 0x7FBF7FBFu), bw[j]);
					acc = __builtin_amdgcn_sad_u8(perm(r2, r1, 0x07050301u) & 0x80808080u, 0u, acc);
				}
				up = acc >> 7;
			} else {
				up = 0;
				int bprev = 0;
#pragma unroll
				for (int t = 0; t < 16; t++) {
					int avv = (int)((av[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu), bv = (int)((bw[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu);
					if (SGN) { avv = (int)(int16_t)avv; bv = (int)(int16_t)bv; }
					if (t > 0) up += (avv - bprev >= 65) ? 1u : 0u;
					up += (bv - avv >= 65) ? 1u : 0u;
					bprev = bv;
				}
			}
			// cluster.py:153,158: up + 1 < current_delta - 2 in uint32; block 0 of the slice wraps: it always fits
			const bool fit = valid && (a_first ? true : ((up + 1u) < (cur - 2u)));
			const uint64_t mk = __ballot(fit);
			if (mk) {
				if (lane == 0) {
					if (cnt < (uint32_t)LCAP) { my_idx[cnt] = (uint16_t)ba; my_mask[cnt] = mk; }
					else { sp_idx[cnt] = (uint16_t)ba; sp_mask[cnt] = mk; }
				}
				cnt++;
			}
		}
	}
	if (lane == 0) misc[M_CNT + wave] = cnt;
	if (carry_lane && !(carry_v & H_VALID)) carry_v = hand_load(carry_src);
	STAMP(5);
	__syncthreads();
	STAMP(6);

	// ---- resolve: greedy first fit per island (two listed blocks more than 63 apart cannot influence each other)
	uint32_t cwr[MAX_SW];                         // entries listed per wave
#pragma unroll
	for (int v = 0; v < MAX_SW; v++) cwr[v] = v < SW ? misc[M_CNT + v] : 0u;
	auto cw_ = [&](uint32_t w) -> uint32_t { return w == 0 ? cwr[0] : (w == 1 ? cwr[1] : (w == 2 ? cwr[2] : cwr[3])); };
	uint32_t E = 0;
#pragma unroll
	for (int v = 0; v < SW; v++) E += cwr[v];
	auto idx_of = [&](uint32_t w, uint32_t r) -> uint32_t {
		uint32_t k;
		if (r < (uint32_t)LCAP) k = lidx[w * LCAP + r]; else k = a.spill_idx[(size_t)sl * NB + abs0 + 256 * w + r];
		return k;
	};
	auto mask_of = [&](uint32_t w, uint32_t r) -> uint64_t {
		uint64_t m;
		if (r < (uint32_t)LCAP) m = lmask[w * LCAP + r]; else m = a.spill_mask[(size_t)sl * NB + abs0 + 256 * w + r];
		return m;
	};
	auto locate = [&](uint32_t e, uint32_t &w, uint32_t &r) {  // entry e of the concatenated lists
		w = 0; r = e;
#pragma unroll
		for (int v = 0; v < SW - 1; v++) if (w == (uint32_t)v && r >= cwr[v]) { r -= cwr[v]; w++; }
	};
	auto next_of = [&](uint32_t &w, uint32_t &r) -> bool {       // the entry after (w, r)
		r++;
		while (w < (uint32_t)SW && r >= cw_(w)) { w++; r = 0; }
		return w < (uint32_t)SW;
	};
	auto walk = [&](uint32_t w, uint32_t r, uint64_t cw) {
		uint32_t i = idx_of(w, r);
		uint64_t mk = mask_of(w, r);
		for (;;) {
			uint32_t wn = w, rn = r;
			const bool more = next_of(wn, rn);
			const uint32_t inext = more ? idx_of(wn, rn) : 0u;
			const uint64_t mnext = more ? mask_of(wn, rn) : 0ull;
			if (!(cw & 1ull)) {
				const uint64_t avail = mk & ~cw & ~1ull;
				if (avail) {
					const int j = __ffsll((long long)avail) - 1;
					roles[i] = (uint8_t)j;
					roles[i + j] = ROLE_PARTNER;
					cw |= 1ull << j;
				}
			}
			if (!more || inext - i > 63u) break;
			cw >>= (inext - i);
			i = inext; mk = mnext; w = wn; r = rn;
		}
	};
	// the island that starts in the group's first 63 blocks waits for the previous group's carry
	bool head_deferred = false;
	if (E) {
		uint32_t w0_ = 0, r0_ = 0;
		while (cw_(w0_) == 0) w0_++;
		head_deferred = g > 0 && idx_of(w0_, r0_) <= 62u;
	}
	for (uint32_t e0 = tid; e0 < E; e0 += ST) {
		uint32_t w, r;
		locate(e0, w, r);
		const uint32_t i0 = idx_of(w, r);
		bool head = e0 == 0;
		if (e0 > 0) {
			uint32_t wp, rp;
			locate(e0 - 1, wp, rp);
			head = i0 - idx_of(wp, rp) > 63u;
			if (head) lds_max(&misc[M_HB], ~e0);   // the first head after entry 0 = the end of the first island
		}
		if (head && !(e0 == 0 && head_deferred)) walk(w, r, 0ull);
	}
	__syncthreads();
	STAMP(7);
	// does the deferred island reach the blocks that decide what this group hands on?
	bool chain = false;
	if (head_deferred) {
		const uint32_t hbv = misc[M_HB];
		const uint32_t hb = hbv ? ~hbv : E;
		uint32_t w, r;
		locate(hb - 1, w, r);
		chain = idx_of(w, r) + 63u >= (uint32_t)(nbg - 64);
	}
	auto publish_carry = [&]() {  // one whole wave
		const uint64_t cm = __ballot(lane < 63 && roles[nbg + lane] != 0);
		uint32_t px = 0;
		if (lane == 0) {
			int q = nbg - 1;
			while (q > 0 && roles[q] >= 0x80u) q--;
			q += roles[q];                                // a pair ends with its partner's last pixel
			px = *(const LDS(uint16_t) *)(pix0 + 32 * q + 30);
			hand_store(&hand[g].carry, cm | H_VALID);
			hand_store(&hand[g].lastpx, (uint64_t)px | H_VALID);
		}
	};
	if (wave == SW - 1 && !last_group && !carry_out_done && !(head_deferred && chain)) publish_carry();
	if (wave == 0 && g > 0 && !(a.dbg & 1)) {
		uint64_t v = carry_v;
		uint32_t spins = 0;
		bool ok = lane >= 2 || (v & H_VALID) != 0;
		const uint64_t *src = carry_src;
		while (!__all(ok)) {
			if (!ok) { v = hand_load(src); ok = (v & H_VALID) != 0; }
			if (++spins > SPIN_LIMIT) { if (lane == 0) lds_or(&misc[M_STATUS], CCT_ST_INTERNAL); break; }
			if (!__all(ok)) __builtin_amdgcn_s_sleep(4);
		}
		const uint32_t c_lo = rdlane((uint32_t)v, 0), c_hi = rdlane((uint32_t)(v >> 32), 0) & 0x7FFFFFFFu;
		const uint32_t lpx = rdlane((uint32_t)v, 1) & 0xFFFFu;
		const uint64_t carry = (uint64_t)c_hi << 32 | c_lo;
		if (lane < 63 && ((carry >> lane) & 1ull)) roles[lane] = ROLE_PARTNER;
		if (lane == 0) { misc[M_LASTPX] = lpx; misc[M_CARRY_LO] = c_lo; misc[M_CARRY_HI] = c_hi; }
		if (head_deferred && lane == 0) {
			uint32_t w = 0;
			while (cw_(w) == 0) w++;
			walk(w, 0u, carry >> (idx_of(w, 0u) & 63u));
		}
	}
	STAMP(8);
	__syncthreads();
	STAMP(9);
	if (wave == SW - 1 && !last_group && !carry_out_done && head_deferred && chain) publish_carry();
	const uint32_t lastpx_in = misc[M_LASTPX];            // 0 for the first group: the slice starts from pixel value 0
	// the meshed pairs, listed by their leaders' lanes
	if (active) {
#pragma unroll
		for (int s = 0; s < 4; s++) {
			const int b = 256 * wave + 64 * s + lane;
			const uint32_t r = roles[b];
			if (r >= 1u && r < 0x80u) pairs[lds_add(&misc[M_NPAIRS], 1u)] = (uint16_t)b;
		}
	}
	__syncthreads();

	// last pixel written before block b's group in the final order (b alone or a leader)
	auto prev_px_final = [&](int b) -> uint32_t {
		if (b == 0) return lastpx_in;
		int q = b - 1;
		uint32_t rq = roles[q];
		if (rq != 0) {
			while (q >= 0 && roles[q] >= 0x80u) q--;
			if (q < 0) return lastpx_in;
			q += roles[q];
		}
		return *(const LDS(uint16_t) *)(pix0 + 32 * q + 30);
	};

	// ---- meshed pairs, one lane each: tokens along A0 B0 A1 B1 ... (cluster.py:173-174, core.py:281-323)
	const uint32_t npairs = misc[M_NPAIRS];
	uint32_t px_[16], p_mb = 0, p_i = 0, p_j = 0;
	bool q7 = false;
#pragma unroll
	for (int t = 0; t < 16; t++) px_[t] = 0;
	if ((uint32_t)tid < npairs) {
		p_i = pairs[tid];
		p_j = roles[p_i];
		const LDS(u32x4) *pa = (const LDS(u32x4) *)(pix0 + 32 * p_i), *pb = (const LDS(u32x4) *)(pix0 + 32 * (p_i + p_j));
		const u32x4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
		const uint32_t av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
		uint32_t wv[16];
#pragma unroll
		for (int k = 0; k < 8; k++) { wv[2 * k] = perm(bv[k], av[k], 0x05040100u); wv[2 * k + 1] = perm(bv[k], av[k], 0x07060302u); }
		const uint32_t pv = prev_px_final((int)p_i);
		deltas<16>(wv, pv, px_);
		if (!wide) {
			p_mb = two_byte_bits(px_) | (two_byte_bits(px_ + 8) << 16);
			if (big) q7 |= any_out_of_q7<16>(px_);
		} else {
			int pu = (int)pv;
#pragma unroll
			for (int i = 0; i < 32; i++) {
				const int v = px16(wv, i), dlt = v - pu;
				if (tok_two(dlt)) { p_mb |= 1u << i; q7 |= out_of_q7(dlt); }
				pu = v;
			}
		}
		roles[p_i + p_j] = (uint8_t)(0x80u | (uint32_t)__popc(p_mb));   // the leader's lane finds the pair's size here
	}
	if (npairs > (uint32_t)PAIR_FAST) {
		// more pairs than lanes (dense noise): the rest leave their bytes in HBM records and are copied into the image later
		uint8_t *rec0 = a.pairrec + ((size_t)sl * (NB / 2) + (size_t)g * (NBG / 2)) * PIPE_PAIR_REC;
		for (uint32_t e = PAIR_FAST + tid; e < npairs; e += ST) {
			const int i = pairs[e], j = roles[i], p = i + j;
			uint8_t *out = rec0 + (size_t)(e - PAIR_FAST) * PIPE_PAIR_REC;
			int n = 0;
			out[n++] = (uint8_t)(0x80 | j);  // core.py:290-294
			int prev = (int)prev_px_final(i);
			for (int t = 0; t < 32; t++) {
				const int v = (int)*(const LDS(uint16_t) *)(pix0 + 32 * ((t & 1) ? p : i) + 2 * (t >> 1));
				const int dlt = v - prev;
				if (tok_two(dlt)) { out[n++] = (uint8_t)(0xE0 | ((dlt >> 8) & 0x0F)); out[n++] = (uint8_t)(dlt & 0xFF); q7 |= out_of_q7(dlt); }
				else out[n++] = (uint8_t)(dlt & 0x7F);
				prev = v;
			}
			roles[p] = (uint8_t)(0x80u | (uint32_t)(n - 33));
		}
	}
	__syncthreads();
	STAMP(10);

	// ---- token bytes of every block along the final order, offsets inside the tile
	uint32_t off[4], run = 0;
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const int b = 256 * wave + 64 * s + lane;
		off[s] = 0;
		if (!active) continue;
		const uint32_t r = roles[b];
		uint32_t sz = 0;
		if (r == 0) {
			const bool fix = b == 0 ? g > 0 : roles[b - 1] != 0;
			if (__any(fix)) {
				if (fix) {
					// the block follows a meshed block (or opens the group): its first delta is against another pixel
					const int d0 = (int)*(const LDS(uint16_t) *)(pix0 + 32 * b), dlt = d0 - (int)prev_px_final(b);
					const uint32_t two = tok_two(dlt) ? 1u : 0u;
					x[s][0] = (x[s][0] & 0xFFFF0000u) | ((uint32_t)dlt & 0xFFFFu);
					const uint32_t mb = (info[s] & 0xFFFEu) | two;
					info[s] = (info[s] & ~((1u << 28) | 0x1FFFFFu)) | mb | ((uint32_t)__popc(mb) << 16) | ((two && out_of_q7(dlt)) ? 1u << 28 : 0u);
				}
			}
			sz = 16u + ((info[s] >> 16) & 31u);
			q7 |= ((info[s] >> 27) & 3u) != 0;
		} else if (r < 0x80u) sz = 33u + (roles[b + (int)r] & 0x7Fu);
		const uint32_t inc = wave_incl_scan(sz);
		off[s] = run + inc - sz;
		run += rdlane(inc, 63);
	}
	if (lane == 0) {
		misc[M_WTOT + wave] = run;
		uint32_t nd = 0;
#pragma unroll
		for (int s = 0; s < 4; s++) nd += (uint32_t)__popcll(bal[s]);
		if (nd) lds_add(&misc[M_NDIFF], nd);
	}
	if (__any(q7) && lane == 0) lds_or(&misc[M_STATUS], CCT_ST_Q7);
	__syncthreads();
	STAMP(11);
	uint32_t tilebase = 0, gtot = 0;
#pragma unroll
	for (int v = 0; v < SW; v++) { const uint32_t t = misc[M_WTOT + v]; if (v < wave) tilebase += t; gtot += t; }
#pragma unroll
	for (int s = 0; s < 4; s++) {
		off[s] += tilebase;
		const int b = 256 * wave + 64 * s + lane;
		const uint32_t r = active ? roles[b] : 0u;
		if (r >= 1u && r < 0x80u) boff[b] = (uint16_t)off[s];    // the pair's lane finds its offset here
	}
	// hand on the totals and ask for the predecessors' (one lane each); the answers are looked at after the image is built
	uint64_t lb_v = 0;
	if (wave == 0) {
		if (lane == 0) hand_store(&hand[g].total, H_VALID | (uint64_t)gtot | (uint64_t)npairs << 24 | (uint64_t)misc[M_NDIFF] << 40);
		if (lane < g && !(a.dbg & 2)) lb_v = hand_load(&hand[lane].total);
	}
	STAMP(12);

	// ---- the pixel stage becomes the payload image (from its byte 0: where the group starts in the slice is not known yet)
	LDS(uint8_t) *stg = lds + L_PIX;
	const bool eofb = last_group && a.e.eof >= 0;
	const uint32_t tot = gtot + (eofb ? 1u : 0u);
	{
		const uint32_t zc = min((uint32_t)(L_IMG_BYTES / 16), (tot + 15u) / 16u + 3u);
		for (uint32_t c = tid; c < zc; c += ST) *(LDS(u32x4) *)(stg + c * 16) = (u32x4){0, 0, 0, 0};
	}
	__syncthreads();
	STAMP(13);
	if (wave == 0 && lane < g && !(lb_v & H_VALID) && !(a.dbg & 2)) lb_v = hand_load(&hand[lane].total);   // asked again
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const int b = 256 * wave + 64 * s + lane;
		const uint32_t r = active ? roles[b] : 0xFFu;
		if (__all(r != 0 || (info[s] & 0xFFFFu) == 0)) {   // no two-byte token in the wave's 64 blocks
			if (r == 0) emit16_short(x[s], off[s], stg);
		} else if (r == 0) (void)emit16(x[s], info[s] & 0xFFFFu, off[s], stg, ttab);
	}
	if ((uint32_t)tid < npairs) {
		uint32_t o = boff[p_i];
		lds_or((LDS(uint32_t) *)(stg + (o & ~3u)), (0x80u | p_j) << ((o & 3u) * 8u));   // core.py:290-294
		o = emit16(px_, p_mb & 0xFFFFu, o + 1u, stg, ttab);
		(void)emit16(px_ + 8, p_mb >> 16, o, stg, ttab);
	}
	if (npairs > (uint32_t)PAIR_FAST) {
		const uint8_t *rec0 = a.pairrec + ((size_t)sl * (NB / 2) + (size_t)g * (NBG / 2)) * PIPE_PAIR_REC;
		for (uint32_t e = PAIR_FAST + tid; e < npairs; e += ST) {
			const int i = pairs[e], p = i + roles[i];
			const uint32_t n = 33u + (roles[p] & 0x7Fu), o = boff[i];
			const uint8_t *src = rec0 + (size_t)(e - PAIR_FAST) * PIPE_PAIR_REC;
			for (uint32_t j = 0; j < n; j++) stg[o + j] = src[j];
		}
	}
	if (eofb && tid == 0) stg[gtot] = (uint8_t)a.e.eof;   // core.py:329-330 (nothing else writes this byte)
	if (wave == 0) {
		// the bytes before the group: the predecessors' totals, 64 at a time
		uint32_t bytes = 0, nj = 0, ndf = 0, spins = 0;
		if (a.dbg & 2) bytes = (uint32_t)g * (uint32_t)(NBG * 20);
		else for (int c0 = 0; c0 < g; c0 += 64) {
			uint64_t v = c0 == 0 ? lb_v : 0ull;
			bool ok = c0 + lane >= g || (v & H_VALID) != 0;
			while (!__all(ok)) {
				if (!ok) { v = hand_load(&hand[c0 + lane].total); ok = (v & H_VALID) != 0; }
				if (++spins > SPIN_LIMIT) { if (lane == 0) lds_or(&misc[M_STATUS], CCT_ST_INTERNAL); break; }
				if (!__all(ok)) __builtin_amdgcn_s_sleep(4);
			}
			if (c0 + lane >= g) v = 0;
			bytes += rdlane(wave_incl_scan((uint32_t)v & 0xFFFFFFu), 63);
			nj += rdlane(wave_incl_scan((uint32_t)(v >> 24) & 0xFFFFu), 63);
			ndf += rdlane(wave_incl_scan((uint32_t)(v >> 40) & 0xFFFFu), 63);
		}
		if (lane == 0) { misc[M_BASE] = bytes; misc[M_ACC_JUMP] = nj; misc[M_ACC_DIFF] = ndf; }
	}
	__syncthreads();
	STAMP(14);

	// ---- flush.  Byte q of the image belongs at base + q of the slice's payload: 16-byte chunks of the payload are put
	// together from five image dwords (v_perm by the byte phase) and stored whole; the two ends shared with the neighbouring
	// groups go byte by byte.
	uint32_t stat = misc[M_STATUS];
	const uint32_t base = misc[M_BASE];
	{
		const uint32_t head = base & 15u;
		const uint32_t end = head + tot;                                      // in bytes from the chunk the group starts in
		const size_t base16 = (size_t)(base & ~15u);
		const bool room = base16 + ((end + 15u) & ~15u) <= a.e.stride;
		uint8_t *out = a.e.payload + (size_t)sl * a.e.stride + base16;
		const uint32_t c_first = head ? 1u : 0u;
		const uint32_t c_end = last_group ? (end + 15u) / 16u : end / 16u;   // the last group owns the padding of the slice
		if (room) {
			if (head == 0) {
				for (uint32_t c = tid; c < c_end; c += ST) *reinterpret_cast<u32x4 *>(out + (size_t)c * 16) = *(const LDS(u32x4) *)(stg + c * 16);
			} else {
				const uint32_t sel = 0x03020100u + 0x01010101u * ((0u - head) & 3u);
				for (uint32_t c = c_first + tid; c < c_end; c += ST) {
					const LDS(uint32_t) *w = (const LDS(uint32_t) *)(stg + ((16u * c - head) & ~3u));
					const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
					*reinterpret_cast<u32x4 *>(out + (size_t)c * 16) = (u32x4){perm(d1, d0, sel), perm(d2, d1, sel), perm(d3, d2, sel), perm(d4, d3, sel)};
				}
				if (tid < 16 && (uint32_t)tid >= head && (uint32_t)tid < end) out[tid] = stg[(uint32_t)tid - head];
			}
			if (!last_group && tid >= 16 && tid < 32) {
				const uint32_t i = c_end * 16u + (uint32_t)(tid - 16);
				if (i < end && (i >= 16u || !head)) out[i] = stg[i - head];
			}
		} else stat |= CCT_ST_CAP;
	}
	if (a.e.roles_out) {
		uint8_t *ro = a.e.roles_out + (size_t)sl * NB + abs0;
		for (int i = tid; i < nbg; i += ST) { const uint32_t r = roles[i]; ro[i] = (uint8_t)(r >= 0x80u ? ROLE_PARTNER : r); }
	}
	if (last_group && tid == 0) {
		const uint32_t total = base + gtot;
		const uint32_t size = total + (a.e.eof >= 0 ? 1u : 0u);
		const bool cap = (size_t)((size + 15u) & ~15u) > a.e.stride;
		if (cap) stat |= CCT_ST_CAP;
		a.e.sizes[sl] = cap ? 0u : size;
		if (a.e.stats) {
			uint32_t *sts = a.e.stats + (size_t)sl * 4;
			const uint32_t njump = misc[M_ACC_JUMP] + npairs;
			const uint32_t nfull = total - (uint32_t)N - njump;
			sts[0] = (uint32_t)N - nfull; sts[1] = nfull; sts[2] = njump; sts[3] = seg ? misc[M_ACC_DIFF] + misc[M_NDIFF] : 0u;
		}
	}
	if (stat && tid == 0) atomicOr(a.e.status + sl, stat);   // the launch zeroes status[]
	if (STAMPS && lane == 0 && stamps) {
		st_[15] = __builtin_amdgcn_s_memtime();
		uint64_t *o = stamps + ((size_t)blockIdx.x * SW + wave) * (N_STAMPS + 2);
		for (int k = 0; k < N_STAMPS; k++) o[k] = st_[k];
		o[N_STAMPS] = rt0_; o[N_STAMPS + 1] = __builtin_amdgcn_s_memrealtime();
	}
}

}  // namespace

static void report_stream_stamps(const uint64_t *d_buf, size_t nwaves)
{
	const int W = N_STAMPS + 2;
	std::vector<uint64_t> h(nwaves * W);
	if (hipMemcpy(h.data(), d_buf, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
	static const char *const name[N_STAMPS - 1] = {"tables+ticket", "loads+permute", "barrier 1", "analysis", "masks", "barrier 2", "resolve", "carry wait",
	                                               "barrier 4", "pairs", "sizes", "totals", "zero", "emit+lookback", "flush"};
	double sum[N_STAMPS] = {0};
	std::vector<double> life, start, endt;
	uint64_t t0 = ~0ull;
	for (size_t i = 0; i < nwaves; i++) if (h[i * W + N_STAMPS]) t0 = std::min(t0, h[i * W + N_STAMPS]);
	size_t m = 0;
	for (size_t i = 0; i < nwaves; i++) {
		const uint64_t *o = h.data() + i * W;
		if (!o[N_STAMPS]) continue;
		m++;
		for (int k = 0; k + 1 < N_STAMPS; k++) sum[k] += (double)(o[k + 1] - o[k]);
		life.push_back((o[N_STAMPS + 1] - o[N_STAMPS]) * 0.01); start.push_back((o[N_STAMPS] - t0) * 0.01); endt.push_back((o[N_STAMPS + 1] - t0) * 0.01);
	}
	if (!m) return;
	double all = 0, lf = 0;
	for (int k = 0; k + 1 < N_STAMPS; k++) all += sum[k];
	for (double v : life) lf += v;
	std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end()); std::sort(endt.begin(), endt.end());
	fprintf(stderr, "[stream stamps] %zu waves, %.0f shader cycles per wave, life us p50 %.1f p90 %.1f max %.1f, mean concurrency %.0f waves, starts p50 %.1f p100 %.1f, ends p100 %.1f\n",
	        m, all / m, life[m / 2], life[m * 9 / 10], life[m - 1], lf / endt[m - 1], start[m / 2], start[m - 1], endt[m - 1]);
	for (int k = 0; k + 1 < N_STAMPS; k++) fprintf(stderr, "    %-16s %8.0f cycles  %5.1f %%\n", name[k], sum[k] / m, 100.0 * sum[k] / all);
}

hipError_t launch_encode_stream(const StreamArgs &sa, int n, hipStream_t s)
{
	// tickets and hand-off words start from zero in every launch (one memset node in a captured graph)
	hipError_t e = hipMemsetAsync(sa.hand, 0, stream_ws_bytes(n, sa.gps), s);
	if (e != hipSuccess) return e;
	e = hipMemsetAsync(sa.e.status, 0, (size_t)n * sizeof(uint32_t), s);  // every group ORs its bits in
	if (e != hipSuccess) return e;
	const bool sg = (sa.e.flags & CCT_FLAG_SIGNED_SEG) != 0;
	static const bool stamps_on = getenv("CCT_STREAM_STAMPS") != nullptr;
	uint64_t *d_st = nullptr;
	const size_t nw = (size_t)n * sa.gps * sa.tpg;
	if (stamps_on) {
		if (hipMalloc(&d_st, nw * (N_STAMPS + 2) * 8) != hipSuccess) return hipErrorOutOfMemory;
		(void)hipMemset(d_st, 0, nw * (N_STAMPS + 2) * 8);
	}
#define CCT_LAUNCH(SWV)                                                                                                          \
	do {                                                                                                                           \
		if (stamps_on) {                                                                                                             \
			if (sg) hipLaunchKernelGGL((stream_kernel<SWV, true, true>), dim3(n * sa.gps), dim3(64 * SWV), 0, s, sa, d_st);          \
			else hipLaunchKernelGGL((stream_kernel<SWV, false, true>), dim3(n * sa.gps), dim3(64 * SWV), 0, s, sa, d_st);           \
		} else if (sg) hipLaunchKernelGGL((stream_kernel<SWV, true, false>), dim3(n * sa.gps), dim3(64 * SWV), 0, s, sa, d_st);    \
		else hipLaunchKernelGGL((stream_kernel<SWV, false, false>), dim3(n * sa.gps), dim3(64 * SWV), 0, s, sa, d_st);            \
	} while (0)
	switch (sa.tpg) {
	case 1: CCT_LAUNCH(1); break;
	case 2: CCT_LAUNCH(2); break;
	case 4: CCT_LAUNCH(4); break;
	default: return hipErrorInvalidValue;
	}
#undef CCT_LAUNCH
	if (stamps_on) {
		(void)hipStreamSynchronize(s);
		report_stream_stamps(d_st, nw);
		(void)hipFree(d_st);
	}
	return hipGetLastError();
}

}  // namespace cct
