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
//                 XOR-swizzled by row, so that the later 2-byte gathers hit distinct banks.
//   LDS gather    lane = half a block (8 pixels): one ds_read_b128 of the pattern table (LDS byte
//                 offset of each traversal position inside a tile, swizzle included), eight
//                 ds_read_u16, then the 8 pixels go back to LDS in traversal order (ds_write_b128).
//   pass 1        |delta| > 64 flags per half block, pair-reduced with one shuffle; difficult blocks
//                 (cluster.py:51-59) are compacted in order with ballots; one WAVE per difficult
//                 block evaluates its 63 mesh candidates (cluster.py:122-158) and ballots the mask.
//   resolve       greedy first fit (cluster.py:79-190) per island of difficult blocks, one lane each.
//   pass 2        token sizes per half block -> wave scan -> staging -> aligned 16-byte stores.
//
// The loop is software-pipelined so that one iteration needs two workgroup barriers:
//   pass 1:  P1 gather(s) | P2 analyse(s), append(s-1), masks(s-2), stage(s+1) -> LDS, load(s+5)
//   pass 2:  P1 gather(s), write tokens(s-2) | P2 size tokens(s-1), flush(s-2), stage(s+1), load(s+5)
// Barriers are `s_waitcnt lgkmcnt(0); s_barrier` (LDS only), so the global loads stay in flight
// across them; __syncthreads() is used only where HBM-resident lists/roles must be published.
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = __shfl_up(v, d);
		if (lane >= d) v += t;
	}
	return v;
}

struct DiffListT {
	uint32_t *l_idx; uint8_t *l_cur; uint64_t *l_mask;
	uint32_t *g_idx; uint8_t *g_cur; uint64_t *g_mask;
	__device__ __forceinline__ void set(uint32_t e, uint32_t idx, uint32_t cur) const
	{
		if (e < ENC_LIST_CAP) { l_idx[e] = idx; l_cur[e] = (uint8_t)cur; }
		else { g_idx[e - ENC_LIST_CAP] = idx; g_cur[e - ENC_LIST_CAP] = (uint8_t)cur; }
	}
	__device__ __forceinline__ uint32_t idx(uint32_t e) const { return e < ENC_LIST_CAP ? l_idx[e] : g_idx[e - ENC_LIST_CAP]; }
	__device__ __forceinline__ uint32_t cur(uint32_t e) const { return e < ENC_LIST_CAP ? l_cur[e] : g_cur[e - ENC_LIST_CAP]; }
	__device__ __forceinline__ uint64_t mask(uint32_t e) const { return e < ENC_LIST_CAP ? l_mask[e] : g_mask[e - ENC_LIST_CAP]; }
	__device__ __forceinline__ void set_mask(uint32_t e, uint64_t m) const
	{
		if (e < ENC_LIST_CAP) l_mask[e] = m; else g_mask[e - ENC_LIST_CAP] = m;
	}
};

__global__ void __launch_bounds__(TT) encode_tiles_kernel(TileEncArgs ta)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const EncArgs &a = ta.e;
	uint8_t *raster = smem + L_RASTER;
	uint16_t *dlin = reinterpret_cast<uint16_t *>(smem + L_DLIN);
	const uint8_t *pat = smem + L_PAT;
	uint8_t *stg = smem + L_LIST;
	uint64_t *l_mask = reinterpret_cast<uint64_t *>(smem + L_LIST);
	uint32_t *l_idx = reinterpret_cast<uint32_t *>(smem + L_LIST + ENC_LIST_CAP * 8);
	uint8_t *l_cur = smem + L_LIST + ENC_LIST_CAP * 12;
	uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + L_MISC + MISC_WCNT);
	uint32_t *wtot = reinterpret_cast<uint32_t *>(smem + L_MISC + MISC_WTOT);
	uint32_t *ctr = reinterpret_cast<uint32_t *>(smem + L_MISC + MISC_CTR);
	uint32_t *torg = reinterpret_cast<uint32_t *>(smem + L_MISC + MISC_TORG);
	uint8_t *tori = smem + L_MISC + MISC_TORI;
	uint8_t *role_lds = smem + L_ROLE;

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int sl = blockIdx.x;
	const int N = a.N, NB = a.NB;
	const int NS = N / STP;
	const bool seg = (a.flags & CCT_FLAG_SEGMENTATION) != 0;
	const bool sgn = (a.flags & CCT_FLAG_SIGNED_SEG) != 0;
	const uint16_t *img = a.images + (size_t)sl * N;
	uint8_t *role = a.ws_role ? a.ws_role + (size_t)sl * NB : role_lds;
	DiffListT dl{l_idx, l_cur, l_mask,
	             a.ws_lidx + (size_t)sl * NB, a.ws_lcur + (size_t)sl * NB, a.ws_lmask + (size_t)sl * NB};

	// ---- one-time LDS tables: patterns, tile origins / orientations
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(ta.patterns);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem + L_PAT);
		const int n16 = ta.n_orient * (TPX * 2 / 16);
		for (int i = tid; i < n16; i += TT) dst[i] = src[i];
		for (int i = tid; i < ta.n_tiles; i += TT) { torg[i] = ta.tile_org[i]; tori[i] = ta.tile_orient[i]; }
		if (tid < 4) ctr[tid] = 0;
	}
	__syncthreads();

	// lane's share of a super-tile load: tile j = tid>>9, 16-byte chunk c = tid&511 of that tile
	const int ld_j = tid >> 9, ld_c = tid & 511;
	const int ld_row = ld_c >> 3, ld_col = (ld_c & 7) ^ (ld_row & 7);  // XOR swizzle on the SOURCE column
	const size_t ld_off = (size_t)ld_row * ta.row_pitch + (size_t)ld_col * 8;
	auto load_st = [&](int s) -> u32x4 {
		const uint16_t *p = img + torg[2 * s + ld_j] + ld_off;
		return *reinterpret_cast<const u32x4 *>(p);
	};
	auto stage_to_lds = [&](int s, const u32x4 &v) {
		*reinterpret_cast<u32x4 *>(raster + (s & 1) * 16384 + tid * 16) = v;
	};
	// gather this lane's 8 traversal-consecutive pixels of super-tile s and store them linearly
	const int g_j = tid >> 9;                 // tile inside the super-tile
	const int g_k = (tid * 8) & (TPX - 1);    // first traversal position inside that tile
	auto gather = [&](int s, int slot) {
		const uint8_t *rt = raster + (s & 1) * 16384 + g_j * 8192;
		const u32x4 pe = *reinterpret_cast<const u32x4 *>(pat + (int)tori[2 * s + g_j] * 8192 + g_k * 2);
		uint32_t o[8] = {pe.x & 0xFFFFu, pe.x >> 16, pe.y & 0xFFFFu, pe.y >> 16,
		                 pe.z & 0xFFFFu, pe.z >> 16, pe.w & 0xFFFFu, pe.w >> 16};
		uint32_t p[8];
#pragma unroll
		for (int i = 0; i < 8; i++) p[i] = *reinterpret_cast<const uint16_t *>(rt + o[i]);
		u32x4 out;
		out.x = p[0] | (p[1] << 16); out.y = p[2] | (p[3] << 16);
		out.z = p[4] | (p[5] << 16); out.w = p[6] | (p[7] << 16);
		*reinterpret_cast<u32x4 *>(dlin + slot * STP + tid * 8) = out;
	};
	// traversal-ordered pixel k from the ring; m = (current s) % 3 and s give the slot of k's super-tile
	auto slot_of = [&](int st_idx, int s, int m) -> int {
		int x = m + (st_idx - s) + 3;
		x -= (x >= 3) ? 3 : 0;
		x -= (x >= 3) ? 3 : 0;
		return x;
	};
	auto DL = [&](int k, int s, int m) -> int {
		return (int)dlin[slot_of(k >> 13, s, m) * STP + (k & (STP - 1))];
	};
	auto SX = [&](int v) -> int { return sgn ? (int)(int16_t)(uint16_t)v : v; };

	uint32_t ndiff = 0;
	// ================================================================== pass 1
	if (seg) {
		u32x4 r0 = load_st(0), r1, r2, r3;
		if (1 < NS) r1 = load_st(1);
		if (2 < NS) r2 = load_st(2);
		if (3 < NS) r3 = load_st(3);
		stage_to_lds(0, r0);
		if (4 < NS) r0 = load_st(4);

		uint32_t p_diff = 0, p_cur = 0, p_rank = 0;  // this lane's block of super-tile s-1
		uint32_t start_prev = 0;                      // first list record of super-tile s-2
		bool heavy = false;                           // records spilled to HBM: use full barriers

#define P1_ITER(S, RNEXT)                                                                          \
	{                                                                                                \
		const int s = (S);                                                                             \
		const int m = s % 3;                                                                           \
		if (heavy) __syncthreads(); else lds_barrier();                                                \
		if (s < NS) gather(s, m);                                                                      \
		if (heavy) __syncthreads(); else lds_barrier();                                                \
		/* (a) analyse super-tile s */                                                                 \
		uint32_t c_diff = 0, c_cur = 0, c_rank = 0;                                                    \
		if (s < NS) {                                                                                  \
			const int k0 = s * STP + tid * 8;                                                            \
			const u32x4 pv = *reinterpret_cast<const u32x4 *>(dlin + m * STP + tid * 8);                 \
			int px[8] = {SX(pv.x & 0xFFFF), SX(pv.x >> 16), SX(pv.y & 0xFFFF), SX(pv.y >> 16),           \
			             SX(pv.z & 0xFFFF), SX(pv.z >> 16), SX(pv.w & 0xFFFF), SX(pv.w >> 16)};          \
			int prev = (k0 > 0) ? SX(DL(k0 - 1, s, m)) : px[0];                                          \
			uint32_t f0 = 0, cnt = 0;                                                                    \
			{ const int d = px[0] - prev; f0 = (d > 64 || d < -64) ? 1u : 0u; }                          \
			_Pragma("unroll") for (int i = 1; i < 8; i++) {                                              \
				const int d = px[i] - px[i - 1];                                                           \
				cnt += (d > 64 || d < -64) ? 1u : 0u;                                                      \
			}                                                                                            \
			const uint32_t h = tid & 1;                                                                  \
			if (h) cnt += f0; /* transition between the two halves is an inner one */                    \
			const uint32_t chg = cnt + __shfl_xor(cnt, 1);                                               \
			const uint32_t enter = __shfl(f0, lane & ~1);                                                \
			c_diff = (chg >= 8u && h == 0) ? 1u : 0u; /* cluster.py:58 */                                \
			c_cur = chg + ((k0 > 0) ? enter : 0u);     /* cluster.py:110 (block 0: Q4, see masks) */      \
			if (h == 0) role[s * 512 + (tid >> 1)] = 0;                                                  \
		}                                                                                              \
		{                                                                                              \
			const uint64_t bal = __ballot(c_diff != 0);                                                  \
			c_rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));                                  \
			if (lane == 0) wcnt[(s & 1) * 16 + wave] = (uint32_t)__popcll(bal);                          \
		}                                                                                              \
		/* (b) append records of super-tile s-1 (counts published one iteration ago) */                \
		const uint32_t start_cur = ndiff;                                                              \
		if (s >= 1 && s <= NS) {                                                                       \
			uint32_t base = 0, tot = 0;                                                                  \
			for (int w = 0; w < NW; w++) {                                                               \
				const uint32_t x = wcnt[((s - 1) & 1) * 16 + w];                                           \
				if (w < wave) base += x;                                                                   \
				tot += x;                                                                                  \
			}                                                                                            \
			if (p_diff) dl.set(ndiff + base + p_rank, (uint32_t)((s - 1) * 512 + (tid >> 1)), p_cur);    \
			ndiff += tot;                                                                                \
			if (ndiff > ENC_LIST_CAP) heavy = true;                                                      \
		}                                                                                              \
		/* (c) candidate masks of super-tile s-2: its look-ahead lies in s-2, s-1 */                   \
		if (s >= 2) {                                                                                  \
			for (uint32_t e = start_prev + wave; e < start_cur; e += NW) {                               \
				const int i = (int)dl.idx(e);                                                              \
				const uint32_t cur = dl.cur(e);                                                            \
				const int p = i + lane;                                                                    \
				bool fit = false;                                                                          \
				if (lane >= 1 && p < NB) {                                                                 \
					const int ka = i * 16, kb = p * 16;                                                      \
					uint32_t up = 0;                                                                         \
					int bprev = 0;                                                                           \
					_Pragma("unroll") for (int t = 0; t < 16; t++) {                                         \
						const int av = SX(DL(ka + t, s, m)), bv = SX(DL(kb + t, s, m));                        \
						if (t > 0) up += (av - bprev >= 65) ? 1u : 0u;                                         \
						up += (bv - av >= 65) ? 1u : 0u;                                                       \
						bprev = bv;                                                                            \
					}                                                                                        \
					fit = (i == 0) ? true : ((up + 1u) < (cur - 2u));                                        \
				}                                                                                          \
				const uint64_t mk = __ballot(fit);                                                         \
				if (lane == 0) dl.set_mask(e, mk);                                                         \
			}                                                                                            \
		}                                                                                              \
		start_prev = start_cur;                                                                        \
		p_diff = c_diff; p_cur = c_cur; p_rank = c_rank;                                               \
		/* (d) next super-tile: registers -> LDS raster image, refill the register */                 \
		if (s + 1 < NS) stage_to_lds(s + 1, RNEXT);                                                    \
		if (s + 5 < NS) RNEXT = load_st(s + 5);                                                        \
	}

		for (int sb = 0; sb < NS + 2; sb += 4) {
			P1_ITER(sb + 0, r1)
			P1_ITER(sb + 1, r2)
			P1_ITER(sb + 2, r3)
			P1_ITER(sb + 3, r0)
		}
#undef P1_ITER
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
						role[i] = (uint8_t)j;
						role[i + j] = ROLE_PARTNER;
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
	{
		u32x4 r0 = load_st(0), r1, r2, r3;
		if (1 < NS) r1 = load_st(1);
		if (2 < NS) r2 = load_st(2);
		if (3 < NS) r3 = load_st(3);
		stage_to_lds(0, r0);
		if (4 < NS) r0 = load_st(4);

		// token state of this lane's half block of super-tile s-1, carried from P2 to the next P1
		u32x4 t_own = {0, 0, 0, 0}, t_par = {0, 0, 0, 0};
		int t_prev = 0, t_r = 0;
		uint32_t t_nbytes = 0, t_excl = 0;
		uint32_t keep_byte = 0;  // remainder byte carried from a flush to the next token write

		auto put = [&](uint8_t *&w, int d) {
			if (d < -63 || d > 64) {  // full delta, core.py:322-323
				*w++ = (uint8_t)(0xE0 | ((d >> 8) & 0x0F));
				*w++ = (uint8_t)(d & 0xFF);
			} else {                  // short delta, core.py:316-319
				*w++ = (uint8_t)(d & 0x7F);
			}
		};
		auto cnt2 = [&](int d, uint32_t &n2) {
			n2 += (d < -63 || d > 64) ? 1u : 0u;
			my_q7 |= (d < -2047 || d > 2048);
		};

#define P2_ITER(S, RNEXT)                                                                          \
	{                                                                                                \
		const int s = (S);                                                                             \
		const int m = s % 3;                                                                           \
		lds_barrier();                                                                                 \
		if (s < NS) gather(s, m);                                                                      \
		/* write tokens of super-tile s-2 (sized one iteration ago) */                                 \
		if (s >= 2 && s <= NS + 1) {                                                                   \
			if ((uint32_t)tid < carry) stg[tid] = (uint8_t)keep_byte;                                    \
			uint32_t wbase = 0;                                                                          \
			for (int w = 0; w < wave; w++) wbase += wtot[((s - 2) & 1) * 16 + w];                        \
			if (t_nbytes) {                                                                              \
				uint8_t *w = stg + carry + wbase + t_excl;                                                 \
				const uint32_t ow[4] = {t_own.x, t_own.y, t_own.z, t_own.w};                               \
				const uint32_t pw[4] = {t_par.x, t_par.y, t_par.z, t_par.w};                               \
				int prev = t_prev;                                                                         \
				if (t_r == 0) {                                                                            \
					_Pragma("unroll") for (int i = 0; i < 8; i++) {                                          \
						const int v = (int)((ow[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                         \
						put(w, v - prev);                                                                      \
						prev = v;                                                                              \
					}                                                                                        \
				} else {                                                                                   \
					if ((tid & 1) == 0) *w++ = (uint8_t)(0x80 | t_r); /* jump tag, core.py:290-294 */        \
					_Pragma("unroll") for (int i = 0; i < 8; i++) {                                          \
						const int va = (int)((ow[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                        \
						const int vb = (int)((pw[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                        \
						put(w, va - prev);                                                                     \
						put(w, vb - va);                                                                       \
						prev = vb;                                                                             \
					}                                                                                        \
				}                                                                                          \
			}                                                                                            \
		}                                                                                              \
		if (s == NS + 1) { /* last flush: stage the EOF byte (core.py:329-330) and zero the pad */      \
			uint32_t bytes = 0;                                                                          \
			for (int w = 0; w < NW; w++) bytes += wtot[((s - 2) & 1) * 16 + w];                          \
			uint32_t total = carry + bytes;                                                              \
			if (a.eof >= 0) { if (tid == 0) stg[total] = (uint8_t)a.eof; total += 1; }                   \
			if (tid >= 1 && tid <= 15) stg[total + tid - 1] = 0;                                         \
		}                                                                                              \
		lds_barrier();                                                                                 \
		/* flush super-tile s-2 */                                                                     \
		if (s >= 2 && s <= NS + 1) {                                                                   \
			uint32_t bytes = 0;                                                                          \
			for (int w = 0; w < NW; w++) bytes += wtot[((s - 2) & 1) * 16 + w];                          \
			const bool last = (s == NS + 1);                                                             \
			uint32_t total = carry + bytes;                                                              \
			if (last && a.eof >= 0) total += 1; /* EOF byte staged before the barrier */                   \
			const uint32_t nflush = last ? ((total + 15u) & ~15u) : (total & ~15u);                      \
			if ((size_t)out_pos + nflush > a.stride) cap_hit = true;                                     \
			if (!cap_hit) {                                                                              \
				const u32x4 *src = reinterpret_cast<const u32x4 *>(stg);                                   \
				u32x4 *dst = reinterpret_cast<u32x4 *>(out + out_pos);                                     \
				for (uint32_t u = tid; u < nflush / 16u; u += TT) dst[u] = src[u];                         \
			}                                                                                            \
			const uint32_t rem = last ? 0u : (total - nflush);                                           \
			if ((uint32_t)tid < rem) keep_byte = stg[nflush + tid];                                      \
			if (last && tid == 0) {                                                                      \
				a.sizes[sl] = cap_hit ? 0u : (out_pos + total);                                            \
			}                                                                                            \
			out_pos += nflush;                                                                           \
			carry = rem;                                                                                 \
		}                                                                                              \
		/* size the tokens of super-tile s-1 */                                                        \
		if (s >= 1 && s <= NS) {                                                                       \
			const int e = s - 1;                                                                         \
			const int me = (m + 2) % 3; /* slot of super-tile s-1 */                                     \
			const int b = e * 512 + (tid >> 1);                                                          \
			const int h = tid & 1;                                                                       \
			const int k0 = e * STP + tid * 8;                                                            \
			t_own = *reinterpret_cast<const u32x4 *>(dlin + me * STP + tid * 8);                         \
			t_r = seg ? (int)role[b] : 0;                                                                \
			t_nbytes = 0;                                                                                \
			if (t_r != ROLE_PARTNER) {                                                                   \
				if (h == 1) t_prev = DL(k0 - 1, s, m);                                                     \
				else if (b == 0) t_prev = 0;                                                               \
				else {                                                                                     \
					int q = b - 1;                                                                           \
					int rq = seg ? (int)role[q] : 0;                                                         \
					while (rq == ROLE_PARTNER) { q--; rq = (int)role[q]; }                                   \
					t_prev = DL((q + rq) * 16 + 15, s, m);                                                   \
				}                                                                                          \
				const uint32_t ow[4] = {t_own.x, t_own.y, t_own.z, t_own.w};                               \
				uint32_t n2 = 0;                                                                           \
				int prev = t_prev;                                                                         \
				if (t_r == 0) {                                                                            \
					_Pragma("unroll") for (int i = 0; i < 8; i++) {                                          \
						const int v = (int)((ow[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                         \
						cnt2(v - prev, n2);                                                                    \
						prev = v;                                                                              \
					}                                                                                        \
					t_nbytes = 8 + n2;                                                                       \
				} else {                                                                                   \
					const int kp = (b + t_r) * 16 + h * 8;                                                   \
					t_par = *reinterpret_cast<const u32x4 *>(dlin + slot_of(kp >> 13, s, m) * STP + (kp & (STP - 1))); \
					if (h == 1) { prev = DL(kp - 1, s, m); t_prev = prev; }                                  \
					const uint32_t pw[4] = {t_par.x, t_par.y, t_par.z, t_par.w};                             \
					_Pragma("unroll") for (int i = 0; i < 8; i++) {                                          \
						const int va = (int)((ow[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                        \
						const int vb = (int)((pw[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);                        \
						cnt2(va - prev, n2);                                                                   \
						cnt2(vb - va, n2);                                                                     \
						prev = vb;                                                                             \
					}                                                                                        \
					t_nbytes = 16 + n2 + (h == 0 ? 1u : 0u);                                                 \
					if (h == 0) my_jump += 1;                                                                \
				}                                                                                          \
				my_full += n2;                                                                             \
			}                                                                                            \
			if (a.roles_out && h == 0) a.roles_out[(size_t)sl * NB + b] = (uint8_t)t_r;                  \
			const uint32_t inc = wave_incl_scan(t_nbytes, lane);                                         \
			t_excl = inc - t_nbytes;                                                                     \
			if (lane == 63) wtot[(e & 1) * 16 + wave] = inc;                                             \
		}                                                                                              \
		if (s + 1 < NS) stage_to_lds(s + 1, RNEXT);                                                    \
		if (s + 5 < NS) RNEXT = load_st(s + 5);                                                        \
	}

		for (int sb = 0; sb < NS + 2; sb += 4) {
			P2_ITER(sb + 0, r1)
			P2_ITER(sb + 1, r2)
			P2_ITER(sb + 2, r3)
			P2_ITER(sb + 3, r0)
		}
#undef P2_ITER
	}
	// ---- statistics / status
	if (my_q7) atomicOr(&ctr[0], CCT_ST_Q7);
	if (a.stats) {
		if (my_full) atomicAdd(&ctr[1], my_full);
		if (my_jump) atomicAdd(&ctr[2], my_jump);
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
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_tiles_kernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(encode_tiles_kernel, dim3(n), dim3(TT), lds, s, ta);
	return hipGetLastError();
}

}  // namespace cct
