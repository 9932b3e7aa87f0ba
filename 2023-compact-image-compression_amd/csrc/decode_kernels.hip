// CompaCT decode, token stream -> raster (core.py:423-520), one workgroup per slice.
//
// The token stream is a 2-state automaton over bytes ("token start" / "second byte of a full
// token"), so token boundaries come from a scan of 2->2 state maps: every lane parses a 16-byte
// segment under BOTH entry hypotheses, a wave/workgroup scan composes the maps, and the lane
// then knows its true entry state, how many pixels precede it and the running pixel value
// (pixel values are a prefix sum of the signed deltas, core.py:503-505, 514-515).
//
//   pass A  parse the payload, collect the mesh-jump tokens (core.py:484-494) with their pixel
//           ordinals;
//   resolve one lane replays the ~100-700 jumps in stream order against a 64-bit window of
//           already-claimed partner blocks -> role[b] (0 single, 1..63 pair leader, 0xFF partner);
//           a workgroup scan over role[] then gives every 16-pixel stream slot its block;
//   pass B  parse again, now scattering each pixel to raster position O[block*bs + t].
//
// Reserved tag bytes (110xxxxx, 1111xxxx) decode as the reference decodes them: one byte, the previous pixel
// repeats (core.py:496-520 takes no branch).  Streams a reference encoder cannot produce for which the reference
// has no defined result (a jump that is not at a block boundary, two jump bytes in a row, a jump onto a claimed
// block, truncation) set CCT_ST_STREAM instead of replaying its accidental behaviour on them; see DESIGN.md.
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

struct SegStats {
	uint32_t exit_state;  // 1: segment ends inside a full token
	uint32_t npix;        // pixel tokens that START in the segment
	uint32_t njump;       // jump tokens in the segment
	int32_t sdelta;       // sum of their deltas
};

__device__ __forceinline__ int tok_delta_full(uint32_t b0, uint32_t b1)
{
	int x = (int)(((b0 << 8) | b1) & 0xFFFu);
	return x > 2048 ? x - 4096 : x;  // signed(x, 12): `x > max/2`, core.py:56-60
}
__device__ __forceinline__ int tok_delta_short(uint32_t b0)
{
	int x = (int)(b0 & 0x7Fu);
	return x > 64 ? x - 128 : x;  // signed(x, 7)
}

// bytes of the segment come as 4 little-endian words + the first byte of the next segment
__device__ __forceinline__ uint32_t seg_byte(const uint4 &w, uint32_t nxt, int i)
{
	const uint32_t word = (i < 4) ? w.x : (i < 8) ? w.y : (i < 12) ? w.z : (i < 16) ? w.w : nxt;
	return (word >> ((i & 3) * 8)) & 0xFFu;
}

// ---- a 16-byte segment as bit masks (bit i = byte i) --------------------------------------------------------------
// Walking the bytes one at a time costs ~30 instructions per byte, twice (both entry states), on a workgroup that is bound
// by instruction issue (16 waves on a CU).  Everything the scans need is a property of three byte classes, so the classes
// are extracted four bytes at a time (SWAR) and the token structure follows from mask arithmetic:
//   F  bytes 1110xxxx (may open a two-byte full delta)     J  bytes 10xxxxxx (jump)     S  bytes 0xxxxxxx (short delta)
// A byte is the second byte of a full delta iff the byte before it is an F byte that is not itself a second byte: inside a
// run of F bytes the roles alternate.  That is the "escaped character" problem of JSON scanners; the carry of one
// subtraction resolves all runs at once (Langdale & Lemire, simdjson: find_escaped).
struct SegMasks { uint32_t F, J, S, V; };

// the MSBs of the four bytes of m (nothing else set) as bits 0..3
__device__ __forceinline__ uint32_t msb_nibble(uint32_t m) { return ((m >> 7) * 0x01020408u) >> 24; }

__device__ __forceinline__ SegMasks seg_masks(const uint4 &w, int nvalid)
{
	const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
	SegMasks m{0, 0, 0, nvalid >= 16 ? 0xFFFFu : ((1u << max(nvalid, 0)) - 1u)};
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t x = ws[k];
		const uint32_t t = (x & 0xF0F0F0F0u) ^ 0xE0E0E0E0u;  // zero byte <=> high nibble 0xE
		const uint32_t nz = (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
		m.F |= msb_nibble(nz ^ 0x80808080u) << (4 * k);
		m.J |= msb_nibble(x & ~(x << 1) & 0x80808080u) << (4 * k);
		m.S |= msb_nibble(~x & 0x80808080u) << (4 * k);
	}
	m.F &= m.V; m.J &= m.V; m.S &= m.V;
	return m;
}

// second bytes of full deltas (`second`) and the F bytes that really open one (`fulls`), for entry state e
__device__ __forceinline__ void seg_structure(const SegMasks &m, uint32_t e, uint32_t &second, uint32_t &fulls)
{
	const uint32_t ODD = 0xAAAAAAAAu;
	const uint32_t pe = m.F & ~e;
	const uint32_t etc = (((pe << 1) | ODD) - pe) ^ ODD;
	second = (etc ^ (m.F | e)) & m.V;
	fulls = etc & m.F;
}

// what seg_walk() of the first version returned: tokens that START in the segment, for entry state e
__device__ __forceinline__ SegStats seg_stats(const uint4 &w, uint32_t nxt, const SegMasks &m, int nvalid, uint32_t e)
{
	SegStats st{e, 0, 0, 0};
	if (nvalid <= 0) return st;
	uint32_t second, fulls;
	seg_structure(m, e, second, fulls);
	const uint32_t starts = m.V & ~second;
	st.exit_state = (fulls >> (nvalid - 1)) & 1u;
	st.njump = (uint32_t)__popc(starts & m.J);
	st.npix = (uint32_t)__popc(starts & ~m.J);
	// short deltas: sum of the selected bytes, minus 128 for every one above 64 (signed(x, 7))
	const uint32_t sh = starts & m.S;
	const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
	uint32_t sum = 0, big = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t sel = (((sh >> (4 * k)) & 15u) * 0x00204081u) & 0x01010101u;  // bits 0..3 -> the low bit of bytes 0..3
		sum = __builtin_amdgcn_udot4(ws[k], sel, sum, false);
		big += (uint32_t)__popc(((((ws[k] & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) & 0x80808080u) >> 7) & sel);
	}
	st.sdelta = (int32_t)sum - 128 * (int32_t)big;
	for (uint32_t f = fulls; f; f &= f - 1) {  // full deltas are few
		const int i = __ffs((int)f) - 1;
		st.sdelta += tok_delta_full(seg_byte(w, nxt, i), seg_byte(w, nxt, i + 1));
	}
	return st;
}

// state maps over {0,1} packed as f(0) | f(1) << 1
__device__ __forceinline__ uint32_t map_apply(uint32_t m, uint32_t s) { return (m >> s) & 1u; }
__device__ __forceinline__ uint32_t map_compose(uint32_t later, uint32_t earlier)
{
	return map_apply(later, map_apply(earlier, 0)) | (map_apply(later, map_apply(earlier, 1)) << 1);
}

// cross-lane moves as DPP modifiers (VALU latency) instead of ds_bpermute (LDS latency); lanes without a source read OLD
template <int CTRL, int OLD = 0, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_from(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(OLD, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
	v += dpp_from<0x111>(v);  // row_shr:1
	v += dpp_from<0x112>(v);  // row_shr:2
	v += dpp_from<0x114>(v);  // row_shr:4
	v += dpp_from<0x118>(v);  // row_shr:8
	v += dpp_from<0x142, 0, 0xA>(v);  // row_bcast:15 into rows 1 and 3
	v += dpp_from<0x143, 0, 0xC>(v);  // row_bcast:31 into rows 2 and 3
	return v;
}

struct Parse {  // what one lane knows about its segment after the workgroup scans
	uint32_t entry;      // true entry state
	uint32_t pix_base;   // pixel ordinal of its first pixel token (within the whole stream)
	int32_t val_base;    // pixel value before its first token
	uint32_t jump_base;  // index of its first jump in the slice's jump list
	SegStats st;
};

// One parsing step over T*16 payload bytes starting at `base`.  Updates the carried stream
// state (st_in, npix, val, njump) and returns the lane's view.  Contains barriers.
__device__ __forceinline__ Parse parse_step(const uint4 &w, uint32_t nxt, int nvalid,
                                           uint32_t *scratch, uint32_t &st_carry,
                                           uint32_t &npix_carry, int32_t &val_carry,
                                           uint32_t &njump_carry)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const SegMasks sm = seg_masks(w, nvalid);
	const SegStats s0 = seg_stats(w, nxt, sm, nvalid, 0);
	const SegStats s1 = seg_stats(w, nxt, sm, nvalid, 1);
	// inclusive scan of state maps inside the wave (lanes without a source compose with the identity map, 2)
	uint32_t m = s0.exit_state | (s1.exit_state << 1);
	m = map_compose(m, dpp_from<0x111, 2>(m));
	m = map_compose(m, dpp_from<0x112, 2>(m));
	m = map_compose(m, dpp_from<0x114, 2>(m));
	m = map_compose(m, dpp_from<0x118, 2>(m));
	m = map_compose(m, dpp_from<0x142, 2, 0xA>(m));
	m = map_compose(m, dpp_from<0x143, 2, 0xC>(m));
	const uint32_t excl = dpp_from<0x138, 2>(m);  // wave_shr:1: the map of the lanes before this one
	// scratch[0..15] holds the maps, scratch[16..47] the counts: a region is rewritten only after a barrier that every
	// wave reaches after its reads of that region, so two barriers per step suffice
	if (lane == 63) scratch[wave] = m;
	__syncthreads();
	uint32_t st_wave = st_carry, st_end = st_carry;
	for (int x = 0; x < nw; x++) {
		const uint32_t wm = scratch[x];
		if (x < wave) st_wave = map_apply(wm, st_wave);
		st_end = map_apply(wm, st_end);
	}
	Parse p;
	p.entry = map_apply(excl, st_wave);
	p.st = p.entry ? s1 : s0;
	// scans of pixel count | jump count << 16 (both <= 16 per lane, <= 16384 per step) and of the delta sum
	const uint32_t ipj = wave_incl_scan_u32(p.st.npix | (p.st.njump << 16));
	const uint32_t iv = wave_incl_scan_u32((uint32_t)p.st.sdelta);
	const uint32_t ip = ipj & 0xFFFFu, ij = ipj >> 16;
	if (lane == 63) { scratch[16 + wave] = ipj; scratch[32 + wave] = iv; }
	__syncthreads();
	uint32_t bpj = 0, bv = 0, tpj = 0, tv = 0;
	for (int x = 0; x < nw; x++) {
		const uint32_t xp = scratch[16 + x], xv = scratch[32 + x];
		if (x < wave) { bpj += xp; bv += xv; }
		tpj += xp; tv += xv;
	}
	const uint32_t bp = bpj & 0xFFFFu, bj = bpj >> 16, tp = tpj & 0xFFFFu, tj = tpj >> 16;
	p.pix_base = npix_carry + bp + ip - p.st.npix;
	p.val_base = val_carry + (int32_t)(bv + iv - (uint32_t)p.st.sdelta);
	p.jump_base = njump_carry + bj + ij - p.st.njump;
	st_carry = st_end;
	npix_carry += tp;
	val_carry += (int32_t)tv;
	njump_carry += tj;
	return p;
}

// TAB_LDS: the per-block role bytes and the slot table (5 bytes per block) live in LDS instead of the HBM
// workspace: pass B looks both up for every pixel (fits up to 24 K blocks, e.g. 512 x 512 at block size 16)
// TILED: the traversal is made of aligned 64x64 tiles; the position -> raster map comes from the pattern tables of
// encode_tiles_kernel staged in LDS (<= 32 KB) instead of one 4-byte HBM/L2 load per pixel.
template <int BS, bool TAB_LDS, bool TILED = false>
__global__ void __launch_bounds__(1024) decode_kernel(DecArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
	__shared__ uint32_t scratch[64];
	__shared__ uint32_t l_jord[DEC_JLIST_CAP];
	__shared__ uint8_t l_jval[DEC_JLIST_CAP];
	__shared__ uint32_t s_status;

	const int tid = threadIdx.x, T = blockDim.x;
	const int s = blockIdx.x;
	const int N = a.N, NB = a.NB;
	const uint8_t *P = a.payload + (size_t)s * a.stride;
	const uint32_t L = a.sizes[s];
	// ByteReader.padding_len = 1 (core.py:136-142): the last byte is never returned by read()
	const uint32_t Lr = min((uint32_t)a.stride, L) > 0 ? min((uint32_t)a.stride, L) - 1u : 0u;
	const int32_t *lut = a.lut;
	uint16_t *out = a.images + (size_t)s * N;
	typedef __attribute__((address_space(3))) uint8_t lds_u8;
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	lds_u32 *l_slot = (lds_u32 *)dyn_lds;                       // NB words
	lds_u8 *l_role = (lds_u8 *)(dyn_lds + (size_t)NB * 4);      // NB bytes
	uint8_t *g_role = a.ws_role + (size_t)s * NB;
	uint32_t *g_slot = a.ws_slot + (size_t)s * NB;
	auto role_rd = [&](uint32_t b) -> uint32_t { return TAB_LDS ? (uint32_t)l_role[b] : (uint32_t)g_role[b]; };
	auto role_wr = [&](uint32_t b, uint32_t v) { if (TAB_LDS) l_role[b] = (uint8_t)v; else g_role[b] = (uint8_t)v; };
	auto slot_rd = [&](uint32_t i) -> uint32_t { return TAB_LDS ? l_slot[i] : g_slot[i]; };
	auto slot_wr = [&](uint32_t i, uint32_t v) { if (TAB_LDS) l_slot[i] = v; else g_slot[i] = v; };
	typedef __attribute__((address_space(3))) uint16_t lds_u16;
	// position p of a tile at p + p / 16: consecutive lanes look up positions 16 apart (one block each), which would be
	// 32 bytes = the same four banks for the whole wave; the pad entry per block spreads them over all banks
	constexpr uint32_t PAT_STRIDE = 4096 + 256;
	lds_u16 *l_pat = (lds_u16 *)(dyn_lds + (((size_t)NB * 5 + 15) & ~(size_t)15));   // n_orient * PAT_STRIDE
	lds_u32 *l_torg = (lds_u32 *)(l_pat + (TILED ? (size_t)a.n_orient * PAT_STRIDE : 0)); // n_tiles
	lds_u8 *l_tori = (lds_u8 *)(l_torg + (TILED ? a.n_tiles : 0));                   // n_tiles
	if (TILED) {
		// the encoder's pattern entries are swizzled LDS byte offsets (dy*128 + ((dx>>3 ^ dy&7) << 4) + (dx&7)*2); pass B wants
		// the raster offset inside the tile, dy * width + dx (< 65536: launch_decode takes this path for widths <= 1024 only)
		for (int i = tid; i < a.n_orient * 4096; i += T) {
			const uint32_t pv = a.patterns[i];
			const uint32_t dy = pv >> 7, dx = ((((pv >> 4) & 7u) ^ (dy & 7u)) << 3) | ((pv & 15u) >> 1);
			const uint32_t o = (uint32_t)i >> 12, p = (uint32_t)i & 4095u;
			l_pat[o * PAT_STRIDE + p + (p >> 4)] = (uint16_t)(dy * (uint32_t)a.width + dx);
		}
		for (int i = tid; i < a.n_tiles; i += T) { l_torg[i] = a.tile_org[i]; l_tori[i] = a.tile_orient[i]; }
	}
	// raster offset of traversal position pos
	auto raster_of = [&](uint32_t pos) -> uint32_t {
		if (TILED) {
			const uint32_t tile = pos >> 12;
			const uint32_t p = pos & 4095u;
			return l_torg[tile] + l_pat[(uint32_t)l_tori[tile] * PAT_STRIDE + p + (p >> 4)];
		}
		return lut ? (uint32_t)lut[pos] : pos;
	};
	const size_t jcap = (size_t)NB / 2 + 1;
	uint32_t *g_jord = a.ws_jord + (size_t)s * jcap;
	uint8_t *g_jval = a.ws_jval + (size_t)s * jcap;

#ifdef CCT_DEC_PROF  // tuning builds only: phase times of workgroup 0
	long long tp[8]; int tpi = 0;
#define DEC_STAMP() do { if (tpi < 8) tp[tpi++] = clock64(); } while (0)
#else
#define DEC_STAMP() do {} while (0)
#endif
	DEC_STAMP();
	if (tid == 0) s_status = 0;
	for (int b = tid; b < NB; b += T) role_wr((uint32_t)b, 0);
	__syncthreads();
	DEC_STAMP();

	// No branch around the loads (the address is clamped, the result selected): a load inside a conditional is waited for on the
	// spot, which defeats requesting the next step's segment ahead of time.  stride is a multiple of 256 and Lr <= stride, so a
	// segment that starts inside the payload is read from its own address.
	const uint32_t seg_max = (uint32_t)a.stride - 16u;
	auto load_seg = [&](uint32_t seg_start, uint4 &w, uint32_t &nxt) {  // the bytes as loaded: nothing is derived from them here,
		w = *reinterpret_cast<const uint4 *>(P + min(seg_start, seg_max));  // or the wait lands where the load is issued
		nxt = P[min(seg_start + 16u, (uint32_t)a.stride - 1u)];
	};
	auto clip_seg = [&](uint32_t seg_start, uint4 &w, uint32_t &nxt, int &nvalid) {  // what lies past the payload reads as zero
		const bool inside = seg_start < Lr;
		nvalid = inside ? (int)min(16u, Lr - seg_start) : 0;
		if (!inside) w = make_uint4(0, 0, 0, 0);
		if (!(seg_start + 16u < Lr)) nxt = 0;
	};
	auto jset = [&](uint32_t k, uint32_t ord, uint32_t j) {
		if (k < DEC_JLIST_CAP) { l_jord[k] = ord; l_jval[k] = (uint8_t)j; }
		else if (k - DEC_JLIST_CAP < jcap) { g_jord[k - DEC_JLIST_CAP] = ord; g_jval[k - DEC_JLIST_CAP] = (uint8_t)j; }
	};

	// ------------------------------------------------------------------ pass A: jump tokens
	uint32_t st_c = 0, npix_c = 0, nj_c = 0;
	int32_t val_c = 0;
	uint2 *pcache = a.ws_pcache + (size_t)s * a.pcache_steps * T;
	uint32_t nsteps = 0;
	uint4 wa_n; uint32_t nxta_n;
	load_seg((uint32_t)tid * DEC_SEG, wa_n, nxta_n);
	for (uint32_t base = 0; base < Lr && npix_c < (uint32_t)N; base += (uint32_t)T * DEC_SEG, nsteps++) {
		uint4 w = wa_n; uint32_t nxt = nxta_n; int nvalid;
		clip_seg(base + (uint32_t)tid * DEC_SEG, w, nxt, nvalid);
		load_seg(base + (uint32_t)(T + tid) * DEC_SEG, wa_n, nxta_n);  // the next step's segment, in flight across the scans below
		const Parse p = parse_step(w, nxt, nvalid, scratch, st_c, npix_c, val_c, nj_c);
		// taken before this step's stores: one in-order counter covers loads and stores, so a wait for the segment placed after
		// them would wait for them too
		asm volatile("" : "+v"(wa_n.x), "+v"(wa_n.y), "+v"(wa_n.z), "+v"(wa_n.w), "+v"(nxta_n) :: "memory");
		pcache[(size_t)nsteps * T + tid] = make_uint2(min(p.pix_base, 0x7FFFFFFFu) | (p.entry << 31), (uint32_t)p.val_base);
		if (p.st.njump) {
			const SegMasks sm = seg_masks(w, nvalid);
			uint32_t second, fulls;
			seg_structure(sm, p.entry, second, fulls);
			const uint32_t starts = sm.V & ~second, pix = starts & ~sm.J;
			uint32_t k = p.jump_base;
			for (uint32_t jm = starts & sm.J; jm; jm &= jm - 1, k++) {
				const int i = __ffs((int)jm) - 1;
				const uint32_t ord = p.pix_base + (uint32_t)__popc(pix & ((1u << i) - 1u));  // pixel tokens before the jump
				if (ord < (uint32_t)N) jset(k, ord, seg_byte(w, nxt, i) & 0x3Fu);
			}
		}
	}
	__syncthreads();
	DEC_STAMP();
	if (npix_c < (uint32_t)N && tid == 0) atomicOr(&s_status, CCT_ST_STREAM);  // ran out of tokens

	// ------------------------------------------------------------------ resolve jumps -> role[]
	if (tid < 64) {
		// The replay is serial; wave 0 runs it with every loaded value made wave-uniform (v_readfirstlane), so that the state
		// lives in SGPRs, the 64-bit window is shifted by scalar instructions and the loops branch without exec-mask juggling.
		// Lane 0 does the stores.  The next jump is fetched while the current one is replayed.
		auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
		const uint32_t nj = uni(nj_c);
		uint32_t F = 0;          // frontier: next block in traversal order not yet emitted as a leader
		uint64_t win = 0;        // bit t: block F+t already claimed as a partner
		uint32_t slots_done = 0; // 16-pixel stream slots consumed before the frontier
		bool bad = false;
		auto fetch = [&](uint32_t k, uint32_t &ord, uint32_t &j) {
			ord = 0xFFFFFFFFu; j = 0;
			if (k < DEC_JLIST_CAP) { ord = l_jord[k]; j = l_jval[k]; }
			else if (k - DEC_JLIST_CAP < jcap) { ord = g_jord[k - DEC_JLIST_CAP]; j = g_jval[k - DEC_JLIST_CAP]; }
		};
		uint32_t ord_n = 0, j_n = 0;
		if (nj) fetch(0, ord_n, j_n);
		for (uint32_t k = 0; k < nj && !bad; k++) {
			if (k >= DEC_JLIST_CAP && k - DEC_JLIST_CAP >= jcap) { bad = true; break; }
			const uint32_t ord = uni(ord_n), j = uni(j_n);
			if (k + 1 < nj) fetch(k + 1, ord_n, j_n);
			if (ord >= (uint32_t)N) break;  // tokens past the last pixel are never read
			if (ord % BS != 0) { bad = true; break; }
			const uint32_t sslot = ord / BS;
			if (sslot < slots_done) { bad = true; break; }
			uint32_t remaining = sslot - slots_done;  // single blocks between the two pairs
			while (remaining) {
				if (win == 0) { F += remaining; remaining = 0; break; }
				const uint32_t o = (uint32_t)__ffsll((long long)win) - 1u;  // singles before next partner
				if (o >= remaining) { F += remaining; win >>= remaining; remaining = 0; }
				else { F += o + 1; win = (o + 1 >= 64u) ? 0ull : (win >> (o + 1)); remaining -= o; }
			}
			{ const uint64_t free_ = ~win; const uint32_t t = free_ ? (uint32_t)__builtin_ctzll(free_) : 64u; F += t; win = t >= 64u ? 0ull : win >> t; }  // partners at the frontier
			const uint32_t Lb = F, pb = F + j;
			if (j == 0 || pb >= (uint32_t)NB || ((win >> j) & 1ull)) { bad = true; break; }
			if (tid == 0) { role_wr(Lb, j); role_wr(pb, ROLE_PARTNER); }
			win |= 1ull << j;
			F = Lb + 1;
			win >>= 1;
			slots_done = sslot + 2;
		}
		if (bad && tid == 0) atomicOr(&s_status, CCT_ST_STREAM);
	}
	__syncthreads();
	DEC_STAMP();

	// ------------------------------------------------------------------ slot table
	{
		uint32_t slot_c = 0;
		for (int base = 0; base < NB; base += T) {
			const int b = base + tid;
			uint32_t r = ROLE_PARTNER, cnt = 0;
			if (b < NB) { r = role_rd((uint32_t)b); cnt = (r == ROLE_PARTNER) ? 0u : (r ? 2u : 1u); }
			const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
			const uint32_t inc = wave_incl_scan_u32(cnt);
			if (lane == 63) scratch[wave] = inc;
			__syncthreads();
			uint32_t bs_ = 0, tot = 0;
			for (int x = 0; x < nw; x++) { const uint32_t v = scratch[x]; if (x < wave) bs_ += v; tot += v; }
			__syncthreads();
			const uint32_t sl = slot_c + bs_ + inc - cnt;
			if (cnt >= 1 && sl < (uint32_t)NB) slot_wr(sl, (uint32_t)b | (cnt == 2 ? (1u << 30) : 0u));
			if (cnt == 2 && sl + 1 < (uint32_t)NB) slot_wr(sl + 1, (uint32_t)b | (2u << 30));
			slot_c += tot;
		}
		if (slot_c != (uint32_t)NB && tid == 0) atomicOr(&s_status, CCT_ST_STREAM);
	}
	__syncthreads();
	DEC_STAMP();

	// ------------------------------------------------------------------ pass B: pixels
	if (!(s_status & CCT_ST_STREAM)) {
		// every lane re-reads its segments and what pass A's scans told it about them: no barrier in this pass; the loads
		// of the next step are issued before this step's pixels are written
		uint4 w_n; uint32_t nxt_n;
		load_seg((uint32_t)tid * DEC_SEG, w_n, nxt_n);
		uint2 pc_n = pcache[tid];  // (slot 0 of the step cache exists even for an empty payload)
#ifdef CCT_DEC_PROF
		long long tb[4] = {0, 0, 0, 0}, tq = clock64();
#define DECB(j) do { const long long t_ = clock64(); tb[j] += t_ - tq; tq = t_; } while (0)
#else
#define DECB(j) do {} while (0)
#endif
		for (uint32_t k = 0; k < nsteps; k++) {
			const uint32_t seg_start = k * (uint32_t)T * DEC_SEG + (uint32_t)tid * DEC_SEG;
			uint4 w = w_n; uint32_t nxt = nxt_n; int nvalid;
			clip_seg(seg_start, w, nxt, nvalid);
			const uint2 pc = pc_n;
			load_seg(seg_start + (uint32_t)T * DEC_SEG, w_n, nxt_n);  // next step (clamped addresses: no branch around the loads)
			pc_n = pcache[(size_t)min(k + 1, nsteps - 1) * T + tid];
			Parse p;
			p.entry = pc.x >> 31; p.pix_base = pc.x & 0x7FFFFFFFu; p.val_base = (int32_t)pc.y;
			// token structure of the segment as masks (see seg_masks): the loop below visits pixel tokens only
			const SegMasks sm = seg_masks(w, nvalid);
			uint32_t second, fulls;
			seg_structure(sm, p.entry, second, fulls);
			uint32_t starts = sm.V & ~second, pix = starts & ~sm.J;
			uint32_t ord = p.pix_base;
			int32_t val = p.val_base;
			uint32_t flags = 0;
			// the reference stops reading after pixel N - 1: tokens behind it do not exist for the decoder
			const uint32_t room = ord < (uint32_t)N ? (uint32_t)N - ord : 0u;
			if ((uint32_t)__popc(pix) > room) {
				uint32_t keep = 0;
				if (room) {
					uint32_t pm = pix;
					for (uint32_t r = 1; r < room; r++) pm &= pm - 1;
					keep = (2u << (__ffs((int)pm) - 1)) - 1u;  // up to and including the last wanted pixel token
				}
				starts &= keep; pix &= keep; fulls &= keep;
			}
			DECB(0);
			const uint32_t jm = starts & sm.J;
			if (jm & (jm << 1)) flags |= CCT_ST_STREAM;  // two jump bytes in a row
			if (nvalid > 0 && ((fulls >> (nvalid - 1)) & 1u) && seg_start + (uint32_t)nvalid >= Lr) flags |= CCT_ST_STREAM;  // second byte missing
			if constexpr (BS >= 16) {
				// A lane's <= 16 pixels fall into at most two stream slots: both slots' entry, partner and tile data are fetched
				// before the loop.  The loop itself runs over the 16 byte positions with a compile-time index (a byte is a bit-field
				// extract, not a select chain) and skips the positions that open no pixel token.
				// per slot and parity (leader / partner of a meshed pair): first traversal position of the block, the tile's raster
				// origin and the index of the block's first pattern entry (the table is padded by one entry per 16 positions) -- a
				// pixel then costs one select of each instead of rebuilding them from block numbers
				struct SlotInfo { uint32_t kind, pos0, pos1, org0, org1, idx0, idx1; };
				auto slot_info = [&](uint32_t sl) {
					SlotInfo si{0, 0, 0, 0, 0, 0, 0};
					const uint32_t ent = slot_rd(min(sl, (uint32_t)NB - 1u));
					const uint32_t blk0 = ent & 0x3FFFFFFFu; si.kind = ent >> 30;
					const uint32_t blk1 = si.kind ? blk0 + role_rd(blk0) : blk0;
					si.pos0 = blk0 * BS; si.pos1 = blk1 * BS;
					if (TILED) {
						const uint32_t t0 = si.pos0 >> 12, t1 = si.pos1 >> 12, b0 = si.pos0 & 4095u, b1 = si.pos1 & 4095u;
						si.org0 = l_torg[t0]; si.idx0 = (uint32_t)l_tori[t0] * PAT_STRIDE + b0 + (b0 >> 4);
						si.org1 = l_torg[t1]; si.idx1 = (uint32_t)l_tori[t1] * PAT_STRIDE + b1 + (b1 >> 4);
					}
					return si;
				};
				const uint32_t slA = ord / BS;
				const SlotInfo A = slot_info(slA), B = slot_info(slA + 1u);
				DECB(1);
				const uint32_t ws[5] = {w.x, w.y, w.z, w.w, nxt};
#pragma unroll
				for (int i = 0; i < 16; i++) {
					if (!((pix >> i) & 1u)) continue;
					const uint32_t c = (ws[i >> 2] >> (8 * (i & 3))) & 0xFFu;
					const uint32_t c1 = (ws[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFFu;
					// (reserved tags 110xxxxx / 1111xxxx: no branch of core.py:500-516 is taken, the previous pixel repeats)
					val += ((sm.S >> i) & 1u) ? tok_delta_short(c) : ((fulls >> i) & 1u) ? tok_delta_full(c, c1) : 0;
					if ((uint32_t)val > 65535u) flags |= CCT_ST_OVERFLOW;  // to_bytes(2), core.py:506 (negative or too large)
					const uint32_t sl = ord / BS, t = ord % BS;
					const bool inA = sl == slA;
					const uint32_t kind = inA ? A.kind : B.kind;
					uint32_t odd = 0, off = t;
					if (kind) { const uint32_t mm = (kind - 1u) * BS + t; odd = mm & 1u; off = mm >> 1; }  // index inside the 2*bs interleave
					uint32_t ras;
					if (TILED) {
						const uint32_t org = inA ? (odd ? A.org1 : A.org0) : (odd ? B.org1 : B.org0);
						const uint32_t idx = inA ? (odd ? A.idx1 : A.idx0) : (odd ? B.idx1 : B.idx0);
						ras = org + l_pat[idx + off + (BS > 16 ? off >> 4 : 0u)];  // (off < BS: inside the block)
					} else ras = raster_of((inA ? (odd ? A.pos1 : A.pos0) : (odd ? B.pos1 : B.pos0)) + off);
					out[ras] = (uint16_t)val;
					ord++;
				}
			} else {
				// a lane's pixels fall into at most two stream slots: slot entry, partner and tile data are fetched once per slot
				uint32_t cur_sl = 0xFFFFFFFFu, kind = 0, blk0 = 0, blk1 = 0;
				uint32_t org0 = 0, org1 = 0, pat0 = 0, pat1 = 0;
				for (uint32_t pm = pix; pm; pm &= pm - 1) {
					const int i = __ffs((int)pm) - 1;
					const uint32_t c = seg_byte(w, nxt, i);
					if ((sm.S >> i) & 1u) val += tok_delta_short(c);
					else if ((fulls >> i) & 1u) val += tok_delta_full(c, seg_byte(w, nxt, i + 1));
					// (reserved tags 110xxxxx / 1111xxxx: no branch of core.py:500-516 is taken, the previous pixel repeats)
					if (val < 0 || val > 65535) flags |= CCT_ST_OVERFLOW;  // to_bytes(2), core.py:506
					const uint32_t sl = ord / BS, t = ord % BS;
					if (sl != cur_sl) {
						cur_sl = sl;
						const uint32_t ent = slot_rd(sl);
						blk0 = ent & 0x3FFFFFFFu; kind = ent >> 30;
						blk1 = kind ? blk0 + role_rd(blk0) : blk0;
						if (TILED) {
							const uint32_t t0 = (blk0 * BS) >> 12, t1 = (blk1 * BS) >> 12;
							org0 = l_torg[t0]; pat0 = (uint32_t)l_tori[t0] * PAT_STRIDE;
							org1 = l_torg[t1]; pat1 = (uint32_t)l_tori[t1] * PAT_STRIDE;
						}
					}
					uint32_t pos, odd = 0;
					if (kind == 0) pos = blk0 * BS + t;
					else {
						const uint32_t mm = (kind - 1u) * BS + t;  // index inside the 2*bs interleave
						odd = mm & 1u;
						pos = (odd ? blk1 : blk0) * BS + (mm >> 1);
					}
					uint32_t ras;
					if (TILED) {
						const uint32_t p = pos & 4095u;
						ras = (odd ? org1 : org0) + l_pat[(odd ? pat1 : pat0) + p + (p >> 4)];
					} else ras = raster_of(pos);
					out[ras] = (uint16_t)val;
					ord++;
				}
			}
			DECB(2);
			if (flags) atomicOr(&s_status, flags);
		}
#ifdef CCT_DEC_PROF
		if (s == 0 && tid == 0) printf("[decode prof] pass B wave 0: masks %lld slots %lld pixels %lld\n", tb[0], tb[1], tb[2]);
#endif
	}
	__syncthreads();
	DEC_STAMP();
#ifdef CCT_DEC_PROF
	if (s == 0 && tid == 0) {
		printf("[decode prof] steps %u jumps %u:", nsteps, nj_c);
		for (int i = 1; i < tpi; i++) printf(" %lld", tp[i] - tp[i - 1]);
		printf("  (init+tables | pass A | resolve | slot table | pass B)\n");
	}
#endif
	if (tid == 0) a.status[s] = s_status;
}

}  // namespace

hipError_t launch_decode(const DecArgs &a, int n, int block_size, int threads, hipStream_t s)
{
	const size_t tab_bytes = (((size_t)a.NB * 5 + 15) & ~(size_t)15) + 16;
	const bool tab_lds = tab_bytes <= 100 * 1024;
	const size_t tile_bytes = (size_t)a.n_orient * (4096 + 256) * 2 + (size_t)a.n_tiles * 5 + 16;
	const bool tiled = tab_lds && block_size == 16 && a.lut && a.n_tiles > 0 && tab_bytes + tile_bytes <= 144 * 1024 && a.width <= 1024;
	void (*k)(DecArgs) = nullptr;
	switch (block_size) {
	case 4: k = tab_lds ? decode_kernel<4, true> : decode_kernel<4, false>; break;
	case 8: k = tab_lds ? decode_kernel<8, true> : decode_kernel<8, false>; break;
	case 16: k = tiled ? decode_kernel<16, true, true> : tab_lds ? decode_kernel<16, true> : decode_kernel<16, false>; break;
	case 32: k = tab_lds ? decode_kernel<32, true> : decode_kernel<32, false>; break;
	case 64: k = tab_lds ? decode_kernel<64, true> : decode_kernel<64, false>; break;
	default: return hipErrorInvalidValue;
	}
	const size_t lds = tab_lds ? tab_bytes + (tiled ? tile_bytes : 0) : 0;
	if (lds) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(k, dim3(n), dim3(threads), lds, s, a);
	return hipGetLastError();
}

}  // namespace cct
