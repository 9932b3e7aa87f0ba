// CompaCT encode, stage (i): traversal -> segmentation/mesh -> delta -> tag-byte pack.
//
// One workgroup owns one slice and streams it twice through an LDS ring of traversal-ordered
// pixel chunks; nothing is exchanged between workgroups, so a batch is ONE launch and slices
// shard across CUs / GPUs with no synchronisation.
//
//   pass 1  (segmentation only)  per chunk: gather pixels into the ring, one lane per block counts
//           the "large delta" transitions (cluster.py:30-59), difficult blocks are appended in
//           order to a list (wave scan), and for each difficult block of the previous chunk one
//           WAVE evaluates its 63 look-ahead candidates, one candidate per lane, and ballots the
//           fit mask (cluster.py:122-158).
//   resolve the greedy first-fit (cluster.py:79-190) only couples difficult blocks <= 63 apart,
//           so each lane walks one such island with a 64-bit "completed" window: role[b] becomes
//           0 (single), 1..63 (pair leader, jump distance) or 0xFF (consumed partner).
//   pass 2  per chunk: one lane per block sizes its tokens (core.py:313-323), a workgroup scan turns
//           sizes into byte offsets, tokens are written to an LDS staging buffer and flushed to
//           HBM as aligned 16-byte units; the jump byte (core.py:290-294) leads a pair's tokens.
//
// Algorithmic HBM traffic per pixel: 2 B read (pass 1) + payload bytes written (~1.05 B on CT);
// pass 2 re-reads the slice (L2 / Infinity Cache resident: it was streamed microseconds earlier).
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

constexpr int RING_PX = ENC_RING * ENC_CH;  // 32768 pixels, 64 KiB

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = __shfl_up(v, d);
		if (lane >= d) v += t;
	}
	return v;
}

// Exclusive scan of v over the workgroup (threads in tid order); total returned in `total`.
// Contains two barriers; `scratch` needs blockDim.x/64 words.
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t v, uint32_t *scratch, uint32_t &total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint32_t inc = wave_incl_scan(v);
	if (lane == 63) scratch[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (int w = 0; w < nw; w++) {
		const uint32_t x = scratch[w];
		if (w < wave) base += x;
		tot += x;
	}
	__syncthreads();
	total = tot;
	return base + inc - v;
}

// difficult-block list: first ENC_LIST_CAP records in LDS, the rest in the HBM workspace
struct DiffList {
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

template <int BS>
__global__ void __launch_bounds__(1024) encode_kernel(EncArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint16_t *ring = reinterpret_cast<uint16_t *>(smem);
	uint8_t *stg = smem + RING_PX * 2;
	uint64_t *l_mask = reinterpret_cast<uint64_t *>(stg + ENC_STG_BYTES);
	uint32_t *l_idx = reinterpret_cast<uint32_t *>(l_mask + ENC_LIST_CAP);
	uint32_t *scratch = l_idx + ENC_LIST_CAP;  // 64 words: [0..31] scan, [32] status bits
	uint8_t *l_cur = reinterpret_cast<uint8_t *>(scratch + 64);
	uint8_t *role_lds = l_cur + ENC_LIST_CAP;

	const int tid = threadIdx.x, T = blockDim.x;
	const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
	const int s = blockIdx.x;
	const int N = a.N, NB = a.NB;
	const bool seg = (a.flags & CCT_FLAG_SEGMENTATION) != 0;
	const bool sgn = (a.flags & CCT_FLAG_SIGNED_SEG) != 0;
	const uint16_t *img = a.images + (size_t)s * N;
	const int32_t *lut = a.lut;
	uint8_t *role = a.ws_role ? a.ws_role + (size_t)s * NB : role_lds;
	DiffList dl{l_idx, l_cur, l_mask,
	            a.ws_lidx + (size_t)s * NB, a.ws_lcur + (size_t)s * NB, a.ws_lmask + (size_t)s * NB};

	constexpr int CB = ENC_CH / BS;  // blocks per chunk
	const int nch = (N + ENC_CH - 1) / ENC_CH;

	if (tid == 0) { scratch[32] = 0; scratch[33] = 0; scratch[34] = 0; }

	auto fill = [&](int c) {
		const int k0 = c * ENC_CH;
		const int npx = min(ENC_CH, N - k0);
		uint16_t *dst = ring + (k0 & (RING_PX - 1));
		if (lut) {
			for (int k = tid; k < npx; k += T) dst[k] = img[lut[k0 + k]];
		} else {
			for (int k = tid; k < npx; k += T) dst[k] = img[k0 + k];
		}
	};
	auto D = [&](int k) -> int { return (int)ring[k & (RING_PX - 1)]; };
	auto Dseg = [&](int k) -> int {  // value as segmentation sees it (core.py:254)
		const uint16_t v = ring[k & (RING_PX - 1)];
		return sgn ? (int)(int16_t)v : (int)v;
	};

	// ------------------------------------------------------------------ pass 1
	uint32_t ndiff = 0;
	if (seg) {
		uint32_t start_prev = 0, start_cur = 0;
		for (int c = 0; c <= nch; c++) {
			if (c < nch) fill(c);
			__syncthreads();
			start_prev = start_cur;
			start_cur = ndiff;
			if (c < nch) {
				const int nblk = min(CB, NB - c * CB);
				for (int base = 0; base < nblk; base += T) {
					const int bl = base + tid;
					const bool act = bl < nblk;
					const int b = c * CB + bl;
					uint32_t diff = 0, cur = 0;
					if (act) {
						const int k0 = b * BS;
						int prev = Dseg(k0);
						uint32_t chg = 0;
#pragma unroll
						for (int t = 1; t < BS; t++) {
							const int v = Dseg(k0 + t);
							const int d = v - prev;
							chg += (d > 64 || d < -64) ? 1u : 0u;  // |d| > 64, cluster.py:38-39
							prev = v;
						}
						if (2 * chg >= (uint32_t)BS) {  // cluster.py:58
							diff = 1;
							uint32_t enter = 0;
							if (k0 > 0) {
								const int d = Dseg(k0) - Dseg(k0 - 1);
								enter = (d > 64 || d < -64) ? 1u : 0u;
							}
							cur = chg + enter;  // cluster.py:110 for i > 0 (i == 0: Q4, handled below)
						}
						role[b] = 0;
					}
					uint32_t tot;
					const uint32_t pos = wg_excl_scan(diff, scratch, tot);
					if (diff) dl.set(ndiff + pos, (uint32_t)b, cur);
					ndiff += tot;
				}
			}
			__syncthreads();  // list records of chunk c visible
			// candidate masks for the difficult blocks of chunk c-1 (their look-ahead lies in c-1, c)
			for (uint32_t e = start_prev + wave; e < start_cur; e += nwaves) {
				const int i = (int)dl.idx(e);
				const uint32_t cur = dl.cur(e);
				const int p = i + lane;
				bool fit = false;
				if (lane >= 1 && p < NB) {
					const int ka = i * BS, kb = p * BS;
					uint32_t up = 0;
					int bprev = 0;
#pragma unroll
					for (int t = 0; t < BS; t++) {
						const int av = Dseg(ka + t), bv = Dseg(kb + t);
						if (t > 0) up += (av - bprev >= 65) ? 1u : 0u;  // A[t] - B[t-1]
						up += (bv - av >= 65) ? 1u : 0u;               // B[t] - A[t]
						bprev = bv;
					}
					// cluster.py:153,158: num_changes = up + 1 < current_delta - 2 in uint32
					// arithmetic; for block 0 current_delta wraps (Q4) and every candidate fits.
					fit = (i == 0) ? true : ((up + 1u) < (cur - 2u));
				}
				const uint64_t m = __ballot(fit);
				if (lane == 0) dl.set_mask(e, m);
			}
		}
		__syncthreads();

		// -------------------------------------------------------------- resolve
		for (uint32_t e0 = tid; e0 < ndiff; e0 += T) {
			const uint32_t i0 = dl.idx(e0);
			if (e0 > 0 && i0 - dl.idx(e0 - 1) <= 63u) continue;  // not the head of an island
			uint64_t cw = 0;  // bit t: block (base + t) already completed
			uint32_t base = i0;
			uint32_t e = e0, i = i0;
			for (;;) {
				const uint32_t sh = i - base;
				cw = (sh >= 64u) ? 0ull : (cw >> sh);
				base = i;
				if (!(cw & 1ull)) {
					const uint64_t avail = dl.mask(e) & ~cw & ~1ull;
					if (avail) {
						const int j = __ffsll((long long)avail) - 1;  // first fit, cluster.py:181
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

	// ------------------------------------------------------------------ pass 2
	uint8_t *out = a.payload + (size_t)s * a.stride;
	uint32_t out_pos = 0;  // bytes already flushed (multiple of 16)
	uint32_t carry = 0;    // bytes waiting at stg[0..carry)
	bool cap_hit = false;
	uint32_t my_full = 0, my_jump = 0;  // token statistics (Encoder.info, core.py:317,322)

	for (int c = 0; c <= nch; c++) {
		if (c < nch) fill(c);
		__syncthreads();
		if (c == 0) continue;
		const int e = c - 1;
		const int nblk = min(CB, NB - e * CB);
		uint32_t chunk_bytes = 0;
		for (int base = 0; base < nblk; base += T) {
			const int bl = base + tid;
			const bool act = bl < nblk;
			const int b = e * CB + bl;
			uint32_t nbytes = 0;
			int r = 0, pv = 0;
			bool q7 = false;
			if (act) {
				r = seg ? (int)role[b] : 0;
				if (a.roles_out) a.roles_out[(size_t)s * NB + b] = (uint8_t)r;
				if (r != ROLE_PARTNER) {
					if (b > 0) {  // last pixel of the previous emitted group
						int q = b - 1;
						int rq = seg ? (int)role[q] : 0;
						while (rq == ROLE_PARTNER) { q--; rq = (int)role[q]; }
						const int src = q + rq;  // rq = 0 for a single
						pv = D(src * BS + BS - 1);
					}
					const int ka = b * BS;
					int prev = pv;
					uint32_t n2 = 0;
					if (r == 0) {
#pragma unroll
						for (int t = 0; t < BS; t++) {
							const int v = D(ka + t);
							const int d = v - prev;
							const bool two = (d < -63 || d > 64);  // core.py:316
							n2 += two ? 1u : 0u;
							q7 |= (d < -2047 || d > 2048);
							prev = v;
						}
						nbytes = BS + n2;
						my_full += n2;
					} else {
						const int kb = (b + r) * BS;
#pragma unroll
						for (int t = 0; t < BS; t++) {
							const int va = D(ka + t), vb = D(kb + t);
							const int d1 = va - prev, d2 = vb - va;
							n2 += (d1 < -63 || d1 > 64) ? 1u : 0u;
							n2 += (d2 < -63 || d2 > 64) ? 1u : 0u;
							q7 |= (d1 < -2047 || d1 > 2048) || (d2 < -2047 || d2 > 2048);
							prev = vb;
						}
						nbytes = 2 * BS + n2 + 1;
						my_full += n2;
						my_jump += 1;
					}
				}
			}
			if (q7) atomicOr(&scratch[32], CCT_ST_Q7);
			uint32_t tot;
			const uint32_t off = wg_excl_scan(nbytes, scratch, tot);
			if (nbytes) {
				uint8_t *w = stg + carry + chunk_bytes + off;
				const int ka = b * BS;
				int prev = pv;
				auto put = [&](int d) {
					if (d < -63 || d > 64) {  // full delta, core.py:322-323
						*w++ = (uint8_t)(0xE0 | ((d >> 8) & 0x0F));
						*w++ = (uint8_t)(d & 0xFF);
					} else {                  // short delta, core.py:316-319
						*w++ = (uint8_t)(d & 0x7F);
					}
				};
				if (r == 0) {
#pragma unroll
					for (int t = 0; t < BS; t++) {
						const int v = D(ka + t);
						put(v - prev);
						prev = v;
					}
				} else {
					*w++ = (uint8_t)(0x80 | r);  // jump tag, core.py:290-294
					const int kb = (b + r) * BS;
#pragma unroll
					for (int t = 0; t < BS; t++) {
						const int va = D(ka + t), vb = D(kb + t);
						put(va - prev);
						put(vb - va);
						prev = vb;
					}
				}
			}
			chunk_bytes += tot;
		}
		// ---- flush staged bytes as aligned 16-byte units
		const bool last = (c == nch);
		if (last && a.stats) {
			if (my_full) atomicAdd(&scratch[33], my_full);
			if (my_jump) atomicAdd(&scratch[34], my_jump);
		}
		uint32_t total = carry + chunk_bytes;
		if (last && a.eof >= 0) {  // core.py:329-330
			if (tid == 0) stg[total] = (uint8_t)a.eof;
			total += 1;
		}
		const uint32_t nflush = last ? ((total + 15u) & ~15u) : (total & ~15u);
		if (last && tid < (int)(nflush - total)) stg[total + tid] = 0;
		__syncthreads();
		if ((size_t)out_pos + nflush > a.stride) cap_hit = true;
		if (!cap_hit) {
			const uint4 *src = reinterpret_cast<const uint4 *>(stg);
			uint4 *dst = reinterpret_cast<uint4 *>(out + out_pos);
			for (uint32_t u = tid; u < nflush / 16u; u += T) dst[u] = src[u];
		}
		const uint32_t rem = last ? 0u : (total - nflush);
		uint8_t keep = 0;
		if ((uint32_t)tid < rem) keep = stg[nflush + tid];
		__syncthreads();
		if ((uint32_t)tid < rem) stg[tid] = keep;
		out_pos += nflush;
		carry = rem;
		if (last && tid == 0) {
			a.sizes[s] = cap_hit ? 0u : (out_pos - nflush + total);
			a.status[s] = scratch[32] | (cap_hit ? CCT_ST_CAP : 0u);
			if (a.stats) {
				uint32_t *st = a.stats + (size_t)s * 4;
				st[0] = (uint32_t)N - scratch[33];  // short tokens
				st[1] = scratch[33];                // full tokens
				st[2] = scratch[34];                // jump tokens
				st[3] = ndiff;                      // difficult blocks
			}
		}
		__syncthreads();
	}
}

}  // namespace

size_t enc_lds_bytes(int NB, bool *role_in_lds)
{
	size_t base = (size_t)RING_PX * 2 + ENC_STG_BYTES + (size_t)ENC_LIST_CAP * (8 + 4 + 1) + 64 * 4;
	const bool in_lds = NB <= ENC_MAX_LDS_ROLE;
	if (role_in_lds) *role_in_lds = in_lds;
	if (in_lds) base += ((size_t)NB + 15) & ~(size_t)15;
	return base;
}

hipError_t launch_encode(const EncArgs &a, int n, int block_size, int threads, hipStream_t s)
{
	bool in_lds;
	const size_t lds = enc_lds_bytes(a.NB, &in_lds);
	void (*k)(EncArgs) = nullptr;
	switch (block_size) {
	case 4: k = encode_kernel<4>; break;
	case 8: k = encode_kernel<8>; break;
	case 16: k = encode_kernel<16>; break;
	case 32: k = encode_kernel<32>; break;
	case 64: k = encode_kernel<64>; break;
	default: return hipErrorInvalidValue;
	}
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k, dim3(n), dim3(threads), lds, s, a);
	return hipGetLastError();
}

}  // namespace cct
