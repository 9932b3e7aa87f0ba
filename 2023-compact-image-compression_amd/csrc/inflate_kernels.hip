// Device INFLATE (RFC 1950/1951) for the decode side: zlib.decompress(file_bytes[13:]) of the reference
// (src/codec/core.py:421).  zlib is a third-party dependency of the reference; inflating is fully
// specified by the stream format, so any correct decoder returns the same bytes (and the Adler-32
// trailer is verified like zlib does).
//
// One wave per stream.  The Huffman decode itself is a serial bit-by-bit dependency chain (the
// format carries no block index), so the wave spends its lanes where the format allows it:
//   * canonical code tables of each dynamic block are built by all 64 lanes (fast LSB-first lookup
//     tables: 11 bits literal/length, 10 bits distance; longer codes fall back to canonical decode);
//   * every LZ77 copy (up to 258 bytes) is done by the whole wave, overlapping copies included;
//   * compressed input is staged through LDS in 4 KiB chunks, output through a 64 KiB LDS ring that
//     is flushed to HBM as aligned 16-byte stores and Adler-summed on the way out.
// Slices are independent, so 256 streams occupy 256 CUs at once.
#include <hip/hip_runtime.h>

#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

constexpr int INF_RING = 65536, INF_RMASK = INF_RING - 1, INF_FLUSH = 4096;
constexpr int INF_IN = 8192, INF_CHUNK = 4096;
constexpr int LL_BITS = 11, D_BITS = 10;

__constant__ uint16_t c_lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t c_lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint16_t c_dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__constant__ uint8_t c_dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t c_clorder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

struct Huff {          // canonical code of one alphabet (puff.c style), in LDS
	uint16_t count[16];  // number of codes of each length
	uint16_t symbol[320];// symbols ordered by (length, symbol)
};

struct InfShared {
	uint8_t ring[INF_RING];
	uint8_t inbuf[INF_IN];
	uint16_t ll_fast[1 << LL_BITS], d_fast[1 << D_BITS];
	Huff ll, dd, cl;
	uint8_t lens[320];
};

struct BitReader {
	const uint8_t *src;   // 16-byte aligned base of the staged stream
	uint64_t avail;       // bytes of the whole input buffer readable from src (zeros are fed beyond)
	uint64_t nbytes;      // bytes from src to the end of THIS stream (consuming more = truncated stream)
	uint64_t bytepos;     // next byte to move into the bit buffer (relative to src)
	uint64_t staged_end;  // inbuf holds [staged_end - INF_IN, staged_end), staged_end % INF_CHUNK == 0
	uint64_t buf;
	int cnt;
};

__device__ __forceinline__ void stage_chunk(InfShared &S, BitReader &br, int lane)
{
	uint4 *dst = reinterpret_cast<uint4 *>(S.inbuf + (br.staged_end & (INF_IN - 1)));
	const uint4 *srcv = reinterpret_cast<const uint4 *>(br.src + br.staged_end);
	for (int t = lane; t < INF_CHUNK / 16; t += 64) {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (br.staged_end + (uint64_t)t * 16 + 16 <= br.avail) v = srcv[t];
		dst[t] = v;
	}
	br.staged_end += INF_CHUNK;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void refill(InfShared &S, BitReader &br, int lane)
{
	while (br.cnt <= 32) {
		while (br.bytepos + 4 > br.staged_end) stage_chunk(S, br, lane);
		const uint8_t *q = S.inbuf;
		const uint32_t w = (uint32_t)q[br.bytepos & (INF_IN - 1)] | ((uint32_t)q[(br.bytepos + 1) & (INF_IN - 1)] << 8) |
		                   ((uint32_t)q[(br.bytepos + 2) & (INF_IN - 1)] << 16) | ((uint32_t)q[(br.bytepos + 3) & (INF_IN - 1)] << 24);
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

// canonical decode, one bit at a time (puff.c decode()); returns -1 on an invalid code
__device__ int slow_decode(BitReader &br, const Huff &h)
{
	int code = 0, first = 0, index = 0;
	for (int len = 1; len <= 15; len++) {
		code |= (int)getbits(br, 1);
		const int count = h.count[len];
		if (code - count < first) return h.symbol[index + (code - first)];
		index += count;
		first += count;
		first <<= 1;
		code <<= 1;
	}
	return -1;
}

// build canonical tables from S.lens[0..n) (wave-cooperative); false = over-subscribed code
__device__ bool build_huff(InfShared &S, Huff &h, int n, uint16_t *fast, int fast_bits, int lane)
{
	if (lane < 16) h.count[lane] = 0;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
	if (lane == 0)
		for (int s = 0; s < n; s++) h.count[S.lens[s]]++;
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
	for (int i = lane; fast && i < (1 << fast_bits); i += 64) fast[i] = 0;
	int left = 1;
	uint16_t offs[16], code_first[16];
	uint32_t c = 0;
	offs[0] = 0; offs[1] = 0; code_first[0] = 0;
	for (int len = 1; len <= 15; len++) {
		left <<= 1;
		left -= h.count[len];
		if (left < 0) return false;
		c = (c + (len > 1 ? h.count[len - 1] : 0)) << 1;  // canonical: code(len) = (code(len-1) + count(len-1)) << 1
		code_first[len] = (uint16_t)c;
		if (len < 15) offs[len + 1] = (uint16_t)(offs[len] + h.count[len]);
	}
	if (lane == 0) {
		uint16_t o[16];
		for (int len = 1; len <= 15; len++) o[len] = offs[len];
		for (int s = 0; s < n; s++) if (S.lens[s]) h.symbol[o[S.lens[s]]++] = (uint16_t)s;
	}
	__builtin_amdgcn_s_waitcnt(0);
	__builtin_amdgcn_wave_barrier();
	if (fast) {
		// symbol s of length l has code code_first[l] + (number of smaller symbols of the same length)
		for (int s = lane; s < n; s += 64) {
			const int l = S.lens[s];
			if (l == 0 || l > fast_bits) continue;
			int rank = 0;
			for (int t = 0; t < s; t++) rank += (S.lens[t] == l) ? 1 : 0;
			const uint32_t cc = code_first[l] + (uint32_t)rank;
			uint32_t r = 0;
			for (int bb = 0; bb < l; bb++) r |= ((cc >> bb) & 1u) << (l - 1 - bb);  // stream order is LSB first
			for (uint32_t k = r; k < (1u << fast_bits); k += (1u << l)) fast[k] = (uint16_t)((s << 4) | l);
		}
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_wave_barrier();
	}
	return true;
}

__global__ void __launch_bounds__(64) inflate_kernel(InflateArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
	InfShared &S = *reinterpret_cast<InfShared *>(smem_raw);
	const int s = blockIdx.x;
	const int lane = threadIdx.x;
	const uint64_t f0 = a.offsets[s] + (uint64_t)a.skip, f1 = a.offsets[s + 1];
	uint8_t *out = a.out + (size_t)s * a.out_stride;
	uint32_t err = 0;
	uint32_t pos = 0, flushed = 0;
	uint32_t adA = 1, adB = 0;

	BitReader br;
	const uint64_t al = f0 & ~(uint64_t)15;
	br.src = a.in + al;
	br.avail = a.in_total > al ? a.in_total - al : 0;
	br.nbytes = f1 > al ? f1 - al : 0;
	br.bytepos = f0 - al;
	br.staged_end = br.bytepos & ~(uint64_t)(INF_CHUNK - 1);
	br.buf = 0; br.cnt = 0;
	if (f1 < f0 + 6) err = CCT_ST_ZLIB;  // shorter than header + trailer

	auto flush_to = [&](uint32_t upto) {  // ring [flushed, upto) -> HBM as 16-byte stores, Adler-32 on the way
		while (flushed < upto) {
			const uint32_t n = min((uint32_t)INF_FLUSH, upto - flushed);
			if ((size_t)flushed + ((n + 15) & ~15u) > a.out_stride) { err |= CCT_ST_STREAM; flushed += n; continue; }  // longer than any valid payload
			uint32_t sa = 0, sb = 0;  // A += sum d ; B += n * A_old + sum (n - i) d_i
			for (uint32_t t = lane; t < n; t += 64) {
				const uint32_t d = S.ring[(flushed + t) & INF_RMASK];
				sa += d; sb += (n - t) * d;
			}
			for (int d = 32; d > 0; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
			adB = (uint32_t)(((uint64_t)adB + (uint64_t)n * adA + sb) % 65521u);
			adA = (adA + sa) % 65521u;
			for (uint32_t t = lane * 16; t < ((n + 15) & ~15u); t += 64 * 16) {
				const uint4 v = *reinterpret_cast<const uint4 *>(S.ring + ((flushed + t) & INF_RMASK));
				*reinterpret_cast<uint4 *>(out + flushed + t) = v;
			}
			flushed += n;
		}
	};
	auto runaway = [&]() -> bool { return consumed_bytes(br) > br.nbytes + 16; };  // decoding zeros past the end

	if (!err) {
		refill(S, br, lane);
		const uint32_t cmf = getbits(br, 8), flg = getbits(br, 8);
		if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) err = CCT_ST_ZLIB;  // incorrect header check
	}
	bool last = false;
	while (!err && !last) {
		refill(S, br, lane);
		last = getbits(br, 1) != 0;
		const uint32_t type = getbits(br, 2);
		if (type == 0) {  // stored
			const int drop = br.cnt & 7;
			getbits(br, drop);
			refill(S, br, lane);
			const uint32_t len = getbits(br, 16);
			refill(S, br, lane);
			const uint32_t nlen = getbits(br, 16);
			if ((len ^ 0xFFFFu) != nlen) { err = CCT_ST_ZLIB; break; }
			for (uint32_t i = 0; i < len; i++) {  // byte-wise through the bit buffer keeps one input path
				refill(S, br, lane);
				const uint32_t b = getbits(br, 8);
				if (lane == 0) S.ring[pos & INF_RMASK] = (uint8_t)b;
				pos++;
				if (pos - flushed >= 2 * INF_FLUSH) flush_to(pos & ~(uint32_t)(INF_FLUSH - 1));
			}
			if (runaway()) { err = CCT_ST_ZLIB; break; }
			continue;
		}
		if (type == 3) { err = CCT_ST_ZLIB; break; }
		if (type == 1) {  // fixed codes
			for (int i = lane; i < 288; i += 64) S.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			build_huff(S, S.ll, 288, S.ll_fast, LL_BITS, lane);
			for (int i = lane; i < 30; i += 64) S.lens[i] = 5;
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			build_huff(S, S.dd, 30, S.d_fast, D_BITS, lane);
		} else {          // dynamic codes
			refill(S, br, lane);
			const int nlen = (int)getbits(br, 5) + 257, ndist = (int)getbits(br, 5) + 1, ncode = (int)getbits(br, 4) + 4;
			if (nlen > 286 || ndist > 30) { err = CCT_ST_ZLIB; break; }
			if (lane < 19) S.lens[lane] = 0;
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			for (int i = 0; i < ncode; i++) {
				refill(S, br, lane);
				const uint32_t v = getbits(br, 3);
				if (lane == 0) S.lens[c_clorder[i]] = (uint8_t)v;
			}
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			if (!build_huff(S, S.cl, 19, nullptr, 0, lane)) { err = CCT_ST_ZLIB; break; }
			// the code lengths of both alphabets, run-length coded; S.lens is reused after cl is built
			__builtin_amdgcn_wave_barrier();
			int idx = 0;
			uint8_t prev_len = 0;
			while (idx < nlen + ndist) {
				refill(S, br, lane);
				const int sym = slow_decode(br, S.cl);
				if (sym < 0) { err = CCT_ST_ZLIB; break; }
				if (sym < 16) { if (lane == 0) S.lens[idx] = (uint8_t)sym; prev_len = (uint8_t)sym; idx++; }
				else {
					int rep; uint8_t v = 0;
					if (sym == 16) { if (idx == 0) { err = CCT_ST_ZLIB; break; } v = prev_len; rep = 3 + (int)getbits(br, 2); }
					else if (sym == 17) rep = 3 + (int)getbits(br, 3);
					else rep = 11 + (int)getbits(br, 7);
					if (idx + rep > nlen + ndist) { err = CCT_ST_ZLIB; break; }
					for (int t = lane; t < rep; t += 64) S.lens[idx + t] = v;
					idx += rep;
					if (sym != 16) prev_len = 0;
				}
			}
			if (err) break;
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			if (S.lens[256] == 0) { err = CCT_ST_ZLIB; break; }  // no end-of-block code
			// distance lengths follow the literal/length lengths: move them to their own array slot first
			uint8_t dl = 0;
			if (lane < ndist) dl = S.lens[nlen + lane];
			__builtin_amdgcn_wave_barrier();
			if (!build_huff(S, S.ll, nlen, S.ll_fast, LL_BITS, lane)) { err = CCT_ST_ZLIB; break; }
			if (lane < 30) S.lens[lane] = lane < ndist ? dl : 0;
			__builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
			if (!build_huff(S, S.dd, ndist, S.d_fast, D_BITS, lane)) { err = CCT_ST_ZLIB; break; }
		}
		// ---- symbols of this block
		for (;;) {
			refill(S, br, lane);
			int sym;
			const uint32_t e = S.ll_fast[br.buf & ((1u << LL_BITS) - 1u)];
			if (e) { sym = (int)(e >> 4); getbits(br, (int)(e & 15)); }
			else sym = slow_decode(br, S.ll);
			if (sym < 0) { err = CCT_ST_ZLIB; break; }
			if (sym < 256) {
				if (lane == 0) S.ring[pos & INF_RMASK] = (uint8_t)sym;
				pos++;
			} else if (sym == 256) {
				break;
			} else {
				const int li = sym - 257;
				if (li >= 29) { err = CCT_ST_ZLIB; break; }
				const uint32_t length = c_lbase[li] + getbits(br, c_lext[li]);
				refill(S, br, lane);
				int ds;
				const uint32_t de = S.d_fast[br.buf & ((1u << D_BITS) - 1u)];
				if (de) { ds = (int)(de >> 4); getbits(br, (int)(de & 15)); }
				else ds = slow_decode(br, S.dd);
				if (ds < 0 || ds >= 30) { err = CCT_ST_ZLIB; break; }
				const uint32_t dist = c_dbase[ds] + getbits(br, c_dext[ds]);
				if (dist > pos) { err = CCT_ST_ZLIB; break; }  // distance too far back
				// wave-wide copy; an overlapping copy repeats its first `dist` bytes
				for (uint32_t i = lane; i < length; i += 64) {
					const uint32_t srci = pos - dist + (dist >= length ? i : i % dist);
					S.ring[(pos + i) & INF_RMASK] = S.ring[srci & INF_RMASK];
				}
				pos += length;
			}
			if (pos - flushed >= 2 * INF_FLUSH) {
				flush_to(pos & ~(uint32_t)(INF_FLUSH - 1));
				if (runaway() || err) { err |= CCT_ST_ZLIB; break; }
			}
		}
		if (runaway()) err |= CCT_ST_ZLIB;
	}
	if (!err) {
		flush_to(pos);
		// Adler-32 trailer, big-endian, after the bit reader is byte aligned
		getbits(br, br.cnt & 7);
		uint32_t want = 0;
		for (int k = 0; k < 4; k++) { refill(S, br, lane); want = (want << 8) | getbits(br, 8); }
		if (consumed_bytes(br) > br.nbytes) err = CCT_ST_ZLIB;         // incomplete or truncated stream
		else if (want != ((adB << 16) | adA)) err = CCT_ST_ZLIB;        // incorrect data check
	}
	if (lane == 0) {
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
	hipLaunchKernelGGL(inflate_kernel, dim3(n), dim3(64), lds, st, a);
	return hipGetLastError();
}

}  // namespace cct
