// PackBits run-length utility of the reference (src/codec/packbits.py:28-163; dead code there: nothing imports it and
// the .cct path never calls it -- SURVEY 8f.4), restated as wave-scan kernels: one wave per byte string.
//
// The reference encoder is a byte-at-a-time state machine (packbits.py:74-129).  What it emits depends only on the
// maximal SEGMENTS of the input -- runs (>= 2 equal bytes) and literal stretches -- and on the position inside a segment:
//   run of L bytes      -> chunks of 127 while more than 128 bytes remain, then one chunk of 2..128 (packbits.py:95-100,
//                          123-125: the count is bumped once more when the run ends, so the last chunk may reach 128);
//                          a chunk is two bytes: 257 - length, value
//   literal stretch     -> chunks of 127 (flushed before the 128th byte is appended, packbits.py:111-114), except that the
//                          very last byte of the DATA is appended without that check (packbits.py:119-121): a stretch that
//                          ends the data may close with a chunk of 128; a chunk is length - 1, then the bytes
// so every byte can decide by itself whether it opens a chunk and how many output bytes it contributes; a prefix sum of
// those contributions gives each byte its output offset.  Segment ends come from a backward sweep (ballot: next
// boundary at or after the lane), segment starts and offsets from a forward sweep.
// Optional byte-delta transform (packbits.py:43-63): x[i] - x[i-1] mod 256 before encoding, prefix sum after decoding.
#include "cct_internal.h"
#include "../../include/compact_hip.h"

namespace cct {
namespace {

struct Str {
	const uint8_t *p; uint32_t n; int delta;
	__device__ __forceinline__ uint32_t at(uint32_t i) const
	{
		const uint32_t v = p[i];
		return (delta && i) ? ((v - p[i - 1]) & 0xFFu) : v;  // packbits.py:43-51
	}
	// byte i belongs to a run: it equals a neighbour
	__device__ __forceinline__ bool in_run(uint32_t i) const { return (i > 0 && at(i) == at(i - 1)) || (i + 1 < n && at(i) == at(i + 1)); }
	// a segment ends after byte i
	__device__ __forceinline__ bool brk(uint32_t i) const
	{
		if (i + 1 >= n) return true;
		const bool r0 = in_run(i), r1 = in_run(i + 1);
		return r0 != r1 || (r0 && at(i) != at(i + 1));
	}
};

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
	uint32_t inc = v;
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
		if ((int)(threadIdx.x & 63) >= d) inc += o;
	}
	total = (uint32_t)__shfl((int)inc, 63);
	return inc - v;
}

__global__ void __launch_bounds__(64) packbits_encode_kernel(const uint8_t *in, const uint64_t *offsets, int delta, uint32_t *segend_ws,
                                                            uint8_t *out, size_t out_stride, uint32_t *out_sizes)
{
	const int s = blockIdx.x, lane = threadIdx.x;
	const uint64_t o0 = offsets[s];
	Str x{in + o0, (uint32_t)(offsets[s + 1] - o0), delta};
	uint8_t *dst = out + (size_t)s * out_stride;
	if (x.n == 0) { if (lane == 0) out_sizes[s] = 0; return; }
	if (x.n == 1) { if (lane == 0) { dst[0] = 0; dst[1] = x.p[0]; out_sizes[s] = 2; } return; }  // packbits.py:82-83
	uint32_t *segend = segend_ws + o0;
	// ---- backward sweep: last byte of the segment that holds byte i
	uint32_t carry = x.n - 1;
	for (int64_t base = (int64_t)((x.n - 1) / 64) * 64; base >= 0; base -= 64) {
		const uint32_t i = (uint32_t)base + lane;
		const bool b = i < x.n && x.brk(i);
		const uint64_t bal = __ballot(b) >> lane;  // boundaries at or after this lane
		if (i < x.n) segend[i] = bal ? i + (uint32_t)__ffsll((long long)bal) - 1u : carry;
		const uint64_t all = __ballot(b);
		if (all) carry = (uint32_t)base + (uint32_t)__ffsll((long long)all) - 1u;
	}
	__syncthreads();  // one wave; makes the scratch stores visible to the loads below
	// ---- forward sweep: segment start, chunk heads, output offsets
	uint32_t seg_carry = 0, out_pos = 0;
	for (uint32_t base = 0; base < x.n; base += 64) {
		const uint32_t i = base + lane;
		const bool valid = i < x.n;
		const bool starts = valid && (i == 0 || x.brk(i - 1));
		const uint64_t sb = __ballot(starts);
		const uint64_t upto = sb & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));  // starts at or before this lane
		const uint32_t sstart = upto ? base + 63u - (uint32_t)__clzll((long long)upto) : seg_carry;
		if (sb) seg_carry = base + 63u - (uint32_t)__clzll((long long)sb);
		uint32_t contrib = 0, hdr = 0, v = 0;
		bool head = false, lit = false;
		if (valid) {
			const uint32_t e = segend[i], L = e - sstart + 1u, j = i - sstart;
			v = x.at(i);
			lit = !x.in_run(i);
			if (lit) {
				uint32_t len;
				if (e + 1 < x.n) { head = j % 127u == 0; len = min(127u, L - j); }  // flushed by the run that follows
				else {  // the stretch ends the data: its last byte joins the open chunk unchecked
					head = (j % 127u == 0) && (j + 1 < L || L == 1);
					const uint32_t c_end = min(j + 127u, L - 1u);
					len = (c_end == L - 1u) ? L - j : 127u;
				}
				hdr = len - 1u;
				contrib = 1u + (head ? 1u : 0u);
			} else {
				const uint32_t k127 = L > 128u ? (L - 128u + 126u) / 127u : 0u;
				const uint32_t tail = 127u * k127;
				head = (j < tail && j % 127u == 0) || j == tail;
				const uint32_t len = j < tail ? 127u : L - tail;
				hdr = 257u - len;  // packbits.py:70-72: 256 - (run - 1)
				contrib = head ? 2u : 0u;
			}
		}
		uint32_t tot;
		const uint32_t off = out_pos + wave_excl_scan(contrib, tot);
		if (valid) {
			if (lit) { if (head) { dst[off] = (uint8_t)hdr; dst[off + 1] = (uint8_t)v; } else dst[off] = (uint8_t)v; }
			else if (head) { dst[off] = (uint8_t)hdr; dst[off + 1] = (uint8_t)v; }
		}
		out_pos += tot;
	}
	if (lane == 0) out_sizes[s] = out_pos;
}

// packbits.py:131-160: header 0..127 -> that many + 1 literal bytes; 129..255 -> 257 - header copies of the next byte;
// 128 -> nothing.  Lane 0 walks the headers (each tells where the next one is), the wave copies / fills.
__global__ void __launch_bounds__(64) packbits_decode_kernel(const uint8_t *in, const uint64_t *offsets, int delta, uint8_t *out,
                                                            size_t out_stride, uint32_t *out_sizes, uint32_t *status)
{
	const int s = blockIdx.x, lane = threadIdx.x;
	const uint8_t *src = in + offsets[s];
	const uint32_t n = (uint32_t)(offsets[s + 1] - offsets[s]);
	uint8_t *dst = out + (size_t)s * out_stride;
	uint32_t pos = 0, o = 0, st = 0;
	while (pos < n) {
		const uint32_t h = src[pos++];
		if (h <= 127u) {
			const uint32_t cnt = min(h + 1u, n - pos);  // a Python slice past the end just comes up short (packbits.py:150)
			if ((size_t)o + cnt > out_stride) { st = CCT_E_CAP; break; }
			for (uint32_t k = lane; k < cnt; k += 64) dst[o + k] = src[pos + k];
			pos += h + 1u; o += cnt;
		} else if (h != 128u) {
			if (pos >= n) { st = CCT_E_STREAM; break; }  // the reference raises IndexError here
			const uint32_t cnt = 257u - h;
			if ((size_t)o + cnt > out_stride) { st = CCT_E_CAP; break; }
			const uint8_t v = src[pos++];
			for (uint32_t k = lane; k < cnt; k += 64) dst[o + k] = v;
			o += cnt;
		}
	}
	__syncthreads();
	if (delta && !st && o > 1) {  // packbits.py:53-63: running sum mod 256 (a delta above 127 counts as negative: the same mod 256)
		uint32_t carry = 0;
		for (uint32_t base = 0; base < o; base += 64) {
			const uint32_t i = base + lane;
			const uint32_t v = i < o ? dst[i] : 0u;
			uint32_t tot;
			const uint32_t ex = wave_excl_scan(v, tot);
			if (i < o) dst[i] = (uint8_t)(carry + ex + v);
			carry += tot;
		}
	}
	if (lane == 0) { out_sizes[s] = st ? 0u : o; status[s] = st; }
}

}  // namespace

hipError_t launch_packbits_encode(const uint8_t *d_in, const uint64_t *d_offsets, int n, int delta, uint32_t *d_ws, uint8_t *d_out,
                                  size_t out_stride, uint32_t *d_out_sizes, hipStream_t st)
{
	hipLaunchKernelGGL(packbits_encode_kernel, dim3(n), dim3(64), 0, st, d_in, d_offsets, delta, d_ws, d_out, out_stride, d_out_sizes);
	return hipGetLastError();
}

hipError_t launch_packbits_decode(const uint8_t *d_in, const uint64_t *d_offsets, int n, int delta, uint8_t *d_out, size_t out_stride,
                                  uint32_t *d_out_sizes, uint32_t *d_status, hipStream_t st)
{
	hipLaunchKernelGGL(packbits_decode_kernel, dim3(n), dim3(64), 0, st, d_in, d_offsets, delta, d_out, out_stride, d_out_sizes, d_status);
	return hipGetLastError();
}

}  // namespace cct
