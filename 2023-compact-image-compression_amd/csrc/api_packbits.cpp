// PackBits utility of the reference (src/codec/packbits.py) behind the C ABI: host side of packbits_kernels.hip.
#include <algorithm>
#include <mutex>
#include <vector>

#include "cct_internal.h"
#include "host.h"

using namespace cct;

extern "C" {

size_t cct_packbits_bound(size_t n_bytes) { return 2 * n_bytes + 2; }  // a byte opens a chunk at worst: two output bytes each

static int packbits_batch(bool encode, const uint8_t *h_in, const uint64_t *h_offsets, int n, int delta, uint8_t *h_out,
                          size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status)
{
	if (n < 0 || !h_offsets || (n > 0 && (!h_in || !h_out || !h_out_sizes))) return fail(CCT_E_ARG, "bad argument");
	if (n == 0) return CCT_OK;
	std::lock_guard<std::mutex> lk(g_mu);
	ApiCall in_call;
	int rc = ensure_ctx();
	if (rc) return rc;
	hipStream_t st = main_stream();
	const size_t total = (size_t)h_offsets[n];
	size_t longest = 0;
	for (int i = 0; i < n; i++) {
		if (h_offsets[i + 1] < h_offsets[i]) return fail(CCT_E_ARG, "offsets must not decrease");
		longest = std::max(longest, (size_t)(h_offsets[i + 1] - h_offsets[i]));
	}
	if (encode && out_stride < cct_packbits_bound(longest)) return fail(CCT_E_CAP, "out_stride %zu below cct_packbits_bound(%zu)", out_stride, longest);
	DevBuf d_in, d_offs, d_ws, d_out, d_sizes, d_status;
	auto release = [&] { exclusive_section([&]() -> int { d_in.release(); d_offs.release(); d_ws.release(); d_out.release(); d_sizes.release(); d_status.release(); return 0; }); };
	if ((rc = d_in.ensure(total + 16)) || (rc = d_offs.ensure((size_t)(n + 1) * 8)) || (rc = d_ws.ensure((total + 1) * 4)) ||
	    (rc = d_out.ensure((size_t)n * out_stride + 16)) || (rc = d_sizes.ensure((size_t)n * 4)) || (rc = d_status.ensure((size_t)n * 4))) { release(); return rc; }
	hipError_t e = hipSuccess;
	auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
	if (total) step(hipMemcpyAsync(d_in.p, h_in, total, hipMemcpyHostToDevice, st));
	step(hipMemcpyAsync(d_offs.p, h_offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
	step(hipMemsetAsync(d_status.p, 0, (size_t)n * 4, st));
	if (encode) step(launch_packbits_encode((const uint8_t *)d_in.p, (const uint64_t *)d_offs.p, n, delta, (uint32_t *)d_ws.p, (uint8_t *)d_out.p,
	                                        out_stride, (uint32_t *)d_sizes.p, st));
	else step(launch_packbits_decode((const uint8_t *)d_in.p, (const uint64_t *)d_offs.p, n, delta, (uint8_t *)d_out.p, out_stride,
	                                 (uint32_t *)d_sizes.p, (uint32_t *)d_status.p, st));
	step(hipMemcpyAsync(h_out_sizes, d_sizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	std::vector<uint32_t> stv(n, 0);
	step(hipMemcpyAsync(stv.data(), d_status.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	step(hipStreamSynchronize(st));
	int first = CCT_OK;
	if (e == hipSuccess)
		for (int i = 0; i < n && e == hipSuccess; i++) {
			if (h_status) h_status[i] = stv[i];
			if (stv[i] && !first) first = (int)stv[i];
			// (asynchronous copies on the library's stream, one wait at the end: a synchronous hipMemcpy runs on the default stream and
			// breaks a graph capture another thread may have open, profiles/r03_capture_vs_free.log)
			if (!stv[i] && h_out_sizes[i]) e = hipMemcpyAsync(h_out + (size_t)i * out_stride, (const uint8_t *)d_out.p + (size_t)i * out_stride, h_out_sizes[i], hipMemcpyDeviceToHost, st);
		}
	{ const hipError_t e2 = hipStreamSynchronize(st); if (e == hipSuccess) e = e2; }
	release();
	if (e != hipSuccess) return fail(CCT_E_DEVICE, "packbits: %s", hipGetErrorString(e));
	return first ? fail(first, "packbits decode: string rejected") : CCT_OK;
}

int cct_packbits_encode_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, int delta_transform, uint8_t *h_out,
                              size_t out_stride, uint32_t *h_out_sizes)
{
	return packbits_batch(true, h_in, h_offsets, n, delta_transform, h_out, out_stride, h_out_sizes, nullptr);
}

int cct_packbits_decode_batch(const uint8_t *h_in, const uint64_t *h_offsets, int n, int delta_transform, uint8_t *h_out,
                              size_t out_stride, uint32_t *h_out_sizes, uint32_t *h_status)
{
	return packbits_batch(false, h_in, h_offsets, n, delta_transform, h_out, out_stride, h_out_sizes, h_status);
}

}  // extern "C"
