// Host-side declarations shared by the translation units of libcompact_hip.so (api.cpp: context, encode, decode;
// api_comm.cpp: RCCL all-gather; api_packbits.cpp: PackBits utility).  Not part of the C ABI.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <mutex>

#include <hip/hip_runtime.h>

#include "../../include/compact_hip.h"

namespace cct {

// last error of the calling thread (cct_last_error); returns `code` so that `return fail(...)` reads well
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                          \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess)                                                                      \
			return ::cct::fail(CCT_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));         \
	} while (0)

struct DevBuf {  // grow-only device (or pinned host) buffer
	void *p = nullptr;
	size_t cap = 0;
	bool pinned_host = false;
	int ensure(size_t bytes)
	{
		if (bytes <= cap) return CCT_OK;
		release();
		const size_t want = bytes + bytes / 8 + 4096;
		hipError_t e = pinned_host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
		if (e != hipSuccess) { p = nullptr; cap = 0; return fail(CCT_E_NOMEM, "allocation of %zu bytes failed: %s", want, hipGetErrorString(e)); }
		cap = want;
		return CCT_OK;
	}
	void release()
	{
		if (p) { if (pinned_host) (void)hipHostFree(p); else (void)hipFree(p); }
		p = nullptr; cap = 0;
	}
};

extern std::mutex g_mu;          // device context, main stream (and with it encode slot 0), every plumbing call
int ensure_ctx(int device = -1); // binds the device on first use (call with g_mu held); CCT_E_DEVICE in a child forked after that
hipStream_t main_stream();       // valid once ensure_ctx() has succeeded
int bound_device();
bool forked_after_init();        // this process is a fork() child of the one that initialised the device
void comm_release();             // cct_shutdown: drop the communicator and its buffers (api_comm.cpp)

}  // namespace cct
