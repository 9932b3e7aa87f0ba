// Host-side declarations shared by the translation units of libcompact_hip.so (api.cpp: context, encode, decode;
// api_comm.cpp: RCCL all-gather; api_packbits.cpp: PackBits utility).  Not part of the C ABI.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
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

// Runtime operations that are rare and heavy -- stream capture and graph instantiation / destruction, the per-shape table
// builds (device allocations plus SYNCHRONOUS copies), workspace growth, stream creation -- never run next to another call of
// the library: every entry point that drives the device holds `g_quiesce` shared from the moment it owns its slot (ApiCall), and
// those operations give it up and take it exclusively (exclusive_section).  Order: slot mutex, then g_quiesce; a thread waiting
// for a slot holds neither, and nobody waits for g_mu while it counts as a call in flight.
// Why: round 2 saw two concurrent encode calls hang inside the runtime about once in ten runs, one of them capturing its graphs
// for the first time.  Round 3 ran the suspects side by side without a lock (tools/debug/capture_vs_free.cpp,
// profiles/r03_capture_vs_free.log): allocations, frees, pinned allocations and stream creation next to an open capture neither
// stalled nor failed; a SYNCHRONOUS hipMemcpy (default stream) did -- hipStreamEndCapture then reports "capturing stream has
// unjoined work" and the runtime dies.  The library's synchronous copies are the table builds (get_tables), which a second
// thread could reach while the first one captured; they are inside the exclusive section now, and no steady-state path copies
// synchronously any more.
struct QuiesceLock {  // shared / exclusive with priority for the exclusive side (glibc's rwlock prefers readers: with three
	std::mutex m;        // threads issuing calls back to back an exclusive section could wait for a long time)
	std::condition_variable cv;
	int readers = 0, writers_waiting = 0;
	bool writer = false;
	void lock_shared() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !writer && writers_waiting == 0; }); readers++; }
	void unlock_shared() { std::unique_lock<std::mutex> l(m); if (--readers == 0) cv.notify_all(); }
	void lock() { std::unique_lock<std::mutex> l(m); writers_waiting++; cv.wait(l, [&] { return !writer && readers == 0; }); writers_waiting--; writer = true; }
	void unlock() { std::unique_lock<std::mutex> l(m); writer = false; cv.notify_all(); }
};
extern QuiesceLock g_quiesce;
extern thread_local int tl_api_depth;
struct ApiCall {
	ApiCall() { if (tl_api_depth++ == 0) g_quiesce.lock_shared(); }
	~ApiCall() { if (--tl_api_depth == 0) g_quiesce.unlock_shared(); }
	ApiCall(const ApiCall &) = delete;
	ApiCall &operator=(const ApiCall &) = delete;
};
template <class F>
auto exclusive_section(F f) -> decltype(f())
{
	const bool shared = tl_api_depth > 0;
	if (shared) g_quiesce.unlock_shared();
	g_quiesce.lock();
	struct Back { bool shared; ~Back() { g_quiesce.unlock(); if (shared) g_quiesce.lock_shared(); } } back{shared};
	return f();
}

struct DevBuf {  // grow-only device (or pinned host) buffer
	void *p = nullptr;
	size_t cap = 0;
	bool pinned_host = false;
	int ensure(size_t bytes)
	{
		if (bytes <= cap) return CCT_OK;
		return exclusive_section([&]() -> int {
			release();
			const size_t want = bytes + bytes / 8 + 4096;
			hipError_t e = pinned_host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
			if (e != hipSuccess) { p = nullptr; cap = 0; return fail(CCT_E_NOMEM, "allocation of %zu bytes failed: %s", want, hipGetErrorString(e)); }
			cap = want;
			return CCT_OK;
		});
	}
	// hipFree / hipHostFree wait for the whole device: callers run them inside an exclusive section (ensure() does) or with
	// every stream of the library drained (cct_shutdown)
	void release()
	{
		if (p) { if (pinned_host) (void)hipHostFree(p); else (void)hipFree(p); }
		p = nullptr; cap = 0;
	}
	void release_exclusive() { if (p) exclusive_section([&]() -> int { release(); return 0; }); }
};

extern std::mutex g_mu;          // device context, main stream (and with it encode slot 0), every plumbing call
int ensure_ctx(int device = -1); // binds the device on first use (call with g_mu held); CCT_E_DEVICE in a child forked after that
hipStream_t main_stream();       // valid once ensure_ctx() has succeeded
int bound_device();
bool forked_after_init();        // this process is a fork() child of the one that initialised the device
void comm_release();             // cct_shutdown: drop the communicator and its buffers (api_comm.cpp)

}  // namespace cct
