// Diagnostic (not part of the product): which HIP runtime calls of one thread block when another thread is between
// hipStreamBeginCapture and hipGraphInstantiate?  Round 2 saw two concurrent encode calls hang "inside the runtime" when one
// of them captured its graphs for the first time while the other grew a workspace; the library has since run such
// operations one at a time (host.h, exclusive_section).  This program runs the two suspects side by side WITHOUT that lock:
//   thread A   capture (thread-local mode) -> instantiate -> launch -> sync -> destroy, in a loop
//   thread B   one of: hipMalloc + hipFree | hipHostMalloc + hipHostFree | hipMalloc only (pool) | hipStreamCreate + Destroy
// Every runtime call is announced in an atomic before it is made; a watchdog prints both announcements and leaves with
// code 3 if neither thread has finished a call for 5 s.     hipcc -O2 -o capture_vs_free capture_vs_free.cpp -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>

static std::atomic<const char *> g_call[2];
static std::atomic<unsigned long> g_done[2];
static std::atomic<bool> g_stop{false};
#define CALL(t, x) do { g_call[t] = #x; hipError_t e_ = (x); g_done[t]++; if (e_ != hipSuccess) { fprintf(stderr, "thread %d: %s -> %s\n", t, #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)

__global__ void touch(int *p) { if (p) atomicAdd(p, 1); }

static void capture_loop(int *d)
{
	hipStream_t s;
	CALL(0, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	while (!g_stop) {
		hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
		CALL(0, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		for (int k = 0; k < 20; k++) hipLaunchKernelGGL(touch, dim3(64), dim3(64), 0, s, d);
		CALL(0, hipStreamEndCapture(s, &g));
		CALL(0, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		CALL(0, hipGraphLaunch(ge, s));
		CALL(0, hipStreamSynchronize(s));
		CALL(0, hipGraphExecDestroy(ge));
		CALL(0, hipGraphDestroy(g));
	}
	CALL(0, hipStreamDestroy(s));
}

static void other_loop(int mode)
{
	hipStream_t s;
	CALL(1, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	void *keep[64] = {nullptr};
	int nkeep = 0;
	while (!g_stop) {
		void *p = nullptr;
		if (mode == 0) {        // grow a device workspace: free (waits for the device) + allocate
			CALL(1, hipMalloc(&p, 64u << 20));
			CALL(1, hipMemsetAsync(p, 0, 1 << 20, s));
			CALL(1, hipStreamSynchronize(s));
			CALL(1, hipFree(p));
		} else if (mode == 1) { // pinned host staging
			CALL(1, hipHostMalloc(&p, 16u << 20, hipHostMallocDefault));
			CALL(1, hipHostFree(p));
		} else if (mode == 2) { // allocate only
			CALL(1, hipMalloc(&p, 1u << 20));
			if (nkeep < 64) keep[nkeep++] = p; else { g_stop = true; }
		} else if (mode == 3) { // stream creation
			hipStream_t t;
			CALL(1, hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
			CALL(1, hipStreamDestroy(t));
		} else {                // synchronous copy (what a table build does)
			static int h[1024];
			CALL(1, hipMalloc(&p, 4096));
			CALL(1, hipMemcpy(p, h, 4096, hipMemcpyHostToDevice));
			CALL(1, hipFree(p));
		}
	}
	for (int i = 0; i < nkeep; i++) (void)hipFree(keep[i]);
	CALL(1, hipStreamDestroy(s));
}

int main(int argc, char **argv)
{
	const int mode = argc > 1 ? atoi(argv[1]) : 0;
	const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
	int *d = nullptr;
	if (hipMalloc(&d, 4) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
	(void)hipMemset(d, 0, 4);
	g_call[0] = g_call[1] = "start";
	std::thread a(capture_loop, d), b(other_loop, mode);
	const auto t0 = std::chrono::steady_clock::now();
	unsigned long last[2] = {0, 0};
	auto last_move = t0;
	for (;;) {
		std::this_thread::sleep_for(std::chrono::milliseconds(100));
		const auto now = std::chrono::steady_clock::now();
		const unsigned long c0 = g_done[0], c1 = g_done[1];
		if (c0 != last[0] || c1 != last[1]) { last[0] = c0; last[1] = c1; last_move = now; }
		if (std::chrono::duration<double>(now - last_move).count() > 5.0) {
			printf("mode %d: STUCK after %lu / %lu calls: capture thread in [%s], other thread in [%s]\n", mode, c0, c1, g_call[0].load(), g_call[1].load());
			fflush(stdout);
			_exit(3);
		}
		if (std::chrono::duration<double>(now - t0).count() > seconds) break;
	}
	g_stop = true;
	a.join(); b.join();
	printf("mode %d: no stall in %.1f s (%lu capture-thread calls, %lu other-thread calls)\n", mode, seconds, g_done[0].load(), g_done[1].load());
	return 0;
}
