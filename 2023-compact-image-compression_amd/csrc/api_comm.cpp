// RCCL all-gather of the compressed sizes across ranks, behind the C ABI (include/compact_hip.h, "multi-GPU").
// librccl is reached through dlopen: the library carries no link-time dependency on it (573 MB), and a single-GPU
// process never loads it.
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "host.h"

struct cct_unique_id_t { char internal[CCT_COMM_ID_BYTES]; };  // ncclUniqueId: passed to ncclCommInitRank by value

namespace cct {
namespace {

struct Rccl {
	void *h = nullptr;
	int (*GetUniqueId)(void *) = nullptr;
	int (*CommInitRank)(void **, int, cct_unique_id_t, int) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
	int (*CommDestroy)(void *) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	void *comm = nullptr;
	int rank = -1, world = 0;
	DevBuf d_send, d_recv;
	// the collective has a stream and a mutex of its own: on the main stream the gather of a few hundred sizes would queue
	// behind the ten milliseconds of kernels of the encode batch in slot 0, once per step
	hipStream_t stream = nullptr;
	std::mutex mu;
} g_rccl;

int rccl_load()
{
	if (g_rccl.h) return CCT_OK;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
	if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h) return fail(CCT_E_DEVICE, "cannot load librccl: %s", dlerror());
	g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
	g_rccl.CommInitRank = (int (*)(void **, int, cct_unique_id_t, int))dlsym(h, "ncclCommInitRank");
	g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
	g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
	g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
	if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) { dlclose(h); return fail(CCT_E_DEVICE, "librccl lacks an expected symbol"); }
	g_rccl.h = h;
	return CCT_OK;
}
int rccl_fail(const char *what, int rc)
{
	return fail(CCT_E_DEVICE, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}

}  // namespace

void comm_release()
{
	std::lock_guard<std::mutex> lkc(g_rccl.mu);
	if (g_rccl.comm) { (void)g_rccl.CommDestroy(g_rccl.comm); g_rccl.comm = nullptr; g_rccl.rank = -1; g_rccl.world = 0; }
	g_rccl.d_send.release(); g_rccl.d_recv.release();
	if (g_rccl.stream) { (void)hipStreamDestroy(g_rccl.stream); g_rccl.stream = nullptr; }
}

}  // namespace cct

using namespace cct;

extern "C" {

int cct_comm_unique_id(void *id128)
{
	if (!id128) return fail(CCT_E_ARG, "null id");
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc || (rc = rccl_load())) return rc;
	cct_unique_id_t id;
	const int r = g_rccl.GetUniqueId(&id);
	if (r) return rccl_fail("ncclGetUniqueId", r);
	memcpy(id128, &id, CCT_COMM_ID_BYTES);
	return CCT_OK;
}

int cct_comm_init(const void *id128, int rank, int world)
{
	if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(CCT_E_ARG, "bad communicator arguments");
	std::lock_guard<std::mutex> lk(g_mu);
	int rc = ensure_ctx();
	if (rc || (rc = rccl_load())) return rc;
	if (g_rccl.comm) return fail(CCT_E_ARG, "a communicator already exists");
	cct_unique_id_t id;
	memcpy(&id, id128, CCT_COMM_ID_BYTES);
	std::lock_guard<std::mutex> lkc(g_rccl.mu);
	// stream creation and the communicator's own allocations: with nothing else of the library in flight (host.h)
	return exclusive_section([&]() -> int {
		if (!g_rccl.stream) HIP_TRY(hipStreamCreateWithFlags(&g_rccl.stream, hipStreamNonBlocking));
		const int r = g_rccl.CommInitRank(&g_rccl.comm, world, id, rank);
		if (r) { g_rccl.comm = nullptr; return rccl_fail("ncclCommInitRank", r); }
		g_rccl.rank = rank; g_rccl.world = world;
		return CCT_OK;
	});
}

int cct_comm_info(int *rank, int *world)
{
	std::lock_guard<std::mutex> lk(g_rccl.mu);
	if (rank) *rank = g_rccl.comm ? g_rccl.rank : -1;
	if (world) *world = g_rccl.comm ? g_rccl.world : 0;
	return CCT_OK;
}

int cct_allgather_u32(const uint32_t *h_local, int n_local, int max_local, uint32_t *h_all)
{
	if (n_local < 0 || max_local < n_local || (n_local && !h_local) || !h_all) return fail(CCT_E_ARG, "bad all-gather arguments");
	std::lock_guard<std::mutex> lk(g_rccl.mu);
	if (!g_rccl.comm) {
		for (int i = 0; i < max_local; i++) h_all[i] = i < n_local ? h_local[i] : 0u;
		return CCT_OK;
	}
	if (forked_after_init()) return fail(CCT_E_DEVICE, "this process was forked after the communicator was created");
	ApiCall in_call;  // a call in flight like the others: captures and allocations of other threads wait for it (host.h)
	HIP_TRY(hipSetDevice(bound_device()));
	int rc;
	const size_t bytes = (size_t)std::max(max_local, 1) * 4;
	if ((rc = g_rccl.d_send.ensure(bytes)) || (rc = g_rccl.d_recv.ensure(bytes * g_rccl.world))) return rc;
	hipStream_t st = g_rccl.stream;
	HIP_TRY(hipMemsetAsync(g_rccl.d_send.p, 0, bytes, st));
	if (n_local) HIP_TRY(hipMemcpyAsync(g_rccl.d_send.p, h_local, (size_t)n_local * 4, hipMemcpyHostToDevice, st));
	const int r = g_rccl.AllGather(g_rccl.d_send.p, g_rccl.d_recv.p, (size_t)max_local, 3 /* ncclUint32 */, g_rccl.comm, st);
	if (r) return rccl_fail("ncclAllGather", r);
	HIP_TRY(hipMemcpyAsync(h_all, g_rccl.d_recv.p, (size_t)max_local * 4 * g_rccl.world, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return CCT_OK;
}

int cct_comm_destroy(void)
{
	std::lock_guard<std::mutex> lk(g_rccl.mu);
	if (g_rccl.comm) { (void)g_rccl.CommDestroy(g_rccl.comm); g_rccl.comm = nullptr; }
	g_rccl.rank = -1; g_rccl.world = 0;
	return exclusive_section([&]() -> int {
		g_rccl.d_send.release(); g_rccl.d_recv.release();
		if (g_rccl.stream) { (void)hipStreamDestroy(g_rccl.stream); g_rccl.stream = nullptr; }
		return CCT_OK;
	});
}

}  // extern "C"
