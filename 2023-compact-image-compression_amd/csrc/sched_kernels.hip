// Device-side gate between the decode stream and the encode stream (api.cpp, cct_decode_batch).
//
// In a pipelined job the decode of batch k runs on its own stream next to the encode of batch k+1, with slack: the step is as
// long as the encode chain.  Which encode kernels it runs next to decides what the overlap costs.  Measured with the kernel
// trace of the bench (profiles/r03_bench_timeline.log): the INFLATE kernel keeps 118 KB of LDS per CU, so a transform+pack
// kernel (41 KB per workgroup, four to a CU) beside it ran 0.37 instead of 0.15 ms; the tree kernel's serial waves ran 1.14 to
// 1.59 instead of 0.73 ms beside the INFLATE and decode kernels, the Adler kernel 118 instead of 23 us; both sort passes
// and the match kernel lose 0 to 10 %.  So the decode kernels of a call are released when the NEXT transform+pack stage has
// ended -- they then run next to the sort and match kernels of that batch and are through before its tree kernel starts.
// "Next" is a launch that may not have happened when the decode call is made, so the gate is a counter in device memory:
//   gate[0]  transform+pack stages that have ended      gate[1]  DEFLATE passes that have ended
// bumped by one-lane kernels on the encode stream, waited for by a one-wave kernel on the decode stream.  When every
// pass that had been issued at the time of the decode call has ended and no new stage follows within `grace`, the wait
// gives up (the end of a pipeline; a lone decode next to a lone encode); `timeout` bounds it in any case.
#include <hip/hip_runtime.h>

#include "cct_internal.h"

namespace cct {
namespace {

__global__ void gate_bump_kernel(uint32_t *word)
{
	if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ticks of the constant 100 MHz counter (s_memrealtime)
__global__ void gate_wait_kernel(const uint32_t *gate, uint32_t want_pass, uint32_t grace, uint32_t timeout)
{
	if (threadIdx.x != 0) return;
	const uint64_t t0 = wall_clock64();
	// the next stage to END after this kernel has started (the host may have queued it long ago: encode calls queue ahead)
	const uint32_t want_stage = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
	uint64_t t_pass = 0;
	for (;;) {
		const uint32_t stage = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((int32_t)(stage - want_stage) >= 0) break;
		const uint64_t now = wall_clock64();
		if (now - t0 > timeout) break;
		const uint32_t pass = __hip_atomic_load(gate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((int32_t)(pass - want_pass) >= 0) {
			if (!t_pass) t_pass = now;
			else if (now - t_pass > grace) break;
		}
		__builtin_amdgcn_s_sleep(64);
	}
}

}  // namespace

hipError_t launch_gate_bump(uint32_t *word, hipStream_t st)
{
	hipLaunchKernelGGL(gate_bump_kernel, dim3(1), dim3(64), 0, st, word);
	return hipGetLastError();
}

hipError_t launch_gate_wait(const uint32_t *gate, uint32_t want_pass, uint32_t grace_us, uint32_t timeout_us, hipStream_t st)
{
	hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, st, gate, want_pass, grace_us * 100u, timeout_us * 100u);
	return hipGetLastError();
}

}  // namespace cct
