// Hardware probe behind csrc/inflate_kernels.hip: what does one step of a dependent LDS table-lookup chain cost a wave
// (one wave per SIMD, as in the INFLATE workgroup)?
//   mode 0: idx = tab[idx & 2047]                         (pure chase, random table)
//   mode 1: the same + an independent read whose lanes are 8 dwords apart (the per-lane input cursor of 256-bit segments)
//   mode 2: mode 0 + 20 dependent VALU operations
//   mode 3: mode 0 + a 64-bit variable shift pair per step
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) chain_kernel(const uint32_t *init, uint32_t *out, long long *cycles, int steps, int stride)
{
	__shared__ uint32_t tab[2048];
	__shared__ uint32_t in32[4096];
	for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = init[i];
	for (int i = threadIdx.x; i < 4096; i += 256) in32[i] = init[2048 + (i & 2047)];
	__syncthreads();
	uint32_t idx = threadIdx.x * 2654435761u, acc = 0, nxt = threadIdx.x * stride;
	uint64_t buf = idx;
	const long long t0 = clock64();
	for (int s = 0; s < steps; s++) {
		uint32_t w = 0;
		if (MODE == 1) w = in32[nxt & 4095];
		uint32_t e = tab[idx & 2047];
		if (MODE == 1) { acc += w; nxt += (e & 1); }
		if (MODE == 2) {
#pragma unroll
			for (int k = 0; k < 20; k++) e = (e ^ (e >> 3)) + k;
		}
		if (MODE == 3) { buf = (buf >> (e & 15)) | ((uint64_t)e << 40); e ^= (uint32_t)buf; }
		idx = e;
	}
	const long long t1 = clock64();
	out[blockIdx.x * 256 + threadIdx.x] = idx + acc + (uint32_t)buf;
	if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
	std::vector<uint32_t> h(4096);
	uint32_t x = 12345;
	for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x >> 7; }
	uint32_t *d_init, *d_out; long long *d_cyc;
	CK(hipMalloc(&d_init, 4096 * 4)); CK(hipMalloc(&d_out, 256 * 256 * 4)); CK(hipMalloc(&d_cyc, 256 * 8));
	CK(hipMemcpy(d_init, h.data(), 4096 * 4, hipMemcpyHostToDevice));
	const int steps = 2000;
	for (int mode = 0; mode < 5; mode++) {
		for (int rep = 0; rep < 2; rep++) {
			const int stride = mode == 4 ? 1 : 8;
			if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(256), dim3(256), 0, 0, d_init, d_out, d_cyc, steps, stride);
			if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(256), dim3(256), 0, 0, d_init, d_out, d_cyc, steps, stride);
			if (mode == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(256), dim3(256), 0, 0, d_init, d_out, d_cyc, steps, stride);
			if (mode == 3) hipLaunchKernelGGL(chain_kernel<3>, dim3(256), dim3(256), 0, 0, d_init, d_out, d_cyc, steps, stride);
			if (mode == 4) hipLaunchKernelGGL(chain_kernel<1>, dim3(256), dim3(256), 0, 0, d_init, d_out, d_cyc, steps, stride);
			CK(hipDeviceSynchronize());
		}
		long long c[256];
		CK(hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost));
		double s = 0; for (long long v : c) s += (double)v;
		printf("mode %d: %.1f cycles per step (256 workgroups of 256 lanes)\n", mode, s / 256 / steps);
	}
	return 0;
}
