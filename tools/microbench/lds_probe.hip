// Hardware probes behind the design of csrc/encode_pipe.hip (run on the GPU box: tools/microbench/run.sh).
//   1. do ds_write_b32 / ds_write_b64 at byte addresses that are not multiples of 4 store the bytes at that address?
//   2. wave-instruction rates of ds_write_b8, ds_write_b32, ds_or_b32, ds_write_b64 (all lanes, conflict-free)
//   3. HBM read rate: 16 B per lane in raster order against 8 B per lane, four rows per 4x4-pixel block with the
//      lanes walking the blocks of a 64x64 tile in Z order (half a 128-byte line per row and wave)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define LDS(T) __attribute__((address_space(3))) T

__global__ void unaligned_kernel(uint8_t *out)
{
	__shared__ __attribute__((aligned(16))) uint8_t buf[256];
	const int t = threadIdx.x;
	for (int i = t; i < 256; i += 64) buf[i] = 0;
	__syncthreads();
	if (t < 4) {
		// lane t writes 0xA0+t.. pattern at byte address 16*t + t (misaligned by t)
		const uint32_t addr = (uint32_t)(uintptr_t)(LDS(uint8_t) *)buf + 16 * t + t;
		const uint32_t v = 0x04030201u + 0x10101010u * t;
		asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
	} else if (t < 8) {
		const int u = t - 4;
		const uint32_t addr = (uint32_t)(uintptr_t)(LDS(uint8_t) *)buf + 64 + 16 * u + u;
		const uint64_t v = 0x0807060504030201ull + 0x1010101010101010ull * u;
		asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
	}
	__syncthreads();
	for (int i = t; i < 256; i += 64) out[i] = buf[i];
}

template <int MODE>
__global__ void __launch_bounds__(256) lds_rate_kernel(uint32_t *out, int iters)
{
	__shared__ __attribute__((aligned(16))) uint32_t buf[4096];
	const int t = threadIdx.x;
	for (int i = t; i < 4096; i += 256) buf[i] = 0;
	__syncthreads();
	const uint32_t base = (uint32_t)(uintptr_t)(LDS(uint32_t) *)buf;
	uint32_t v = t * 2654435761u;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 16; u++) {
			if (MODE == 0) {  // bytes, consecutive lanes consecutive bytes
				const uint32_t a = base + t + u * 256;
				asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory");
			} else if (MODE == 1) {
				const uint32_t a = base + t * 4 + u * 1024;
				asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
			} else if (MODE == 2) {
				const uint32_t a = base + t * 4 + u * 1024;
				asm volatile("ds_or_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
			} else if (MODE == 3) {
				const uint32_t a = base + t * 8 + (u & 7) * 2048;
				const uint64_t vv = ((uint64_t)v << 32) | v;
				asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(vv) : "memory");
			} else if (MODE == 4) {  // unaligned b32: lane stride 5 bytes
				const uint32_t a = base + t * 5 + u * 1280 + 1;
				asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
			} else if (MODE == 5) {  // bytes at stride 17 (what token bytes of 16-pixel blocks look like)
				const uint32_t a = base + t * 17 + u;
				asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory");
			}
			v += 0x01010101u;
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	__syncthreads();
	if (t == 0) out[blockIdx.x] = buf[1] + buf[100];
}

// 16 B per lane, raster order
__global__ void __launch_bounds__(256) read_raster_kernel(const uint4 *in, size_t n16, uint32_t *out)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		const uint4 v = in[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// one workgroup per 64x64 tile of 512x512 slices; lane = 4x4 block in Z order; four 8-byte row loads
__global__ void __launch_bounds__(256) read_blocks_kernel(const uint16_t *in, int n_tiles_total, uint32_t *out)
{
	uint32_t acc = 0;
	for (int tile = blockIdx.x; tile < n_tiles_total; tile += gridDim.x) {
		const int sl = tile >> 6, ti = tile & 63;
		const int ty = ti >> 3, tx = ti & 7;
		const int k = threadIdx.x;
		int bx = 0, by = 0;
		for (int b = 0; b < 4; b++) { bx |= ((k >> (2 * b)) & 1) << b; by |= ((k >> (2 * b + 1)) & 1) << b; }
		const uint16_t *p = in + (size_t)sl * 262144 + (size_t)(ty * 64 + by * 4) * 512 + tx * 64 + bx * 4;
		const uint2 r0 = *reinterpret_cast<const uint2 *>(p);
		const uint2 r1 = *reinterpret_cast<const uint2 *>(p + 512);
		const uint2 r2 = *reinterpret_cast<const uint2 *>(p + 1024);
		const uint2 r3 = *reinterpret_cast<const uint2 *>(p + 1536);
		acc += r0.x ^ r0.y ^ r1.x ^ r1.y ^ r2.x ^ r2.y ^ r3.x ^ r3.y;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// one wave per 64x32-pixel half tile of a 512x512 slice: lane = 8x4 pixels, four 16-byte row loads (eight 128-byte
// lines per wave-instruction, 1 KB apart) -- the front end of encode_pipe.hip
__global__ void __launch_bounds__(64) read_halftiles_kernel(const uint16_t *in, uint32_t *out, int rows_per_lane)
{
	const int ht = blockIdx.x & 127, sl = blockIdx.x >> 7;
	const int tile = ht >> 1, half = ht & 1;
	const int ty = tile >> 3, tx = tile & 7;
	const int lane = threadIdx.x;
	const uint16_t *p = in + (size_t)sl * 262144 + (size_t)(ty * 64 + half * 32 + (lane >> 3) * 4) * 512 + tx * 64 + (lane & 7) * 8;
	uint32_t acc = 0;
	for (int q = 0; q < rows_per_lane; q++) {
		const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)q * 512);
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
// the same bytes, but a 256-lane workgroup reads 64 pixel rows x 64 pixels as whole 128-byte lines of ONE tile, four
// lanes... variant B: workgroup = a band of 4 image rows x 512 pixels: every wave-instruction reads 1 KB contiguous
__global__ void __launch_bounds__(256) read_bands_kernel(const uint16_t *in, uint32_t *out)
{
	// band = 16 rows of one slice (16 KB contiguous): lane reads 4 x 16 bytes at 4 KB stride
	const size_t base = (size_t)blockIdx.x * 8192;  // pixels
	uint32_t acc = 0;
	for (int q = 0; q < 4; q++) {
		const uint4 v = *reinterpret_cast<const uint4 *>(in + base + (size_t)q * 2048 + threadIdx.x * 8);
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F f, int reps)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	f();
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int i = 0; i < reps; i++) f();
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	return ms / reps;
}

int main()
{
	uint8_t *d_out;
	CK(hipMalloc(&d_out, 256));
	hipLaunchKernelGGL(unaligned_kernel, dim3(1), dim3(64), 0, 0, d_out);
	CK(hipDeviceSynchronize());
	uint8_t h[256];
	CK(hipMemcpy(h, d_out, 256, hipMemcpyDeviceToHost));
	printf("unaligned ds_write_b32 (lane t at 16t+t):\n");
	for (int t = 0; t < 4; t++) { printf("  t=%d:", t); for (int i = 0; i < 16; i++) printf(" %02x", h[16 * t + i]); printf("\n"); }
	printf("unaligned ds_write_b64 (lane u at 64+16u+u):\n");
	for (int t = 0; t < 4; t++) { printf("  u=%d:", t); for (int i = 0; i < 16; i++) printf(" %02x", h[64 + 16 * t + i]); printf("\n"); }

	uint32_t *d_o32;
	CK(hipMalloc(&d_o32, 4096 * 4));
	const int iters = 2000, grid = 256 * 4;
	const char *names[6] = {"ds_write_b8 (dense)", "ds_write_b32", "ds_or_b32", "ds_write_b64", "ds_write_b32 unaligned stride 5", "ds_write_b8 stride 17"};
	float ms[6];
	ms[0] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<0>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	ms[1] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<1>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	ms[2] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<2>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	ms[3] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<3>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	ms[4] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<4>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	ms[5] = time_ms([&] { hipLaunchKernelGGL(lds_rate_kernel<5>, dim3(grid), dim3(256), 0, 0, d_o32, iters); }, 3);
	for (int m = 0; m < 6; m++) {
		// per CU: 4 workgroups x 4 waves x iters x 16 wave-instructions
		const double winst = 4.0 * 4 * iters * 16;
		printf("%-34s %8.3f ms  -> %.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", names[m], ms[m],
		       ms[m] * 1e6 / winst, ms[m] * 1e6 / winst * 2.4);
	}

	const size_t bytes = (size_t)3 * 256 * 512 * 512 * 2;  // 402 MB > Infinity Cache
	uint16_t *d_img;
	CK(hipMalloc(&d_img, bytes));
	CK(hipMemset(d_img, 1, bytes));
	const float t_r = time_ms([&] { hipLaunchKernelGGL(read_raster_kernel, dim3(2048), dim3(256), 0, 0, (const uint4 *)d_img, bytes / 16, d_o32); }, 10);
	const float t_b = time_ms([&] { hipLaunchKernelGGL(read_blocks_kernel, dim3(8192), dim3(256), 0, 0, d_img, 3 * 256 * 64, d_o32); }, 10);
	const float t_b2 = time_ms([&] { hipLaunchKernelGGL(read_blocks_kernel, dim3(3 * 256 * 64), dim3(256), 0, 0, d_img, 3 * 256 * 64, d_o32); }, 10);
	const float t_h = time_ms([&] { hipLaunchKernelGGL(read_halftiles_kernel, dim3(3 * 256 * 128), dim3(64), 0, 0, d_img, d_o32, 4); }, 10);
	const float t_bd = time_ms([&] { hipLaunchKernelGGL(read_bands_kernel, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, d_img, d_o32); }, 10);
	printf("read 402 MB: half-tile waves (8 x 128-byte lines per instruction) %.3f ms = %.0f GB/s; 16-row bands (1 KB contiguous per instruction) %.3f ms = %.0f GB/s\n",
	       t_h, bytes / t_h * 1e-6, t_bd, bytes / t_bd * 1e-6);
	printf("read 402 MB: raster 16 B/lane %.3f ms = %.0f GB/s; 4x4 blocks 8 B/lane %.3f ms = %.0f GB/s (grid 8192), %.3f ms = %.0f GB/s (one tile per workgroup)\n",
	       t_r, bytes / t_r * 1e-6, t_b, bytes / t_b * 1e-6, t_b2, bytes / t_b2 * 1e-6);
	return 0;
}
