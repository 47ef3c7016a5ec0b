// tools/readbw.hip -- what a read-once stream can reach on this part: register loads (the decimator's current
// load path) vs LDS-DMA (global_load_lds_dwordx4) with and without the nt hint.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/readbw tools/readbw.hip && /tmp/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SPAN 16384            // dwords per workgroup (64 KiB), like k_fm_decimate

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_reg(const u32x4 *__restrict__ src, unsigned *__restrict__ out)
{
	const u32x4 *p = src + (size_t)blockIdx.x * (SPAN / 4) + threadIdx.x;
	unsigned acc = 0;
	u32x4 v[INFLIGHT];
#pragma unroll
	for (int t = 0; t < 16; t += INFLIGHT) {
#pragma unroll
		for (int u = 0; u < INFLIGHT; u++)
			v[u] = __builtin_nontemporal_load(p + (t + u) * 256);
#pragma unroll
		for (int u = 0; u < INFLIGHT; u++)
			acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u)
		out[blockIdx.x] = acc;
}

// register loads plus WORK rounds of 8 independent VALU operations per 16-byte load (how much arithmetic the stream tolerates)
template <int WORK, bool DPP>
__global__ __launch_bounds__(256) void k_work(const u32x4 *__restrict__ src, unsigned *__restrict__ out)
{
	const u32x4 *p = src + (size_t)blockIdx.x * (SPAN / 4) + threadIdx.x;
	unsigned acc = 0;
#pragma unroll
	for (int h = 0; h < 2; h++) {
		u32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++)
			v[u] = __builtin_nontemporal_load(p + (h * 8 + u) * 256);
#pragma unroll
		for (int u = 0; u < 8; u++) {
			unsigned a = v[u].x, b = v[u].y, c = v[u].z, d = v[u].w;
#pragma unroll
			for (int w = 0; w < WORK; w++) {
				a = a * 3u + b; b = b ^ (c >> 3); c = c + (d << 2); d = d - a;
				a ^= c; b += d; c ^= 0x55aau + w; d += 77u;
			}
			unsigned t = a ^ b ^ c ^ d;
			if (DPP) {
#pragma unroll
				for (int k = 0; k < 6; k++)
					t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xf, 0xf, false);
			}
			acc += t;
		}
	}
	if (acc == 0x12345678u)
		out[blockIdx.x] = acc;
}

// each wave streams its 16 KiB quarter of the span through a ring of RING 1-KiB LDS slots
template <int RING, int AUX>
__global__ __launch_bounds__(256) void k_lds(const u32x4 *__restrict__ src, unsigned *__restrict__ out)
{
	__shared__ u32x4 ring[4][RING][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const u32x4 *p = src + (size_t)blockIdx.x * (SPAN / 4) + wave * 1024 + lane;     // 16 tiles of 64 x 16 B
	unsigned acc = 0;
#pragma unroll
	for (int t = 0; t < RING; t++)
		__builtin_amdgcn_global_load_lds((const void *)(p + t * 64), (__attribute__((address_space(3))) void *)&ring[wave][t][0], 16, 0, AUX);
#pragma unroll
	for (int t = 0; t < 16; t++) {
		if (16 - t > RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 1) : "memory");
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const u32x4 v = ring[wave][t % RING][lane];
		acc += v.x ^ v.y ^ v.z ^ v.w;
		if (t + RING < 16) {
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot's read is done before it is refilled
			__builtin_amdgcn_global_load_lds((const void *)(p + (t + RING) * 64), (__attribute__((address_space(3))) void *)&ring[wave][t % RING][0], 16, 0, AUX);
		}
	}
	if (acc == 0x12345678u)
		out[blockIdx.x] = acc;
}

__global__ void k_fill(unsigned *p, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		unsigned x = (unsigned)i * 2654435761u + 12345u;
		x ^= x << 13; x ^= x >> 17; x ^= x << 5;
		p[i] = x;
	}
}

template <typename F>
static void run(const char *name, F launch, size_t bytes)
{
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 5; i++) launch();
	hipDeviceSynchronize();
	const int reps = 40;
	hipEventRecord(a);
	for (int i = 0; i < reps; i++) launch();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	printf("%-28s %8.1f us/launch  %7.1f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char **argv)
{
	const size_t bytes = 4ull << 30;
	u32x4 *src; unsigned *out;
	hipMalloc(&src, bytes);
	if (argc > 1) { hipMemset(src, 1, bytes); printf("constant data\n"); }
	else { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)src, bytes / 4); printf("random data\n"); }
	const unsigned grid = (unsigned)(bytes / (SPAN * 4));
	hipMalloc(&out, grid * 4);
	run("reg nt x8", [&] { hipLaunchKernelGGL(k_reg<8>, dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg nt x16", [&] { hipLaunchKernelGGL(k_reg<16>, dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 8 valu/load", [&] { hipLaunchKernelGGL((k_work<1, false>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 24 valu/load", [&] { hipLaunchKernelGGL((k_work<3, false>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 48 valu/load", [&] { hipLaunchKernelGGL((k_work<6, false>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 72 valu/load", [&] { hipLaunchKernelGGL((k_work<9, false>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 96 valu/load", [&] { hipLaunchKernelGGL((k_work<12, false>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 24 valu + 6 dpp", [&] { hipLaunchKernelGGL((k_work<3, true>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("reg x8 + 48 valu + 6 dpp", [&] { hipLaunchKernelGGL((k_work<6, true>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("lds-dma ring8", [&] { hipLaunchKernelGGL((k_lds<8, 0>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("lds-dma ring8 nt", [&] { hipLaunchKernelGGL((k_lds<8, 2>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("lds-dma ring4 nt", [&] { hipLaunchKernelGGL((k_lds<4, 2>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	run("lds-dma ring16 nt", [&] { hipLaunchKernelGGL((k_lds<16, 2>), dim3(grid), dim3(256), 0, 0, src, out); }, bytes);
	if (hipGetLastError() != hipSuccess) { printf("error\n"); return 1; }
	return 0;
}
