// tools/rwmix.hip -- what HBM gives a kernel that READS a stream and WRITES a fraction of it back: the ceiling of the rx_fm chains
// that are bound by the bytes they move.  For read:write ratios from pure read to 1:1, contiguous non-temporal
// 16-byte loads (the decimators' pattern) and either contiguous 16-byte stores or 16-byte pieces 1 KiB apart (what the tiled pcm
// layout receives from one wave), and the register cascade's own shape (4 KiB in, a 236-byte run of dwords out per wave).  Ratios below 16
// only: with fewer than 256 output vectors per workgroup some threads' loads would be dead code.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/rwmix.hip && /tmp/rwmix [json-out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a workgroup reads 64 KiB (16 vectors per thread, all in flight by eights) and writes 64 KiB / RATIO; RATIO == 0: no writes.
// SCATTER: the 16-byte pieces of a wave's output go 1 KiB apart instead of side by side.
template <int RATIO, bool SCATTER>
__global__ __launch_bounds__(256) void k_rw(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n_wg)
{
	const unsigned per = gridDim.x >> 3;                      // XCD-contiguous order like the product's kernels
	const size_t wg = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
	if (wg >= n_wg)
		return;
	const u32x4 *p = src + wg * 4096 + threadIdx.x;
	u32x4 acc = (u32x4)(0u);
#pragma unroll
	for (int h = 0; h < 2; h++) {
		u32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++)
			v[u] = __builtin_nontemporal_load(p + (h * 8 + u) * 256);
#pragma unroll
		for (int u = 0; u < 8; u++)
			acc += v[u];
	}
	if constexpr (RATIO > 0) {
		// 4096 / RATIO vectors out per workgroup: thread t writes vector t (and t + 256, ...) while t < 4096 / RATIO
		constexpr int OUT = 4096 / RATIO;
#pragma unroll
		for (int k = 0; k < (OUT + 255) / 256; k++) {
			const int t = threadIdx.x + 256 * k;
			if (t < OUT) {
				size_t o = wg * OUT + t;
				if (SCATTER)                                      // piece i of a 64-piece group lands 64 pieces (1 KiB) apart: (i % 64) * 64 + i / 64 within 4096
					o = (o & ~(size_t)4095) | ((o & 63) << 6) | ((o >> 6) & 63);
				__builtin_nontemporal_store(acc + (u32x4)(k), dst + o);
			}
		}
	} else {
		if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
			dst[wg] = acc;
	}
}


// the register cascade's shape (k_fm_fifth_regn<., 4>): a wave reads 4 KiB and stores ONE DWORD PER LANE from 59 lanes -- 236-byte runs laid end to
// end, so that neighbouring waves complete each other's lines
__global__ __launch_bounds__(256) void k_rw_runs(const u32x4 *__restrict__ src, unsigned *__restrict__ dst, size_t n_waves)
{
	const unsigned per = gridDim.x >> 3;
	const size_t wg = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
	const size_t wave = wg * 4 + (threadIdx.x >> 6);
	const unsigned lane = threadIdx.x & 63;
	if (wave >= n_waves)
		return;
	const u32x4 *p = src + wave * 256 + lane;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
		v[u] = __builtin_nontemporal_load(p + u * 64);
	const u32x4 acc = v[0] + v[1] + v[2] + v[3];
	unsigned r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	r ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x138, 0xf, 0xf, true);      // every lane's loads feed a stored value
	if (lane >= 5)
		__builtin_nontemporal_store(r, dst + wave * 59 + (lane - 5));
}

__global__ void k_fill(unsigned *p, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		unsigned x = (unsigned)i * 2654435761u + 12345u;
		x ^= x << 13; x ^= x >> 17; x ^= x << 5;
		p[i] = x;
	}
}

int main(int argc, char **argv)
{
	const size_t bytes = (size_t)4 << 30, n_wg = bytes / 65536;
	u32x4 *src, *dst;
	CHK(hipMalloc(&src, bytes));
	CHK(hipMalloc(&dst, bytes + (1 << 20)));
	hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)src, bytes / 4);
	CHK(hipDeviceSynchronize());
	const unsigned grid = (unsigned)((n_wg + 7) / 8 * 8);
	hipEvent_t a, b;
	CHK(hipEventCreate(&a));
	CHK(hipEventCreate(&b));
	std::string json = "{\"unit\": \"TB/s of read + written bytes, 4 GiB read per launch\", \"legs\": {";
	bool first = true;
	auto run = [&](const char *name, auto launch, double wbytes) {
		for (int i = 0; i < 3; i++) launch();
		CHK(hipDeviceSynchronize());
		const int reps = 20;
		CHK(hipEventRecord(a));
		for (int i = 0; i < reps; i++) launch();
		CHK(hipEventRecord(b));
		CHK(hipEventSynchronize(b));
		float ms = 0;
		CHK(hipEventElapsedTime(&ms, a, b));
		const double t = ms * 1e-3 / reps, tot = ((double)bytes + wbytes) / t / 1e12, rd = (double)bytes / t / 1e12;
		printf("%-34s %7.3f ms  total %5.2f TB/s  (read stream alone %5.2f TB/s)\n", name, t * 1e3, tot, rd);
		char buf[200];
		snprintf(buf, sizeof(buf), "%s\"%s\": {\"ms\": %.4f, \"total_TBs\": %.3f, \"read_TBs\": %.3f}", first ? "" : ", ", name, t * 1e3, tot, rd);
		json += buf;
		first = false;
	};
#define LEG(NAME, R, S) run(NAME, [&] { hipLaunchKernelGGL((k_rw<R, S>), dim3(grid), dim3(256), 0, 0, src, dst, n_wg); }, (R) ? (double)bytes / (R) : 0.0)
	LEG("read only", 0, false);
	LEG("read:write 16:1 contiguous", 16, false);
	LEG("read:write 16:1 scattered 16 B", 16, true);
	LEG("read:write 8:1 contiguous", 8, false);
	LEG("read:write 8:1 scattered 16 B", 8, true);
	LEG("read:write 4:1 contiguous", 4, false);
	LEG("read:write 2:1 contiguous", 2, false);
	LEG("read:write 1:1 contiguous (copy)", 1, false);
	{
		const size_t n_waves = bytes / 4096;
		const unsigned g2 = (unsigned)(((n_waves + 3) / 4 + 7) / 8 * 8);
		run("read:write 17:1, 236-byte runs of dwords", [&] { hipLaunchKernelGGL(k_rw_runs, dim3(g2), dim3(256), 0, 0, src, (unsigned *)dst, n_waves); }, (double)n_waves * 236);
	}
	json += "}}";
	if (argc > 1) {
		FILE *f = fopen(argv[1], "w");
		if (f) { fputs(json.c_str(), f); fputc('\n', f); fclose(f); }
	}
	return 0;
}
