// sdr_kernels.hip -- rx_sdr's output converters (rtl_sdr.c:354-391) as device code.
//
// Four element-wise maps over a captured stream, all HBM-bound: every input byte is read once with
// 16-byte loads, every output byte written once with 16-byte stores.
//   CS16 -> CS8   (int8)(x / 32767.0 * 128.0 + 0.4)            rtl_sdr.c:368-370 (fp64 in the reference)
//   CS16 -> CU8   (uint8)(x / 32767.0 * 128.0 + 127.4)         rtl_sdr.c:376-378
//   CS16 -> CF32  x * 1.0f / SHRT_MAX                          rtl_sdr.c:384-386
//   CS12 -> CS16  three packed bytes -> two left-aligned int16 rtl_sdr.c:356-363
// The two 8-bit maps share rx_fm's callback scaling (rtl_fm.c:845-848): one fp32 fma reproduces the fp64
// expression for every int16 input (checked exhaustively in tests/test_gpu_sdr.py).  CS8's value 128
// (x >= 32665) does not fit an int8; the reference build wraps it to -128 and so does the byte store here.
// CU8 adds 127.4 instead of 0.4; the truncation toward zero differs from a floor only where the CS8 sum is
// negative (x <= -103), and the one input below zero (x = -32768 -> -0.6) truncates to 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// x * 1.0f / 32767.0f, correctly rounded like the reference's fp32 division, in three operations: the quotient by
// the rounded reciprocal, its exact residual, one correction (Markstein); all 65536 inputs are checked in
// tests/test_gpu_sdr.py::test_every_int16_matches_reference
__device__ __forceinline__ float sdr_unit(int x)
{
	const float d = 32767.0f, r = 1.0f / 32767.0f;
	const float xf = (float)x;
	const float q = xf * r;
	const float e = __builtin_fmaf(-q, d, xf);
	return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ int sdr_scale(int x)
{
	return (int)__builtin_fmaf((float)x, (float)(128.0 / 32767.0), 0.4f);
}

template <bool UNSIGNED>
__device__ __forceinline__ uint32_t sdr_to8(int x)
{
	int v = sdr_scale(x);
	if (UNSIGNED) {
		v += (x <= -103) ? 126 : 127;
		v = v < 0 ? 0 : v;
	}
	return (uint32_t)v & 0xffu;
}

template <bool UNSIGNED>
__device__ __forceinline__ uint32_t sdr_pack4(uint32_t a, uint32_t b)
{
	return sdr_to8<UNSIGNED>((int)(short)(a & 0xffffu)) | (sdr_to8<UNSIGNED>((int)a >> 16) << 8) |
	       (sdr_to8<UNSIGNED>((int)(short)(b & 0xffffu)) << 16) | (sdr_to8<UNSIGNED>((int)b >> 16) << 24);
}

// Every kernel below gives consecutive lanes consecutive pieces of BOTH streams: a wave instruction then reads or writes one
// contiguous run (1 KiB for 16-byte pieces).  The first versions gave a thread 32-64 contiguous bytes of its own, i.e. two to four
// instructions each touching every other (third, fourth) 16-byte piece of a 2-4 KiB window: 3.3-5.3 TB/s against a copy
// ceiling of 6.3.

// n16 int16 in, n16 bytes out; one thread per 8 values (16 bytes in, 8 bytes out), two units in flight
template <bool UNSIGNED>
__global__ __launch_bounds__(256) void k_sdr_cs16_to_8(const int16_t *__restrict__ in, u64 n16, uint8_t *__restrict__ out)
{
	const u64 units = n16 >> 3;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += 2 * stride) {
		const u64 u2 = u + stride;
		const bool two = u2 < units;
		const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + u);
		const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + (two ? u2 : u));
		const u32x2 oa = {sdr_pack4<UNSIGNED>(a.x, a.y), sdr_pack4<UNSIGNED>(a.z, a.w)};
		const u32x2 ob = {sdr_pack4<UNSIGNED>(b.x, b.y), sdr_pack4<UNSIGNED>(b.z, b.w)};
		__builtin_nontemporal_store(oa, reinterpret_cast<u32x2 *>(out) + u);
		if (two)
			__builtin_nontemporal_store(ob, reinterpret_cast<u32x2 *>(out) + u2);
	}
	if (blockIdx.x == 0 && threadIdx.x < (n16 & 7)) {                  // ragged end
		const u64 i = (units << 3) + threadIdx.x;
		out[i] = (uint8_t)sdr_to8<UNSIGNED>(in[i]);
	}
}

// n16 int16 in, n16 floats out; one thread per 4 values (8 bytes in, 16 bytes out), two units in flight
__global__ __launch_bounds__(256) void k_sdr_cs16_to_cf32(const int16_t *__restrict__ in, u64 n16, float *__restrict__ out)
{
	const u64 units = n16 >> 2;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += 2 * stride) {
		const u64 u2 = u + stride;
		const bool two = u2 < units;
		const u32x2 a = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(in) + u);
		const u32x2 b = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(in) + (two ? u2 : u));
		const f32x4 oa = {sdr_unit((int)(short)(a.x & 0xffffu)), sdr_unit((int)a.x >> 16), sdr_unit((int)(short)(a.y & 0xffffu)), sdr_unit((int)a.y >> 16)};
		const f32x4 ob = {sdr_unit((int)(short)(b.x & 0xffffu)), sdr_unit((int)b.x >> 16), sdr_unit((int)(short)(b.y & 0xffffu)), sdr_unit((int)b.y >> 16)};
		__builtin_nontemporal_store(oa, reinterpret_cast<f32x4 *>(out) + u);
		if (two)
			__builtin_nontemporal_store(ob, reinterpret_cast<f32x4 *>(out) + u2);
	}
	if (blockIdx.x == 0 && threadIdx.x < (n16 & 3)) {
		const u64 i = (units << 2) + threadIdx.x;
		out[i] = sdr_unit(in[i]);
	}
}

// 12 packed bytes = 4 elements: b0 b1 b2 | b0 b1 b2 | ...;  I = (b1 << 12) | (b0 << 4),  Q = (b2 << 8) | (b1 & 0xf0)
__device__ __forceinline__ uint32_t sdr_cs12(uint32_t b0, uint32_t b1, uint32_t b2)
{
	return (((b1 << 12) | (b0 << 4)) & 0xffffu) | (((b2 << 8) | (b1 & 0xf0u)) << 16);
}

// n_elems elements of 3 bytes in, n_elems (I,Q) int16 pairs out; one thread per 4 elements (12 B -> 16 B), two units in flight
__global__ __launch_bounds__(256) void k_sdr_cs12_to_cs16(const uint8_t *__restrict__ in, u64 n_elems, uint32_t *__restrict__ out)
{
	const u64 units = n_elems >> 2;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	const uint32_t *src = reinterpret_cast<const uint32_t *>(in);
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += 2 * stride) {
		const u64 u2 = u + stride;
		const bool two = u2 < units;
		uint32_t w[2][3];
#pragma unroll
		for (int k = 0; k < 3; k++) {                                  // dword loads at 12-byte lane pitch: merged into dwordx3
			w[0][k] = __builtin_nontemporal_load(src + 3 * u + k);
			w[1][k] = __builtin_nontemporal_load(src + 3 * (two ? u2 : u) + k);
		}
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const uint32_t x = w[h][0], y = w[h][1], z = w[h][2];
			const u32x4 q = {sdr_cs12(x & 0xffu, (x >> 8) & 0xffu, (x >> 16) & 0xffu), sdr_cs12(x >> 24, y & 0xffu, (y >> 8) & 0xffu),
			                 sdr_cs12((y >> 16) & 0xffu, y >> 24, z & 0xffu), sdr_cs12((z >> 8) & 0xffu, (z >> 16) & 0xffu, z >> 24)};
			if (h == 0 || two)
				__builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(out) + (h ? u2 : u));
		}
	}
	if (blockIdx.x == 0 && threadIdx.x < (n_elems & 3)) {
		const u64 i = (units << 2) + threadIdx.x;
		out[i] = sdr_cs12(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
	}
}

static unsigned sdr_grid(u64 units)
{
	const u64 want = (units + 255) / 256;
	const u64 cap = 256ull * 16;                                       // 16 workgroups per CU, grid-stride beyond
	const u64 g = want < cap ? want : cap;
	return (unsigned)(g ? g : 1);
}

#define LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return (int)e_; } while (0)

extern "C" int rxk_sdr_cs16_to_8(void *stream, const int16_t *in, u64 n16, int is_unsigned, uint8_t *out)
{
	if (!n16)
		return 0;
	if (is_unsigned)
		hipLaunchKernelGGL((k_sdr_cs16_to_8<true>), dim3(sdr_grid(n16 >> 4)), dim3(256), 0, (hipStream_t)stream, in, n16, out);
	else
		hipLaunchKernelGGL((k_sdr_cs16_to_8<false>), dim3(sdr_grid(n16 >> 4)), dim3(256), 0, (hipStream_t)stream, in, n16, out);
	LAUNCH_RET();
}

extern "C" int rxk_sdr_cs16_to_cf32(void *stream, const int16_t *in, u64 n16, float *out)
{
	if (!n16)
		return 0;
	hipLaunchKernelGGL(k_sdr_cs16_to_cf32, dim3(sdr_grid(n16 >> 3)), dim3(256), 0, (hipStream_t)stream, in, n16, out);
	LAUNCH_RET();
}

extern "C" int rxk_sdr_cs12_to_cs16(void *stream, const uint8_t *in, u64 n_elems, int16_t *out)
{
	if (!n_elems)
		return 0;
	hipLaunchKernelGGL(k_sdr_cs12_to_cs16, dim3(sdr_grid(n_elems >> 3)), dim3(256), 0, (hipStream_t)stream, in, n_elems,
	                   (uint32_t *)out);
	LAUNCH_RET();
}
