// fft_device.h -- the reference's fixed-point butterfly (rtl_power.c:256-262, 302-314) as device code,
// shared by the rx_power kernels and the rx_fm channeliser.
#ifndef RXGPU_FFT_DEVICE_H
#define RXGPU_FFT_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t pw_pack(int i, int q) { return ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16); }
__device__ __forceinline__ int pw_lo(uint32_t w) { return (int)(short)(w & 0xffffu); }
__device__ __forceinline__ int pw_hi(uint32_t w) { return (int)w >> 16; }

// FIX_MPY, rtl_power.c:256-262: c = (a*b)>>14; (c>>1) + (c&1)  ==  (a*b + 16384) >> 15,
// then truncated to int16 by the return type.
__device__ __forceinline__ int fix_mpy(int a, int b) { return (int)(short)((a * b + 16384) >> 15); }

// one radix-2 DIT butterfly of fix_fft, rtl_power.c:302-314 (shift == 1 always)
__device__ __forceinline__ void butterfly(uint32_t &lo, uint32_t &hi, uint32_t tw)
{
	const int wr = pw_lo(tw), wi = pw_hi(tw);
	const int xr = pw_lo(hi), xi = pw_hi(hi);
	const int tr = (int)(short)(fix_mpy(wr, xr) - fix_mpy(wi, xi));
	const int ti = (int)(short)(fix_mpy(wr, xi) + fix_mpy(wi, xr));
	const int qr = pw_lo(lo) >> 1, qi = pw_hi(lo) >> 1;
	hi = pw_pack(qr - tr, qi - ti);
	lo = pw_pack(qr + tr, qi + ti);
}


#endif
