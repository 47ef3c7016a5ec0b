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


typedef short pw_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pw_pk_add(uint32_t a, uint32_t b)
{
	pw_s16x2 r = __builtin_bit_cast(pw_s16x2, a) + __builtin_bit_cast(pw_s16x2, b);
	return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pw_pk_sub(uint32_t a, uint32_t b)
{
	pw_s16x2 r = __builtin_bit_cast(pw_s16x2, a) - __builtin_bit_cast(pw_s16x2, b);
	return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pw_pk_mul(uint32_t a, uint32_t b)
{
	pw_s16x2 r = __builtin_bit_cast(pw_s16x2, a) * __builtin_bit_cast(pw_s16x2, b);   // low 16 bits: the int16 wrap
	return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pw_pk_half(uint32_t a)
{
	pw_s16x2 r = __builtin_bit_cast(pw_s16x2, a) >> (pw_s16x2)(1);
	return __builtin_bit_cast(uint32_t, r);
}

// The same butterfly as `butterfly` above on packed registers, 10 VALU operations.
// FIX_MPY(w, x) = (w*x + 16384) >> 15 = HIGH HALF of (2w)*x + 32768, and 2w fits an int16 because the
// reference halves its twiddles first (wr = Sinewave[..] >> 1, rtl_power.c:300-301).  So with the twiddle
// table doubled (second half of rxgpu_twiddle_table) each product is one v_mad_i32_i16 that picks its int16 operands with op_sel,
// tr = A1 - A2 and ti = A3 + A4 are packed ops reading the high halves, and only the low 16 bits of
// either survive the int16 stores (truncation to int16 is a ring homomorphism).

// lo, hi: packed (re, im); tw2: packed (2*wr, 2*wi).  One asm block: none of these write a partial
// register, so no wait states are needed between them (the compiler pads every separate asm statement).
__device__ __forceinline__ void bfly_pk(uint32_t &lo, uint32_t &hi, uint32_t tw2)
{
	uint32_t p1, p2, p3, p4, q;
	asm("v_mad_i32_i16 %[p1], %[w], %[hi], %[c]\n\t"                              // 2wr*xr + 32768
	    "v_mad_i32_i16 %[p2], %[w], %[hi], %[c] op_sel:[1,1,0,0]\n\t"             // 2wi*xi
	    "v_mad_i32_i16 %[p3], %[w], %[hi], %[c] op_sel:[0,1,0,0]\n\t"             // 2wr*xi
	    "v_mad_i32_i16 %[p4], %[w], %[hi], %[c] op_sel:[1,0,0,0]\n\t"             // 2wi*xr
	    "v_pk_ashrrev_i16 %[q], 1, %[lo] op_sel_hi:[0,1]\n\t"                     // qr, qi
	    "v_pk_sub_i16 %[p1], %[p1], %[p2] op_sel:[1,1] op_sel_hi:[1,1]\n\t"       // tr (both halves)
	    "v_pk_add_u16 %[p3], %[p3], %[p4] op_sel:[1,1] op_sel_hi:[1,1]\n\t"       // ti (both halves)
	    "v_bfi_b32 %[p1], %[m], %[p1], %[p3]\n\t"                                 // (tr, ti)
	    "v_pk_sub_i16 %[hi], %[q], %[p1]\n\t"
	    "v_pk_add_u16 %[lo], %[q], %[p1]"
	    : [lo] "+v"(lo), [hi] "+v"(hi), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4), [q] "=&v"(q)
	    : [w] "v"(tw2), [c] "s"(32768), [m] "s"(0xffff));
}

// re^2 + im^2 of a packed bin as one v_dot2: at most 2^31, which the uint32 holds (rtl_power.c:664-668)
__device__ __forceinline__ uint32_t pw_norm(uint32_t v)
{
	const pw_s16x2 a = __builtin_bit_cast(pw_s16x2, v);
	return (uint32_t)__builtin_amdgcn_sdot2(a, a, 0, false);
}

// acc + a.lo*b.lo + a.hi*b.hi (v_dot2c_i32_i16); with b = (1,0) / (0,1) it adds one half of a packed sample
__device__ __forceinline__ int pw_dot(uint32_t a, uint32_t b, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(pw_s16x2, a), __builtin_bit_cast(pw_s16x2, b), acc, false);
}

template <int BITS> __device__ __forceinline__ constexpr int crev(int v)
{
	int r = 0;
	for (int i = 0; i < BITS; i++) r |= ((v >> i) & 1) << (BITS - 1 - i);
	return r;
}


// ------------------------------------------------------------------ register-blocked fix_fft, N = 2^M

// The reference runs its DIT stages on the bit-reversed array (rtl_power.c:275-318).  Here the data
// stay in natural order: stage s pairs n with n + 2^(M-1-s) and its twiddle index is
// rev_s(top s bits of n) << (M-1-s).  A thread holds the 16 values of one 4-bit field of n, does up to
// four stages in registers, then the workgroup transposes through LDS to the next field.  Fields
// are taken from the top; when M is not a multiple of 4 the last field (bits 3..0) overlaps the
// previous one and only its remaining stages run.  N/16 threads per transform.
#include "kernels.h"
template <int M> struct fft_geom {
	static constexpr int P = (M + 3) / 4;                       // passes
	static constexpr int N = 1 << M;
	static constexpr int TPF = N / 16;                          // threads per transform
	static constexpr int ROW = 16;                              // 16 data dwords per LDS row; every FOUR rows the area shifts by 4 dwords (fft_exchange)
	__host__ __device__ static constexpr int f(int p) { return (M - 4 * (p + 1)) > 0 ? (M - 4 * (p + 1)) : 0; }   // field's low bit
	__host__ __device__ static constexpr int u(int p) { return M - 4 - f(p); }                                    // bits above the field
	__host__ __device__ static constexpr int sp0(int p) { return (p == P - 1) ? (4 * P - M) : 0; }                // stages already done
	// the workgroup's LDS copy of the (doubled) twiddle table, N/2 entries PERMUTED: entry j at row j >> TWC, column rev_TWC(j), rows TWS
	// dwords apart (fft_tw_fill / fft_tw_addr below)
	static constexpr int TWC = M - 4;
	static constexpr int TWS = (1 << TWC) + 8;
	static constexpr int TW_WORDS = 8 * TWS;
	// dwords to reserve per thread-row of a transpose area: ROW plus the skew of fft_exchange (4 dwords every 4 rows) = RXK_FFT_XROW (kernels.h: the launchers)
	static constexpr int XROW = ROW + 1;
	static_assert(XROW == RXK_FFT_XROW, "the launchers size the transpose areas with RXK_FFT_XROW");
};

// ---- twiddles from LDS (round 4; north_star: "fix_fft twiddles staged in LDS").  Stage s' of a pass indexes the table with
//   j = (rev_U(x) << (SH - s')) + K(s', g),   x = the thread's U index bits above the pass's field, K a compile-time multiple of 2^(M-1-s')
// -- a different entry for every thread, which the compiler leaves inside the caller's block loop as 15 global_load_dword per pass (an L1
// hit still costs several hundred cycles, waited for right behind the transpose).  The table is small (N/2 dwords): a workgroup keeps a
// copy in LDS.  In natural order the bit-reversed index would put a wave's reads of one stage on 1, 2, 4 or 8 banks; the copy is
// therefore permuted: entry j sits at row j >> (M-4) (eight rows), column rev_{M-4}(j's low bits).  For the thread, with k = 3 - s':
//   column = x >> k   (consecutive lanes, consecutive columns),   row = rev_k(x's low k bits) + K / 2^(M-4),
// and K's part is an immediate offset of the ds_read_b32.  Rows are 2^(M-4) + 8 dwords apart (8 mod 32 banks for M >= 9): at most a
// 2-way conflict on any read.  Pass 0 (U = 0) has wave-uniform indices: those stay scalar loads from the global table.
template <int M>
__device__ __forceinline__ void fft_tw_fill(uint32_t *__restrict__ tl, const uint32_t *__restrict__ tw, int tid, int threads)
{
	typedef fft_geom<M> G;
	for (int j = tid; j < G::N / 2; j += threads)
		tl[(j >> G::TWC) * G::TWS + (int)(__brev((unsigned)j) >> (32 - G::TWC))] = tw[j];
}

template <int M, int PASS>
__device__ __forceinline__ void fft_tw_addr(unsigned tq, unsigned (&ta)[4])
{
	typedef fft_geom<M> G;
	const unsigned x = tq >> G::f(PASS);
#pragma unroll
	for (int sp = 0; sp < 4; sp++) {
		const int k = 3 - sp;
		const unsigned low = x & ((1u << k) - 1u);
		const unsigned row = k ? (__brev(low) >> (32 - k)) : 0u;
		ta[sp] = row * G::TWS + (x >> k);
	}
}

// The threads of one transform exchange their values through LDS between two passes.  With N/16 <= 64 threads per transform (N <= 1024)
// those threads are lanes of ONE wave (the kernels give a transform an aligned run of threads): a wave's LDS instructions execute in
// program order, so no s_barrier is needed -- only that the compiler keeps the order -- and the waves of a workgroup stop waiting for
// each other three times per transform.  Larger transforms span waves: a workgroup barrier.
// The wave-private form is only right on 64-wide waves with a transform's TPF threads an aligned run of lanes inside one wave: gfx950 is
// wave64 only, and the callers assert the layout (TPF divides 64, blockDim a multiple of 64).
// (ROCm 7 no longer defines __AMDGCN_WAVEFRONT_SIZE; the target is checked instead, and rxgpu_init refuses a device whose warpSize is not 64.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fft_sync and the DPP scans of librxgpu are written for the 64-wide wavefronts of gfx950"
#endif
template <int M>
__device__ __forceinline__ void fft_sync()
{
	static_assert(((1 << M) / 16) > 64 || 64 % ((1 << M) / 16) == 0, "a transform's threads must be an aligned run of lanes of one wave");
	if constexpr (((1 << M) / 16) <= 64) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	} else {
		__syncthreads();
	}
}

template <int M, int PASS>
__device__ __forceinline__ void fft_pass(uint32_t (&v)[16], const uint32_t *__restrict__ tw, unsigned tq)
{
	typedef fft_geom<M> G;
	constexpr int F = G::f(PASS), U = G::u(PASS), SH = M - 1 - U;
	const unsigned base = U ? (__brev(tq >> F) >> (32 - (U ? U : 1))) : 0u;     // rev_U of the bits above the field
#pragma unroll
	for (int sp = G::sp0(PASS); sp < 4; sp++) {
		const int d = 8 >> sp;
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const unsigned j = (U ? (base << (SH - sp)) : 0u) + ((unsigned)crev<4>(g << (4 - sp)) << (M - 1 - sp));
			const uint32_t w = tw[j];                                      // doubled table (rxgpu_twiddle_table)
#pragma unroll
			for (int q = 0; q < d; q++) {
				const int r = g * 2 * d + q;
				bfly_pk(v[r], v[r + d], w);
			}
		}
	}
}

// the same pass with its twiddles from the workgroup's permuted LDS copy (fft_tw_fill); ta: fft_tw_addr<M, PASS>(tq)
template <int M, int PASS>
__device__ __forceinline__ void fft_pass_lds(uint32_t (&v)[16], const uint32_t *tl, const unsigned (&ta)[4])
{
	typedef fft_geom<M> G;
	static_assert(G::u(PASS) > 0, "pass 0 has wave-uniform twiddles: scalar loads from the global table");
#pragma unroll
	for (int sp = G::sp0(PASS); sp < 4; sp++) {
		const int d = 8 >> sp;
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const unsigned krow = ((unsigned)crev<4>(g << (4 - sp)) << (M - 1 - sp)) >> G::TWC;     // compile-time
			const uint32_t w = tl[ta[sp] + krow * G::TWS];
#pragma unroll
			for (int q = 0; q < d; q++) {
				const int r = g * 2 * d + q;
				bfly_pk(v[r], v[r + d], w);
			}
		}
	}
}

// ... or from registers the caller loaded once for all its transforms (a thread's twiddles do not change from block to block): twr[15],
// slot (1 << sp) - 1 + g
template <int M, int PASS>
__device__ __forceinline__ void fft_tw_regs(uint32_t (&twr)[15], const uint32_t *__restrict__ tw, unsigned tq)
{
	typedef fft_geom<M> G;
	constexpr int F = G::f(PASS), U = G::u(PASS), SH = M - 1 - U;
	const unsigned base = U ? (__brev(tq >> F) >> (32 - (U ? U : 1))) : 0u;
#pragma unroll
	for (int sp = 0; sp < 4; sp++)
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const unsigned j = (U ? (base << (SH - sp)) : 0u) + ((unsigned)crev<4>(g << (4 - sp)) << (M - 1 - sp));
			twr[(1 << sp) - 1 + g] = sp >= G::sp0(PASS) ? tw[j] : 0u;
		}
}
template <int M, int PASS>
__device__ __forceinline__ void fft_pass_regs(uint32_t (&v)[16], const uint32_t (&twr)[15])
{
	typedef fft_geom<M> G;
#pragma unroll
	for (int sp = G::sp0(PASS); sp < 4; sp++) {
		const int d = 8 >> sp;
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const uint32_t w = twr[(1 << sp) - 1 + g];
#pragma unroll
			for (int q = 0; q < d; q++) {
				const int r = g * 2 * d + q;
				bfly_pk(v[r], v[r + d], w);
			}
		}
	}
}

// Layout of the transposes (round 6).  Row t (the 16 values thread t reads back with four ds_read_b128) starts at dword 16 t + 4 (t >> 2): rows
// of 16 dwords, every four rows the area shifts by 4 dwords (16-byte aligned).  A ds_read_b128 is served in four groups of 16 lanes over 64
// banks (MI355X_MICROARCH.md, LDS): with rows 20 dwords apart and a shift every eight rows (rounds 3-5) three of the 16 lanes of every group
// met on a bank quad -- the reads took three times their cycles, and rocprofv3 counted 42 % of k_ch_fftR<10>'s LDS cycles as conflicts -- here
// the 16 rows of a group fall on 16 different quads for every (M, PASS) from 2^8 to 2^13 (exhaustive over the lane groups).  The WRITE side
// (one dword per lane into 64 different rows) is left with 2-way conflicts in one of a transform's exchanges, which a ds_write_b32 hides behind its
// own 4-cycle data transfer.  17 dwords per row instead of 21: 4 KiB less LDS per 256 rows.
// the same choice by the number of threads that exchange with each other (k_pwm_tail: a SUB-transform of the large transform; its threads are
// an aligned run of lanes of one wave when there are at most 64 of them)
template <int THREADS>
__device__ __forceinline__ void fft_sync_n()
{
	static_assert(THREADS > 64 || 64 % THREADS == 0, "the exchanging threads must be an aligned run of lanes of one wave");
	if constexpr (THREADS <= 64) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	} else {
		__syncthreads();
	}
}

// LDS transpose between the layouts of pass PASS and PASS+1.  row0 (a multiple of 8): the first row of the transform that `lds` holds --
// a workgroup that owns rows row0 .. of a larger transform (k_pwm_tail) passes its own area and its first row
// XT: threads that exchange with each other (default: the whole transform's N/16)
template <int M, int PASS, int XT = (1 << M) / 16>
__device__ __forceinline__ void fft_exchange(uint32_t (&v)[16], uint32_t *__restrict__ lds, unsigned tq, unsigned row0 = 0)
{
	typedef fft_geom<M> G;
	constexpr int F = G::f(PASS), F2 = G::f(PASS + 1);
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const unsigned n = ((tq >> F) << (F + 4)) | ((unsigned)r << F) | (tq & ((1u << F) - 1u));
		const unsigned row = (((n >> (F2 + 4)) << F2) | (n & ((1u << F2) - 1u))) - row0;
		const unsigned col = (n >> F2) & 15u;
		lds[row * G::ROW + (row >> 2) * 4 + col] = v[r];
	}
	fft_sync_n<XT>();
	const unsigned tl = tq - row0;
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const uint4 t4 = *reinterpret_cast<const uint4 *>(&lds[tl * G::ROW + (tl >> 2) * 4 + 4 * c]);
		v[4 * c] = t4.x; v[4 * c + 1] = t4.y; v[4 * c + 2] = t4.z; v[4 * c + 3] = t4.w;
	}
}

// in:  v[r] = x[tq + r * N/16]  (natural order)          out: v[r] = X[rev_M((tq << 4) | r)]
// lds_a/lds_b: this transform's transpose areas (N/16 rows, XROW dwords reserved per row); if they are the same
// buffer the caller's DOUBLE must be false and a barrier separates reuse.
// tl != nullptr (TWL): passes 1.. take their twiddles from the workgroup's LDS copy (fft_tw_fill), ta[p - 1] = fft_tw_addr<M, p>(tq).
template <int M, bool DOUBLE, bool TWL = false>
__device__ __forceinline__ void fft_reg(uint32_t (&v)[16], unsigned tq, uint32_t *lds_a, uint32_t *lds_b,
                                        const uint32_t *__restrict__ tw, const uint32_t *tl = nullptr, const unsigned (*ta)[4] = nullptr)
{
	typedef fft_geom<M> G;
	// the previous transform's last reads of lds_a must be over before this one's first writes
	if (!DOUBLE || (G::P % 2) == 0)
		fft_sync<M>();
	fft_pass<M, 0>(v, tw, tq);
	if constexpr (G::P > 1) {
		fft_exchange<M, 0>(v, lds_a, tq);
		if constexpr (TWL) fft_pass_lds<M, 1>(v, tl, ta[0]);
		else fft_pass<M, 1>(v, tw, tq);
	}
	if constexpr (G::P > 2) {
		if (!DOUBLE) fft_sync<M>();
		fft_exchange<M, 1>(v, DOUBLE ? lds_b : lds_a, tq);
		if constexpr (TWL) fft_pass_lds<M, 2>(v, tl, ta[1]);
		else fft_pass<M, 2>(v, tw, tq);
	}
	if constexpr (G::P > 3) {
		if (!DOUBLE) fft_sync<M>();
		fft_exchange<M, 2>(v, lds_a, tq);
		if constexpr (TWL) fft_pass_lds<M, 3>(v, tl, ta[2]);
		else fft_pass<M, 3>(v, tw, tq);
	}
}

// ta for fft_reg<.., TWL = true>
template <int M>
__device__ __forceinline__ void fft_tw_addr_all(unsigned tq, unsigned (&ta)[3][4])
{
	typedef fft_geom<M> G;
	if constexpr (G::P > 1) fft_tw_addr<M, 1>(tq, ta[0]);
	if constexpr (G::P > 2) fft_tw_addr<M, 2>(tq, ta[1]);
	if constexpr (G::P > 3) fft_tw_addr<M, 3>(tq, ta[2]);
}

#endif
