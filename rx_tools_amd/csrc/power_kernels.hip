// power_kernels.hip -- gfx950 kernels for the rx_power scanner() per-tune chain.
//
// Reference behaviour (file:line under /root/reference/src/rtl_power.c):
//   P1 copy 715-720      P2 boxcar 723-733     P3 fifth_order 582-607 / downsample_iq 656-662
//   P4 remove_dc 609-624 P5 window 749-758     P6 FIX_MPY 256-262
//   P7 fix_fft 264-320   P8 real_conj + accumulate 664-668, 760-768
//   P9 rms_power sums 403-417                  generic_fir 626-654
//
// Everything is int16/int32/int64 and bit-exact with the C reference, including the int16
// wrap of the window product, of every FIX_MPY result and of every butterfly store.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include <stdlib.h>

typedef unsigned long long u64;
typedef long long i64;

#include "fft_device.h"

// ------------------------------------------------------------------ P4-P8, any N that fits LDS

// One workgroup per (tune, pass group).  For each pass of the group: remove_dc sums over
// the tune buffer, then per FFT block: window -> LDS (bit-reversed) -> m radix-2 stages in LDS
// -> |X|^2 accumulated.  N <= 4096 keeps per-thread int64 accumulators across the group's
// passes (bins k = tid + 256 r) and touches global memory once; larger N adds per block.
template <int ACC>
__global__ __launch_bounds__(256) void k_pw_fft(
	const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int passes, int bin_e, int eff_len,
	const int *__restrict__ window, const uint32_t *__restrict__ twiddle, int peak_hold, int ppg,
	i64 *__restrict__ avg)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t x[];     // N packed IQ, then 16 reduction words
	const int n = 1 << bin_e;
	i64 *red = (i64 *)(x + n);
	const int tune = blockIdx.x;
	const int p_begin = blockIdx.y * ppg;
	const int p_end = min(passes, p_begin + ppg);
	const int tid = threadIdx.x;
	const int n_blocks = (eff_len + 2 * n - 1) / (2 * n);
	i64 acc[ACC > 0 ? ACC : 1];
#pragma unroll
	for (int r = 0; r < (ACC > 0 ? ACC : 1); r++)
		acc[r] = 0;
	i64 *avg_t = avg + (size_t)tune * n;

	for (int pass = p_begin; pass < p_end; pass++) {
		const uint32_t *buf = (const uint32_t *)(in + (size_t)pass * pass_stride + (size_t)tune * tune_stride);
		// ---- remove_dc, rtl_power.c:609-624 via 744-745: I over int16 indices 0,2,.. < L,
		// Q over 1,3,.. < L (i.e. data+1, length L-1); divide by the int16 COUNT
		const int L = eff_len;
		const int ci = (L + 1) / 2, cq = L / 2;            // complex samples taking part
		i64 si = 0, sq = 0;
		for (int c = tid; c < ci; c += 256) {
			const uint32_t w = buf[c];
			si += pw_lo(w);
			if (c < cq) sq += pw_hi(w);
		}
		for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
		__syncthreads();
		if ((tid & 63) == 0) { red[tid >> 6] = si; red[4 + (tid >> 6)] = sq; }
		__syncthreads();
		si = red[0] + red[1] + red[2] + red[3];
		sq = red[4] + red[5] + red[6] + red[7];
		const int ave_i = (int)(short)(si / (i64)L);
		const int ave_q = (L > 1) ? (int)(short)(sq / (i64)(L - 1)) : 0;

		for (int blk = 0; blk < n_blocks; blk++) {
			const int c0 = blk * n;
			// ---- window, rtl_power.c:749-758, into bit-reversed LDS slots (275-290)
			for (int j = tid; j < n; j += 256) {
				const int c = c0 + j;
				const uint32_t w = buf[c];
				int vi = pw_lo(w), vq = pw_hi(w);
				if (c < ci) vi = (int)(short)(vi - ave_i);
				if (c < cq) vq = (int)(short)(vq - ave_q);
				const int coef = window[j];
				vi = vi * coef;                                // int32, then (int16) truncation
				vq = vq * coef;
				x[__brev((unsigned)j) >> (32 - bin_e)] = pw_pack(vi, vq);
			}
			__syncthreads();
			// ---- fix_fft stages, rtl_power.c:291-318
			for (int s = 0; s < bin_e; s++) {
				const int half = 1 << s;
				for (int b = tid; b < n / 2; b += 256) {
					const int t = b & (half - 1);
					const int lo_i = ((b >> s) << (s + 1)) | t;
					uint32_t lo = x[lo_i], hi = x[lo_i + half];
					butterfly(lo, hi, twiddle[t << (bin_e - 1 - s)]);
					x[lo_i] = lo;
					x[lo_i + half] = hi;
				}
				__syncthreads();
			}
			// ---- real_conj + accumulate, rtl_power.c:664-668, 760-768
			if (ACC > 0) {
#pragma unroll
				for (int r = 0; r < ACC; r++) {
					const int k = tid + 256 * r;
					if (k < n) {
						const uint32_t w = x[k];
						const i64 re = pw_lo(w), im = pw_hi(w);
						const i64 pw = re * re + im * im;
						acc[r] = peak_hold ? (pw > acc[r] ? pw : acc[r]) : acc[r] + pw;
					}
				}
			} else {
				for (int k = tid; k < n; k += 256) {
					const uint32_t w = x[k];
					const i64 re = pw_lo(w), im = pw_hi(w);
					const i64 pw = re * re + im * im;
					if (peak_hold) atomicMax((long long *)&avg_t[k], pw);
					else atomicAdd((unsigned long long *)&avg_t[k], (unsigned long long)pw);
				}
			}
			__syncthreads();
		}
	}
	if (ACC > 0) {
#pragma unroll
		for (int r = 0; r < ACC; r++) {
			const int k = tid + 256 * r;
			if (k < n) {
				if (peak_hold) atomicMax((long long *)&avg_t[k], acc[r]);
				else atomicAdd((unsigned long long *)&avg_t[k], (unsigned long long)acc[r]);
			}
		}
	}
}

// ------------------------------------------------------------------ P4-P8, N = 4096, register-blocked

// Four consecutive radix-2 stages on 16 registers.  The reference runs its DIT stages on the
// bit-reversed array; on the natural-order index n stage s pairs n with n + 2^(11-s) and its
// twiddle index is rev_s(top s bits of n) << (11-s).  A thread holds the 16 values of one
// 4-bit field of n, so within a pass the pair distance is 8,4,2,1 registers and the twiddle index
// is  (base << (sh - s')) + (rev_s'(r >> (4-s')) << (11 - s'))  with `base` the reversed bits above
// the field (0 for the first pass).
template <int SH, bool FIRST>
__device__ __forceinline__ void radix16_pass(uint32_t (&v)[16], const uint32_t *__restrict__ tw, unsigned base)
{
#pragma unroll
	for (int sp = 0; sp < 4; sp++) {
		const int d = 8 >> sp;
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const unsigned j = (FIRST ? 0u : (base << (SH - sp))) + ((unsigned)crev<4>(g << (4 - sp)) << (11 - sp));
			const uint32_t w = tw[j];                                      // doubled table
#pragma unroll
			for (int q = 0; q < d; q++) {
				const int r = g * 2 * d + q;
				bfly_pk(v[r], v[r + d], w);
			}
		}
	}
}

// ---- the twiddle table in LDS (round 4).  Stages 4-11 index the table with the thread's own reversed index bits: 15 loads per pass
// and thread that the compiler keeps inside the block loop as global_load_dword (an L1 hit is still several hundred cycles, waited for
// right behind the transpose's barrier: four exposed stalls per pass of the tune -- the counters' 26 % of wave time waiting).  The
// doubled table of N = 4096 is 2048 dwords: one copy per workgroup, loaded once for all the passes of its group, PERMUTED so that the
// reads are (nearly) conflict-free: entry j sits at row (j >> 8), column rev8(j & 255), rows F4K_TWS dwords apart.  For the last field
// (stages 8-11, base = rev8(tid)) the index is (base << (3 - s')) + K: its low eight bits reversed are tid >> (3 - s') -- consecutive
// lanes, consecutive columns -- and its row is rev(tid's low 3 - s' bits) + K / 256; the middle field (stages 4-7, base = rev4(tid >> 4))
// is the same with tid >> 4 in place of tid (sixteen lanes share an address: a broadcast).  K / 256 * F4K_TWS is an immediate offset of
// the ds_read_b32.  Row stride 264 = 8 mod 32 banks: at most a 2-way conflict on any of the 30 reads.
#define F4K_TWS 264
#define F4K_TW_WORDS (8 * F4K_TWS)

// the four (one per stage of the pass) LDS word addresses of a thread's twiddles, without the K / 256 * F4K_TWS part; x = tid >> 4 for the
// middle field, tid for the last one
__device__ __forceinline__ void f4k_tw_addr(unsigned x, unsigned (&ta)[4])
{
#pragma unroll
	for (int sp = 0; sp < 4; sp++) {
		const int k = 3 - sp;                                           // index bits that spill into the row
		const unsigned low = x & ((1u << k) - 1u);
		const unsigned row = k ? (__brev(low) >> (32 - k)) : 0u;
		ta[sp] = row * F4K_TWS + (x >> k);
	}
}

__device__ __forceinline__ void radix16_pass_lds(uint32_t (&v)[16], const uint32_t *tl, const unsigned (&ta)[4])
{
#pragma unroll
	for (int sp = 0; sp < 4; sp++) {
		const int d = 8 >> sp;
#pragma unroll
		for (int g = 0; g < (1 << sp); g++) {
			const unsigned krow = ((unsigned)crev<4>(g << (4 - sp)) << (11 - sp)) >> 8;      // compile-time
			const uint32_t w = tl[ta[sp] + krow * F4K_TWS];
#pragma unroll
			for (int q = 0; q < d; q++) {
				const int r = g * 2 * d + q;
				bfly_pk(v[r], v[r + d], w);
			}
		}
	}
}

#define F4K_ROW 20                       // 16 data dwords + 4 pad: rows stay 16-byte aligned, b128 reads conflict-free
#define F4K_BUF (256 * F4K_ROW)

// One workgroup (256 threads, 16 values each) per (tune, pass group); NB FFT blocks per tune buffer
// are loaded once into registers, remove_dc is reduced from them, then per block:
//   window -> stages 0-3 in registers -> LDS transpose -> stages 4-7 -> LDS transpose -> stages 8-11
//   -> |X|^2 into 16 per-thread int64 accumulators (bin = rev8(tid) + 256 * rev4(r)).
template <int NB, bool PEAK, bool TWL = true>
__global__ __launch_bounds__(256) void k_pw_fft4096(
	const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int passes,
	const int *__restrict__ window, const uint32_t *__restrict__ twiddle, int ppg, i64 *__restrict__ avg, i64 *__restrict__ partial)
{
	__shared__ __attribute__((aligned(16))) uint32_t xa[F4K_BUF];
	__shared__ __attribute__((aligned(16))) uint32_t xb[F4K_BUF + 16 * 16];
	__shared__ i64 red[8];
	__shared__ uint32_t tl[TWL ? F4K_TW_WORDS : 1];
	const int tid = threadIdx.x;
	unsigned ta_b[4], ta_c[4];
	if (TWL) {
		// the doubled table, permuted (see F4K_TWS): 8 entries per thread, once per workgroup; the first __syncthreads of the pass loop orders it
#pragma unroll
		for (int k = 0; k < 8; k++)
			tl[k * F4K_TWS + (__brev((unsigned)tid) >> 24)] = twiddle[tid + 256 * k];
		f4k_tw_addr((unsigned)tid >> 4, ta_b);
		f4k_tw_addr((unsigned)tid, ta_c);
	}
	const int tune = blockIdx.x;
	const int p_begin = blockIdx.y * ppg;
	const int p_end = min(passes, p_begin + ppg);
	const unsigned hi4 = tid >> 4, lo4 = tid & 15;
	// Banks of the first transpose's ds_write_b32 (row = 16 r + lo4, column = hi4, rows 20 dwords apart): the LDS serves 32 lanes per
	// cycle from 32 banks and 20 k mod 32 takes only EIGHT values, so rows k and k + 8 of a half-wave met in one bank (rocprofv3:
	// a third of the LDS cycles were conflicts).  Rows with bit 3 set sit 2 dwords further inside their 20: 8 x 4 different banks.
	// Such rows are 8-byte aligned, the read side takes its 16 bytes as two 8-byte halves (ds_read2_b64).
	const unsigned xa_w = lo4 * F4K_ROW + hi4 + 2u * ((lo4 >> 3) & 1u), xa_r = (unsigned)tid * F4K_ROW + 2u * (((unsigned)tid >> 3) & 1u);
	// the second transpose (row = 16 hi4 + r, column = lo4) put the two 16-row groups of a half-wave on the same 16 banks:
	// every group starts 16 dwords further than 16 rows would put it (336 g: the bank offset alternates 0, 16; rows stay 16-byte
	// aligned for the ds_read_b128 side)
	const unsigned xb_w = (hi4 * 16u) * F4K_ROW + lo4 + 16u * hi4, xb_r = (unsigned)tid * F4K_ROW + 16u * hi4;
	const unsigned base_b = __brev(hi4) >> 28;           // rev4 of bits 11..8
	const unsigned base_c = __brev((unsigned)tid) >> 24; // rev8 of bits 11..4
	uint32_t wcoef[16];
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const uint32_t c = (uint32_t)window[tid + 256 * r] & 0xffffu;
		wcoef[r] = c | (c << 16);
	}
	i64 acc[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = 0;

	// TWL: the next pass's samples are requested while this pass is transformed (the LDS table frees the registers: 129 + 16 NB VGPRs,
	// three waves per SIMD either way -- the workgroup's LDS allows no more); without it every pass began by waiting for HBM
	uint32_t dn[TWL ? NB : 1][16];
	if (TWL && p_begin < p_end) {
		const uint32_t *buf0 = (const uint32_t *)(in + (size_t)p_begin * pass_stride + (size_t)tune * tune_stride);
#pragma unroll
		for (int b = 0; b < NB; b++)
#pragma unroll
			for (int r = 0; r < 16; r++)
				dn[b][r] = buf0[b * 4096 + tid + 256 * r];
	}
	for (int pass = p_begin; pass < p_end; pass++) {
		const uint32_t *buf = (const uint32_t *)(in + (size_t)pass * pass_stride + (size_t)tune * tune_stride);
		uint32_t d[NB][16];
		int si = 0, sq = 0;              // 8192 * NB samples of int16: fits int32 for NB <= 8
#pragma unroll
		for (int b = 0; b < NB; b++)
#pragma unroll
			for (int r = 0; r < 16; r++) {
				d[b][r] = TWL ? dn[b][r] : buf[b * 4096 + tid + 256 * r];
				si = pw_dot(d[b][r], 0x00000001u, si);           // += I
				sq = pw_dot(d[b][r], 0x00010000u, sq);           // += Q
			}
		if (TWL && pass + 1 < p_end) {
			const uint32_t *bufn = (const uint32_t *)(in + (size_t)(pass + 1) * pass_stride + (size_t)tune * tune_stride);
#pragma unroll
			for (int b = 0; b < NB; b++)
#pragma unroll
				for (int r = 0; r < 16; r++)
					dn[b][r] = bufn[b * 4096 + tid + 256 * r];
		}
		// remove_dc, rtl_power.c:609-624 via 744-745 (L = 2*4096*NB int16, both halves complete)
		for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
		__syncthreads();
		if ((tid & 63) == 0) { red[tid >> 6] = si; red[4 + (tid >> 6)] = sq; }
		__syncthreads();
		const i64 ti64 = red[0] + red[1] + red[2] + red[3], tq64 = red[4] + red[5] + red[6] + red[7];
		const int L = 2 * 4096 * NB;
		const uint32_t ave = pw_pack((int)(short)(ti64 / L), (int)(short)(tq64 / (L - 1)));

#pragma unroll
		for (int b = 0; b < NB; b++) {
			uint32_t v[16];
#pragma unroll
			for (int r = 0; r < 16; r++)
				v[r] = pw_pk_mul(pw_pk_sub(d[b][r], ave), wcoef[r]);       // window, rtl_power.c:749-758
			// stages 0-3: this thread holds n = tid + 256 r, r = bits 11..8
			radix16_pass<0, true>(v, twiddle, 0u);
			// transpose 1: next field is bits 7..4; row (hi4', lo4) gets the 16 values of that field
#pragma unroll
			for (int r = 0; r < 16; r++)
				xa[xa_w + r * 16 * F4K_ROW] = v[r];
			__syncthreads();
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const uint2 ta = *reinterpret_cast<const uint2 *>(&xa[xa_r + 4 * c]), tb = *reinterpret_cast<const uint2 *>(&xa[xa_r + 4 * c + 2]);
				v[4 * c] = ta.x; v[4 * c + 1] = ta.y; v[4 * c + 2] = tb.x; v[4 * c + 3] = tb.y;
			}
			// stages 4-7: n = hi4<<8 | r<<4 | lo4
			if (TWL) radix16_pass_lds(v, tl, ta_b);
			else radix16_pass<7, false>(v, twiddle, base_b);
			// transpose 2: next field is bits 3..0; row = bits 11..4
#pragma unroll
			for (int r = 0; r < 16; r++)
				xb[xb_w + r * F4K_ROW] = v[r];
			__syncthreads();
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const uint4 t4 = *reinterpret_cast<const uint4 *>(&xb[xb_r + 4 * c]);
				v[4 * c] = t4.x; v[4 * c + 1] = t4.y; v[4 * c + 2] = t4.z; v[4 * c + 3] = t4.w;
			}
			// stages 8-11: n = tid<<4 | r
			if (TWL) radix16_pass_lds(v, tl, ta_c);
			else radix16_pass<3, false>(v, twiddle, base_c);
			// real_conj + accumulate, rtl_power.c:664-668, 760-768; re^2+im^2 <= 2^31 fits u32
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const i64 pw = (i64)pw_norm(v[r]);
				acc[r] = PEAK ? (pw > acc[r] ? pw : acc[r]) : acc[r] + pw;
			}
		}
	}
	// few tunes: this group's spectrum to partial[(group * tunes + tune) * 4096 + bin] (k_pwm_reduce folds the groups into avg);
	// otherwise straight into avg with one atomic per bin and group
	i64 *avg_t = avg + (size_t)tune * 4096;
	i64 *dst = partial ? partial + ((size_t)blockIdx.y * gridDim.x + tune) * 4096 : nullptr;
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const unsigned bin = base_c + 256u * (unsigned)crev<4>(r);
		if (dst) dst[bin] = acc[r];
		else if (PEAK) atomicMax((long long *)&avg_t[bin], acc[r]);
		else atomicAdd((unsigned long long *)&avg_t[bin], (unsigned long long)acc[r]);
	}
}

// ------------------------------------------------------------------ P4-P8, N = 2^M (M = 8..13), register-blocked

// Same structure as k_pw_fft4096 for any power of two: N/16 threads per transform, 256/(N/16) transforms
// side by side in a 256-thread workgroup for N < 4096.  remove_dc needs the whole tune first, so a pass
// starts with a reduction over the tune buffer and the blocks are then re-read (L2) group by group.
template <int M, bool PEAK>
__global__ __launch_bounds__(((1 << M) / 16 > 256) ? (1 << M) / 16 : 256) void k_pw_fftR(
	const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int passes, int nb_total,
	const int *__restrict__ window, const uint32_t *__restrict__ twiddle, int ppg, i64 *__restrict__ avg, i64 *__restrict__ partial)
{
	typedef fft_geom<M> G;
	constexpr int N = G::N, TPF = G::TPF, T = TPF > 256 ? TPF : 256, FPW = T / TPF;
	constexpr bool DB = M <= 12;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *xa = lds, *xb = DB ? lds + T * G::XROW : lds;
	i64 *red = (i64 *)(lds + (DB ? 2 : 1) * T * G::XROW);
	uint32_t *tl = (uint32_t *)(red + 32);                  // the permuted twiddle copy (fft_tw_fill), G::TW_WORDS dwords
	const int tid = threadIdx.x, fid = tid / TPF;
	const unsigned tq = tid % TPF;
	const int tune = blockIdx.x;
	const int p_begin = blockIdx.y * ppg, p_end = min(passes, p_begin + ppg);
	fft_tw_fill<M>(tl, twiddle, tid, T);                   // ordered by the pass loop's first __syncthreads
	unsigned ta[3][4];
	fft_tw_addr_all<M>(tq, ta);
	constexpr bool WREG = M <= 12;                         // 512- and 1024-thread workgroups are short of VGPRs: reload instead
	constexpr bool ACCREG = M <= 13;                       // (a 1024-thread build would have no room for 16 int64 accumulators)
	uint32_t wcoef[WREG ? 16 : 1];
	if (WREG) {
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const uint32_t c = (uint32_t)window[tq + r * TPF] & 0xffffu;
			wcoef[r] = c | (c << 16);
		}
	}
	i64 acc[ACCREG ? 16 : 1];
#pragma unroll
	for (int r = 0; r < (ACCREG ? 16 : 1); r++)
		acc[r] = 0;
	i64 *avg_t = avg + (size_t)tune * N;
	const int total = nb_total * N;                        // complex samples of the tune buffer that take part

	for (int pass = p_begin; pass < p_end; pass++) {
		const uint32_t *buf = (const uint32_t *)(in + (size_t)pass * pass_stride + (size_t)tune * tune_stride);
		i64 si = 0, sq = 0;
		for (int c = tid; c < total; c += T) {
			const uint32_t w = buf[c];
			si += pw_lo(w);
			sq += pw_hi(w);
		}
		for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
		__syncthreads();
		if ((tid & 63) == 0) { red[tid >> 6] = si; red[16 + (tid >> 6)] = sq; }
		__syncthreads();
		si = 0; sq = 0;
#pragma unroll
		for (int w = 0; w < T / 64; w++) { si += red[w]; sq += red[16 + w]; }
		const i64 L = 2 * (i64)total;
		const uint32_t ave = pw_pack((int)(short)(si / L), (int)(short)(sq / (L - 1)));   // rtl_power.c:609-624 via 744-745

		for (int g = 0; g < nb_total; g += FPW) {
			const int blk = g + fid;
			const bool live = blk < nb_total;
			uint32_t v[16];
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const uint32_t w = live ? buf[blk * N + tq + r * TPF] : 0u;
				uint32_t wc;
				if (WREG) {
					wc = wcoef[r];
				} else {
					const uint32_t c = (uint32_t)window[tq + r * TPF] & 0xffffu;
					wc = c | (c << 16);
				}
				v[r] = pw_pk_mul(pw_pk_sub(w, ave), wc);                       // window, rtl_power.c:749-758
			}
			fft_reg<M, DB, true>(v, tq, xa + fid * TPF * G::XROW, xb + fid * TPF * G::XROW, twiddle, tl, ta);
			if (live) {
#pragma unroll
				for (int r = 0; r < 16; r++) {
					const i64 pw = (i64)pw_norm(v[r]);
					if (ACCREG) {
						acc[r] = PEAK ? (pw > acc[r] ? pw : acc[r]) : acc[r] + pw;
					} else {
						const unsigned bin = __brev((tq << 4) | (unsigned)r) >> (32 - M);
						if (PEAK) atomicMax((long long *)&avg_t[bin], pw);
						else if (pw) atomicAdd((unsigned long long *)&avg_t[bin], (unsigned long long)pw);
					}
				}
			}
		}
	}
	if (ACCREG) {
		// few tunes: the groups' spectra go to partial[((group * tunes + tune) * FPW + fid) * N + bin] and k_pwm_reduce folds them in
		// (every pass of a one-tune sweep lands on the same N bins: the int64 atomics were most of the launch)
		i64 *dst = partial ? partial + ((((size_t)blockIdx.y * gridDim.x + tune) * FPW + fid) << M) : nullptr;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const unsigned bin = __brev((tq << 4) | (unsigned)r) >> (32 - M);
			if (dst) dst[bin] = acc[r];
			else if (PEAK) atomicMax((long long *)&avg_t[bin], acc[r]);
			else if (acc[r]) atomicAdd((unsigned long long *)&avg_t[bin], (unsigned long long)acc[r]);
		}
	}
}

// k_pw_fftR for N = 256 ... 2048 where a tune buffer is a few groups of side-by-side transforms (the usual 16384-int16 buffer: two groups): k_pw_fft4096's
// shape instead of the two passes over the buffer above.  A thread takes ALL its samples of the pass into registers at once (NG groups x 16 values),
// remove_dc's sums come from those registers, the NEXT pass's samples are requested while this one is transformed, and the input is read once.
// Measured at the configs[2] buffer shape (599 tunes x 16384 int16 x 256 passes): the two-pass form 315-337 G bins/s for N = 256 ... 2048 -- under
// N = 4096's 485 although it has a quarter to a twelfth fewer stages per bin.
template <int M, int NG, bool PEAK>
__global__ __launch_bounds__(((1 << M) / 16 > 256) ? (1 << M) / 16 : 256) void k_pw_fftR2(
	const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int passes,
	const int *__restrict__ window, const uint32_t *__restrict__ twiddle, int ppg, i64 *__restrict__ avg, i64 *__restrict__ partial)
{
	typedef fft_geom<M> G;
	constexpr int N = G::N, TPF = G::TPF, T = TPF > 256 ? TPF : 256, FPW = T / TPF;
	constexpr bool DB = M <= 12;                           // N = 8192 (512 threads): one transpose area, a barrier between its uses
	static_assert(M >= 5 && M <= 13, "side-by-side transforms in one workgroup (N >= 32: two threads per transform), or one transform of 512 threads");
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *xa = lds, *xb = DB ? lds + T * G::XROW : lds;
	i64 *red = (i64 *)(lds + (DB ? 2 : 1) * T * G::XROW);
	uint32_t *tl = (uint32_t *)(red + 32);
	const int tid = threadIdx.x, fid = tid / TPF;
	const unsigned tq = tid % TPF;
	const int tune = blockIdx.x;
	const int p_begin = blockIdx.y * ppg, p_end = min(passes, p_begin + ppg);
	fft_tw_fill<M>(tl, twiddle, tid, T);                   // ordered by the pass loop's first __syncthreads
	unsigned ta[3][4];
	fft_tw_addr_all<M>(tq, ta);
	uint32_t wcoef[16];
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const uint32_t c = (uint32_t)window[tq + r * TPF] & 0xffffu;
		wcoef[r] = c | (c << 16);
	}
	i64 acc[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = 0;
	i64 *avg_t = avg + (size_t)tune * N;
	constexpr int L = 2 * NG * FPW * N;                    // int16 of a tune buffer that take part
	uint32_t dn[NG][16];
	if (p_begin < p_end) {
		const uint32_t *buf0 = (const uint32_t *)(in + (size_t)p_begin * pass_stride + (size_t)tune * tune_stride);
#pragma unroll
		for (int g = 0; g < NG; g++)
#pragma unroll
			for (int r = 0; r < 16; r++)
				dn[g][r] = buf0[(g * FPW + fid) * N + tq + r * TPF];
	}
	for (int pass = p_begin; pass < p_end; pass++) {
		uint32_t d[NG][16];
		int si = 0, sq = 0;                                    // 16 NG int16 per lane: far inside int32
#pragma unroll
		for (int g = 0; g < NG; g++)
#pragma unroll
			for (int r = 0; r < 16; r++) {
				d[g][r] = dn[g][r];
				si = pw_dot(d[g][r], 0x00000001u, si);
				sq = pw_dot(d[g][r], 0x00010000u, sq);
			}
		if (pass + 1 < p_end) {
			const uint32_t *bufn = (const uint32_t *)(in + (size_t)(pass + 1) * pass_stride + (size_t)tune * tune_stride);
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int r = 0; r < 16; r++)
					dn[g][r] = bufn[(g * FPW + fid) * N + tq + r * TPF];
		}
		for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
		__syncthreads();
		if ((tid & 63) == 0) { red[tid >> 6] = si; red[16 + (tid >> 6)] = sq; }
		__syncthreads();
		i64 ti64 = 0, tq64 = 0;
#pragma unroll
		for (int w = 0; w < T / 64; w++) { ti64 += red[w]; tq64 += red[16 + w]; }
		const uint32_t ave = pw_pack((int)(short)(ti64 / L), (int)(short)(tq64 / (L - 1)));   // rtl_power.c:609-624 via 744-745
#pragma unroll
		for (int g = 0; g < NG; g++) {
			uint32_t v[16];
#pragma unroll
			for (int r = 0; r < 16; r++)
				v[r] = pw_pk_mul(pw_pk_sub(d[g][r], ave), wcoef[r]);                   // window, rtl_power.c:749-758
			fft_reg<M, DB, true>(v, tq, xa + fid * TPF * G::XROW, xb + fid * TPF * G::XROW, twiddle, tl, ta);
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const i64 pw = (i64)pw_norm(v[r]);
				acc[r] = PEAK ? (pw > acc[r] ? pw : acc[r]) : acc[r] + pw;
			}
		}
	}
	// few tunes: the groups' spectra go to partial[((group * tunes + tune) * FPW + fid) * N + bin] and k_pwm_reduce folds them in
	i64 *dst = partial ? partial + ((((size_t)blockIdx.y * gridDim.x + tune) * FPW + fid) << M) : nullptr;
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const unsigned bin = __brev((tq << 4) | (unsigned)r) >> (32 - M);
		if (dst) dst[bin] = acc[r];
		else if (PEAK) atomicMax((long long *)&avg_t[bin], acc[r]);
		else if (acc[r]) atomicAdd((unsigned long long *)&avg_t[bin], (unsigned long long)acc[r]);
	}
}

// P1 for the drop-in (rtl_power.c:715-720): scanner() copies each tune's buf16 -- 599 separate mallocs of the caller -- into fft_buf.  Here the
// caller's buffers are page-locked once and their device-visible addresses sit in a table: one launch pulls every tune's capture across PCIe
// into the contiguous [tunes][buf_len] input of the scan (16-byte pieces, four on their way per lane), no host memcpy, no staging copy.
__global__ __launch_bounds__(256) void k_pw_gather_rows(const uint4 *const *__restrict__ rows, uint4 *__restrict__ out, unsigned units_per_row)
{
	const uint4 *__restrict__ src = rows[blockIdx.y];
	uint4 *__restrict__ dst = out + (size_t)blockIdx.y * units_per_row;
	const unsigned base = blockIdx.x * 1024u + threadIdx.x;
	uint4 v[4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const unsigned u = base + q * 256u;
		v[q] = u < units_per_row ? src[u] : make_uint4(0, 0, 0, 0);
	}
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const unsigned u = base + q * 256u;
		if (u < units_per_row)
			dst[u] = v[q];
	}
}

// rxgpu_scan_sync's merge in place: host row r (a tune's avg[], page-locked, device-visible) += or max= the device's accumulator row r, which is
// zeroed on the way; 16-byte units = two int64 (rtl_power.c:760-768 does the same sums one sweep at a time)
// HZ: the host rows are known to hold zeros (rxgpu_csv_dbm cleared them): not read, the accumulators are written over them
template <bool HZ>
__global__ __launch_bounds__(256) void k_pw_merge_rows(uint4 *const *__restrict__ rows, uint4 *__restrict__ acc, unsigned units_per_row, int peak)
{
	uint4 *__restrict__ dst = rows[blockIdx.y];
	uint4 *__restrict__ src = acc + (size_t)blockIdx.y * units_per_row;
	const unsigned base = blockIdx.x * 1024u + threadIdx.x;
	uint4 h[4], d[4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const unsigned u = base + q * 256u;
		h[q] = (!HZ && u < units_per_row) ? dst[u] : make_uint4(0, 0, 0, 0);
		d[q] = u < units_per_row ? src[u] : make_uint4(0, 0, 0, 0);
	}
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const unsigned u = base + q * 256u;
		if (u >= units_per_row)
			continue;
		const i64 h0 = (i64)(((unsigned long long)h[q].y << 32) | h[q].x), h1 = (i64)(((unsigned long long)h[q].w << 32) | h[q].z);
		const i64 d0 = (i64)(((unsigned long long)d[q].y << 32) | d[q].x), d1 = (i64)(((unsigned long long)d[q].w << 32) | d[q].z);
		const i64 r0 = peak ? (d0 > h0 ? d0 : h0) : h0 + d0, r1 = peak ? (d1 > h1 ? d1 : h1) : h1 + d1;
		dst[u] = make_uint4((unsigned)r0, (unsigned)((unsigned long long)r0 >> 32), (unsigned)r1, (unsigned)((unsigned long long)r1 >> 32));
		src[u] = make_uint4(0, 0, 0, 0);
	}
}

__global__ void k_pw_samples(int *samples, int tunes, int add)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < tunes)
		samples[t] += add;
}

// ------------------------------------------------------------------ P2 boxcar

// rtl_power.c:723-733.  In-place on the reference: slot k (int16 2k,2k+1) ends up holding the
// int16-wrapped sum of complex samples [k*ds, (k+1)*ds) that exist, every other position of
// the buffer is left zero.
// Only the first n_write slots of every buffer are produced: the transform reads eff_len / 2 of them (rounded up to whole FFT
// blocks), the rest of the reference's zero-filled tail is never looked at again -- writing it cost as much as reading the input.
__global__ void k_pw_boxcar(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n_bufs, int n_complex, int ds, int n_write)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n_bufs * (u64)n_write)
		return;
	const u64 b = gid / (unsigned)n_write;
	const int k = (int)(gid - b * (unsigned)n_write);
	const uint32_t *src = in + b * (u64)n_complex;
	int si = 0, sq = 0;
	const i64 first = (i64)k * ds;
	if (first < n_complex) {
		const int last = (int)min((i64)n_complex, first + ds);
		for (int c = (int)first; c < last; c++) {
			const uint32_t w = src[c];
			si += pw_lo(w);
			sq += pw_hi(w);
		}
	}
	out[b * (u64)n_complex + k] = pw_pack(si, sq);
}

// When every buffer holds a whole number of boxcar windows, P2 over the concatenated buffers IS rx_fm's low_pass without
// scale and rotation: the fast decimator of fm_kernels.hip (coalesced 16-byte loads, wave prefix scan) does the sums at
// HBM rate and leaves the first window ending in each of its spans in head/tail form; this finishes those.
// dc_sums != NULL (rxk_pw_boxcar_sums in front): the span's four wave sums (wave_sums[span * 4 + wave] = {I, Q}) and its seam output join the
// remove_dc sums of the span's buffer (spans_per_buf whole spans per buffer)
__global__ void k_pw_boxcar_seams(uint32_t *__restrict__ lp, const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail,
                                  u64 n_spans, u64 M, int ds, i64 *__restrict__ dc_sums, const int2 *__restrict__ wave_sums, unsigned spans_per_buf)
{
	const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_spans)
		return;
	const u64 m = (g << RXK_DEC_SPAN_LOG2) / (u64)ds;          // windows completed before the span = the first one ending in it
	i64 si = 0, sq = 0;
	if (m < M) {
		const uint32_t v = pw_pk_add(g ? tail[g - 1] : 0u, head[g]);
		lp[m] = v;
		si = pw_lo(v);
		sq = pw_hi(v);
	}
	if (dc_sums) {
#pragma unroll
		for (int w = 0; w < 4; w++) {
			const int2 p = wave_sums[g * 4 + w];
			si += p.x;
			sq += p.y;
		}
		const u64 buf = g / spans_per_buf;
		atomicAdd((unsigned long long *)&dc_sums[2 * buf], (unsigned long long)si);
		atomicAdd((unsigned long long *)&dc_sums[2 * buf + 1], (unsigned long long)sq);
	}
}
extern "C" int rxk_pw_boxcar_seams(void *stream, uint32_t *lp, const uint32_t *head, const uint32_t *tail, unsigned long long T, int ds,
                                   long long *dc_sums, const int *wave_sums, unsigned spans_per_buf)
{
	const u64 n_spans = (T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN;
	hipLaunchKernelGGL(k_pw_boxcar_seams, dim3((unsigned)((n_spans + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lp, head, tail,
	                   n_spans, T / (u64)ds, ds, (i64 *)dc_sums, (const int2 *)wave_sums, spans_per_buf);
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------ P3 fifth_order (stateless)

// One pass of rtl_power.c:582-607 on both components (downsample_iq 656-662), out of place:
// the in-place original never overwrites a sample it still has to read, so every output is a
// function of the pass input.  n_in complex samples in (int16 length 2*n_in for I, 2*n_in-1 for
// Q: both yield outputs k with 4k < length, i.e. k < ceil(n_in/2)).
__global__ void k_pw_fifth(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n_bufs,
                           int n_in, int in_stride, int out_stride)
{
	const int n_out = (n_in + 1) / 2;
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n_bufs * (u64)n_out)
		return;
	const u64 b = gid / (unsigned)n_out;
	const int k = (int)(gid - b * (unsigned)n_out);
	const uint32_t *src = in + b * (u64)in_stride;
	int oi, oq;
	if (k >= 5) {
		int ti[6], tq[6];
#pragma unroll
		for (int t = 0; t < 6; t++) {
			const uint32_t w = src[2 * k - 5 + t];
			ti[t] = pw_lo(w); tq[t] = pw_hi(w);
		}
		oi = (ti[0] + (ti[1] + ti[4]) * 5 + (ti[2] + ti[3]) * 10 + ti[5]) >> 4;
		oq = (tq[0] + (tq[1] + tq[4]) * 5 + (tq[2] + tq[3]) * 10 + tq[5]) >> 4;
	} else {
		// ease-in, rtl_power.c:587-597 and the first two loop turns, which re-read sample 5
		int si[9], sq[9];
#pragma unroll
		for (int t = 0; t < 9; t++) {
			const uint32_t w = (t < n_in) ? src[t] : 0u;
			si[t] = pw_lo(w); sq[t] = pw_hi(w);
		}
#define EASE(s, o) do { \
		const int a = s[0], b_ = s[1], c = s[2], d = s[3], e = s[4], f = s[5]; \
		switch (k) { \
		case 0: o = ((a + b_) * 10 + (c + d) * 5 + d + f) >> 4; break; \
		case 1: o = ((b_ + c) * 10 + (a + d) * 5 + e + f) >> 4; break; \
		case 2: o = (a + (b_ + e) * 5 + (c + d) * 10 + f) >> 4; break; \
		case 3: o = (c + (d + f) * 5 + (e + f) * 10 + s[6]) >> 4; break; \
		default: o = (e + (f + s[7]) * 5 + (f + s[6]) * 10 + s[8]) >> 4; break; \
		} } while (0)
		EASE(si, oi);
		EASE(sq, oq);
#undef EASE
	}
	out[b * (u64)out_stride + k] = pw_pack(oi, oq);
}

// rtl_power.c:626-654: samples 0..8 pass through, sample t >= 9 becomes FIR(s[t-9..t-1]) >> 15
__global__ void k_pw_droop(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n_bufs, int n,
                           int stride, const int *__restrict__ fir)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n_bufs * (u64)n)
		return;
	const u64 b = gid / (unsigned)n;
	const int t = (int)(gid - b * (unsigned)n);
	const uint32_t *src = in + b * (u64)stride;
	if (t < 9) {
		out[b * (u64)stride + t] = src[t];
		return;
	}
	int hi[9], hq[9];
#pragma unroll
	for (int j = 0; j < 9; j++) {
		const uint32_t w = src[t - 9 + j];
		hi[j] = pw_lo(w); hq[j] = pw_hi(w);
	}
	const int f1 = fir[1], f2 = fir[2], f3 = fir[3], f4 = fir[4], f5 = fir[5];
	// 24-bit multiplies: sums of two int16 and cic_9_tables coefficients (< 2^17) fit; low 32 bits of the product = the wrapping int
	const int si = __mul24(hi[0] + hi[8], f1) + __mul24(hi[1] + hi[7], f2) + __mul24(hi[2] + hi[6], f3) + __mul24(hi[3] + hi[5], f4) + __mul24(hi[4], f5);
	const int sq = __mul24(hq[0] + hq[8], f1) + __mul24(hq[1] + hq[7], f2) + __mul24(hq[2] + hq[6], f3) + __mul24(hq[3] + hq[5], f4) + __mul24(hq[4], f5);
	out[b * (u64)stride + t] = pw_pack(si >> 15, sq >> 15);
}

// the same, outputs 4k .. 4k+3 of buffer blockIdx.y per thread: they need samples 4k-9 .. 4k+2 -- three aligned 16-byte loads (4k-12 .. 4k-1 + one
// more) -- and leave as one 16-byte store.  The one-output kernel above spent its time on nine loads and a 64-bit division per sample
// (0.17 VALU instructions per SIMD-cycle, waves waiting 70 % of the time).
__global__ __launch_bounds__(256) void k_pw_droop4(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int n, int stride,
                                                   const int *__restrict__ fir)
{
	const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);         // group of four outputs
	if (4 * k >= n)
		return;
	const uint32_t *src = in + (size_t)blockIdx.y * (size_t)stride;
	uint32_t *dst = out + (size_t)blockIdx.y * (size_t)stride;
	const int t0 = 4 * k;
	// samples t0 - 12 .. t0 + 3 (the last group's own four are also what passes through for t < 9)
	uint32_t w[16];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const int at = t0 - 12 + 4 * q;
		const uint4 v = at >= 0 ? *reinterpret_cast<const uint4 *>(src + at) : make_uint4(0u, 0u, 0u, 0u);
		w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
	}
	const int f1 = fir[1], f2 = fir[2], f3 = fir[3], f4 = fir[4], f5 = fir[5];
	uint32_t o[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int t = t0 + j;
		if (t < 9) {
			o[j] = w[12 + j];                                         // rtl_power.c:626-654: samples 0..8 pass through
			continue;
		}
		// s[t-9+m] = w[3 + j + m], m = 0..8
		int hi[9], hq[9];
#pragma unroll
		for (int m = 0; m < 9; m++) {
			hi[m] = pw_lo(w[3 + j + m]); hq[m] = pw_hi(w[3 + j + m]);
		}
		const int si = __mul24(hi[0] + hi[8], f1) + __mul24(hi[1] + hi[7], f2) + __mul24(hi[2] + hi[6], f3) + __mul24(hi[3] + hi[5], f4) + __mul24(hi[4], f5);
		const int sq = __mul24(hq[0] + hq[8], f1) + __mul24(hq[1] + hq[7], f2) + __mul24(hq[2] + hq[6], f3) + __mul24(hq[3] + hq[5], f4) + __mul24(hq[4], f5);
		o[j] = pw_pack(si >> 15, sq >> 15);
	}
	*reinterpret_cast<uint4 *>(dst + t0) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------ P9 rms_power sums

__global__ __launch_bounds__(256) void k_pw_rms_sums(const int16_t *__restrict__ in, size_t n_bufs, int buf_len,
                                                     i64 *__restrict__ t_out, i64 *__restrict__ p_out)
{
	__shared__ i64 red[8];
	const size_t b = blockIdx.x;
	if (b >= n_bufs)
		return;
	const int16_t *src = in + b * (size_t)buf_len;
	i64 t = 0, p = 0;
	int i0 = 0;
	if ((((size_t)src) & 15) == 0) {
		// 16-byte loads, four on their way per lane; per dword (two int16) one v_dot2 for the sum and one for the squares (x0^2 + x1^2 <= 2^31: an
		// unsigned 32-bit value, added into 64 bits).  The first version read one int16 per load and multiplied in 64 bits: 1.97 TB/s of input
		const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
		const int nv = buf_len / 8;
		for (int v = threadIdx.x; v < nv; v += 1024) {
			uint4 w[4];
#pragma unroll
			for (int u = 0; u < 4; u++)
				w[u] = v + 256 * u < nv ? s4[v + 256 * u] : make_uint4(0, 0, 0, 0);
			int ts = 0;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t d[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
				for (int k = 0; k < 4; k++) {
					ts = pw_dot(d[k], 0x00010001u, ts);                       // 32 int16 per turn: far inside int32
					p += (i64)pw_norm(d[k]);
				}
			}
			t += ts;
		}
		i0 = nv * 8;
	}
	for (int i = i0 + threadIdx.x; i < buf_len; i += 256) {
		const i64 s = src[i];
		t += s;
		p += s * s;
	}
	for (int off = 32; off; off >>= 1) { t += __shfl_down(t, off); p += __shfl_down(p, off); }
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = t; red[4 + (threadIdx.x >> 6)] = p; }
	__syncthreads();
	if (threadIdx.x == 0) {
		t_out[b] = red[0] + red[1] + red[2] + red[3];
		p_out[b] = red[4] + red[5] + red[6] + red[7];
	}
}

// rtl_power.c:418-428: the fp64 dc correction and the accumulate, pass after pass per tune.
// Plain IEEE fp64 ops in the reference's order (this file is built with -ffp-contract=off).
__global__ void k_pw_rms_apply(const i64 *__restrict__ t_in, const i64 *__restrict__ p_in, int passes, int tunes,
                               int buf_len, int peak_hold, i64 *__restrict__ avg, int *__restrict__ samples)
{
	const int tune = blockIdx.x * blockDim.x + threadIdx.x;
	if (tune >= tunes)
		return;
	i64 a = avg[tune];
	for (int pass = 0; pass < passes; pass++) {
		const i64 t = t_in[(size_t)pass * tunes + tune];
		i64 p = p_in[(size_t)pass * tunes + tune];
		const double dc = (double)t / (double)buf_len;
		const double lhs = (double)(t * 2) * dc;
		const double rhs = dc * dc * (double)buf_len;
		const double err = lhs - rhs;
		p -= (i64)round(err);
		a = peak_hold ? (p > a ? p : a) : a + p;
	}
	avg[tune] = a;
	samples[tune] += passes;
}

// ------------------------------------------------------------------ launchers

#define LAUNCH_RET() return (int)hipGetLastError()

// ------------------------------------------------------------------ P4-P8, N > 2^15: fix_fft in global memory

// The reference takes FFT lengths up to 2^21 (rtl_power.c:485); beyond 2^15 a block no longer fits LDS, so the same
// radix-2 network (bit-reversed load, rtl_power.c:275-290, then one launch per stage, 291-318) runs on a scratch copy
// in HBM, many blocks side by side.  Correct and plain -- these lengths are for 1-Hz-class bins on a narrow range.
// remove_dc's sums (rtl_power.c:609-624 via 744-745) of every (pass, tune) buffer, in front of the transform.  Round 4: a buffer is cut into
// `slices` pieces, one workgroup each, 16-byte loads with four in flight per lane, int64 atomics into sums[2 * (pass * tunes + tune)]
// (zeroed by the launcher), and k_pwb_dc_fin divides.  One workgroup per buffer with dword loads moved 0.64 TB/s and was HALF of the
// N = 2^18 launch (420 of 790 us, rocprofv3).
__global__ __launch_bounds__(256) void k_pwb_dc(const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int tunes,
                                                int eff_len, int slices, i64 *__restrict__ sums)
{
	__shared__ i64 red[8];
	// the buffers in REVERSE order: what this pass touched last is what the transform's first workgroups read first (Infinity Cache, A/B round 6)
	const int ptr_ = blockIdx.x / slices, sl = blockIdx.x - ptr_ * slices, tid = threadIdx.x;
	const int pt = (int)(gridDim.x / slices) - 1 - ptr_;
	const int pass = pt / tunes, tune = pt - pass * tunes;
	const uint32_t *buf = (const uint32_t *)(in + (size_t)pass * pass_stride + (size_t)tune * tune_stride);
	const int L = eff_len;
	const int ci = (L + 1) / 2, cq = L / 2;                              // complex samples whose I / Q half takes part
	// slice: whole 16-byte vectors [v0, v1) of the buffer; the ragged end (ci not a multiple of 4, or an odd L) goes to the last slice, dword by dword
	const int nv = cq / 4, per = (nv + slices - 1) / slices;
	const int v0 = sl * per, v1 = min(nv, v0 + per);
	i64 si = 0, sq = 0;
	const bool aligned = (((size_t)buf) & 15) == 0;
	if (aligned) {
		const uint4 *b4 = reinterpret_cast<const uint4 *>(buf);
		int v = v0 + tid;
		for (; v + 768 < v1; v += 1024) {
			const uint4 a = b4[v], b = b4[v + 256], c = b4[v + 512], d = b4[v + 768];
			int i32 = 0, q32 = 0;                                        // 16 samples of int16: far from int32's range
#define ACC4(VV) do { i32 = pw_dot((VV).x, 1u, i32); q32 = pw_dot((VV).x, 0x10000u, q32); i32 = pw_dot((VV).y, 1u, i32); q32 = pw_dot((VV).y, 0x10000u, q32); \
			     i32 = pw_dot((VV).z, 1u, i32); q32 = pw_dot((VV).z, 0x10000u, q32); i32 = pw_dot((VV).w, 1u, i32); q32 = pw_dot((VV).w, 0x10000u, q32); } while (0)
			ACC4(a); ACC4(b); ACC4(c); ACC4(d);
			si += i32; sq += q32;
		}
		for (; v < v1; v += 256) {
			const uint4 a = b4[v];
			int i32 = 0, q32 = 0;
			ACC4(a);
#undef ACC4
			si += i32; sq += q32;
		}
	}
	const int c_lo = aligned ? ((sl == slices - 1) ? nv * 4 : ci) : (int)((i64)ci * sl / slices);
	const int c_hi = aligned ? ci : (int)((i64)ci * (sl + 1) / slices);
	for (int c = c_lo + tid; c < c_hi; c += 256) {
		const uint32_t w = buf[c];
		si += pw_lo(w);
		if (c < cq) sq += pw_hi(w);
	}
	for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
	if ((tid & 63) == 0) { red[tid >> 6] = si; red[4 + (tid >> 6)] = sq; }
	__syncthreads();
	if (tid == 0) {
		si = red[0] + red[1] + red[2] + red[3];
		sq = red[4] + red[5] + red[6] + red[7];
		atomicAdd((unsigned long long *)&sums[2 * (size_t)pt], (unsigned long long)si);
		atomicAdd((unsigned long long *)&sums[2 * (size_t)pt + 1], (unsigned long long)sq);
	}
}

// (samples != NULL: the launch's `samples[tune] += add`, rtl_power.c:769, rides along -- one launch of a few threads less per scan)
__global__ void k_pwb_dc_fin(const i64 *__restrict__ sums, int n, int eff_len, int *__restrict__ dc, int *__restrict__ samples = nullptr, int tunes = 0, int add = 0)
{
	const int pt = blockIdx.x * blockDim.x + threadIdx.x;
	if (samples && pt < tunes)
		samples[pt] += add;
	if (pt >= n)
		return;
	const i64 L = eff_len;
	dc[2 * pt] = (int)(short)(sums[2 * pt] / L);                                                // remove_dc, rtl_power.c:609-624
	dc[2 * pt + 1] = (L > 1) ? (int)(short)(sums[2 * pt + 1] / (L - 1)) : 0;
}

// dc: 2 ints per (pass, tune), then -- 16-byte aligned -- 2 int64 per (pass, tune) of scratch for the sums (RXK_PW_DC_BYTES per pair)
extern "C" long long *rxk_pw_dc_sums(int *dc, size_t n_pass_tunes)
{
	return (long long *)(dc + ((2 * n_pass_tunes + 3) & ~(size_t)3));
}

static void pwb_dc(hipStream_t s, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes, int eff_len, int *dc,
                   int *samples = nullptr, int add = 0)
{
	const size_t n = (size_t)passes * tunes;
	i64 *sums = (i64 *)rxk_pw_dc_sums(dc, n);
	(void)hipMemsetAsync(sums, 0, n * 16, s);
	// enough workgroups to fill the chip (about eight per CU), slices of at least 16 KiB
	int slices = (int)((2048 + n - 1) / n);
	const int max_slices = eff_len / 8192 > 0 ? eff_len / 8192 : 1;
	if (slices > max_slices) slices = max_slices;
	if (slices < 1) slices = 1;
	hipLaunchKernelGGL(k_pwb_dc, dim3((unsigned)(n * slices)), dim3(256), 0, s, in, tune_stride, pass_stride, tunes, eff_len, slices, sums);
	hipLaunchKernelGGL(k_pwb_dc_fin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sums, (int)n, eff_len, dc, samples, tunes, add);
}

// block q = (pass * tunes + tune) * nbpt + blk;  this launch covers blocks q0 .. q0+nq
__global__ void k_pwb_load(const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int tunes, int nbpt, int bin_e,
                           int eff_len, const int *__restrict__ window, const int *__restrict__ dc, size_t q0, size_t nq,
                           uint32_t *__restrict__ scratch)
{
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ql = gid >> bin_e;
	if (ql >= nq)
		return;
	const int n = 1 << bin_e, j = (int)(gid & (size_t)(n - 1));
	const size_t q = q0 + ql, pt = q / (size_t)nbpt;
	const int blk = (int)(q - pt * (size_t)nbpt);
	const size_t pass = pt / (size_t)tunes, tune = pt - pass * (size_t)tunes;
	const uint32_t *buf = (const uint32_t *)(in + pass * pass_stride + tune * tune_stride);
	const int ci = (eff_len + 1) / 2, cq = eff_len / 2;
	const int c = blk * n + j;
	const uint32_t w = buf[c];
	int vi = pw_lo(w), vq = pw_hi(w);
	if (c < ci) vi = (int)(short)(vi - dc[2 * pt]);
	if (c < cq) vq = (int)(short)(vq - dc[2 * pt + 1]);
	const int coef = window[j];                              // rtl_power.c:749-758: int32 product, int16 truncation
	scratch[(ql << bin_e) + (__brev((unsigned)j) >> (32 - bin_e))] = pw_pack(vi * coef, vq * coef);
}

__global__ void k_pwb_stage(uint32_t *__restrict__ scratch, size_t nq, int bin_e, int s, const uint32_t *__restrict__ twiddle)
{
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ql = gid >> (bin_e - 1);
	if (ql >= nq)
		return;
	const unsigned b = (unsigned)(gid & (((size_t)1 << (bin_e - 1)) - 1));
	const unsigned half = 1u << s, t = b & (half - 1);
	const unsigned lo_i = ((b >> s) << (s + 1)) | t;
	uint32_t *x = scratch + (ql << bin_e);
	uint32_t lo = x[lo_i], hi = x[lo_i + half];
	butterfly(lo, hi, twiddle[(size_t)t << (bin_e - 1 - s)]);
	x[lo_i] = lo;
	x[lo_i + half] = hi;
}

__global__ void k_pwb_acc(const uint32_t *__restrict__ scratch, int tunes, int nbpt, int bin_e, size_t q0, size_t nq, int peak_hold,
                          i64 *__restrict__ avg)
{
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ql = gid >> bin_e;
	if (ql >= nq)
		return;
	const size_t k = gid & (((size_t)1 << bin_e) - 1);
	const size_t pt = (q0 + ql) / (size_t)nbpt, tune = pt % (size_t)tunes;
	const uint32_t w = scratch[gid];
	const i64 re = pw_lo(w), im = pw_hi(w);
	const i64 pw = re * re + im * im;
	i64 *a = avg + (tune << bin_e) + k;
	if (peak_hold) atomicMax((long long *)a, pw);
	else atomicAdd((unsigned long long *)a, (unsigned long long)pw);
}

// ------------------------------------------------------------------ N = 2^14 .. 2^21: the register-blocked transform in two to four launches
//
// N/16 threads per transform is 1024 or 2048: one workgroup would have 128 VGPRs per lane (the single-kernel build spilled and lost to
// the LDS radix-2 kernel) or does not exist.  But after the FIRST radix-16 pass -- stages 0-3 on the top four index bits, which a thread
// does alone on its 16 values n = col + (N/16) r -- the transform falls apart into 16 independent sub-transforms of N/16 points: every
// later stage pairs indices that differ below bit M-4.  So:
//   k_pwm_head  one thread per column: remove_dc + window + stages 0-3, written back in natural order to a scratch copy in HBM
//               (8 bytes of extra traffic per bin);
//   k_pwm_tail  N/256 threads per sub-transform, 256 / (N/256) of them side by side in a workgroup: the remaining passes of
//               fft_device.h (LDS transposes among 64 or 128 threads), |X|^2 into 16 int64 accumulators per thread over every pass
//               of the launch, one atomic per bin at the end.
// Block q = (pass * tunes + tune) * nbpt + blk as in the stage-per-launch path below; a launch covers whole passes.
template <int M>
__global__ __launch_bounds__(256) void k_pwm_head(const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int tunes, int nbpt,
                                                 const int *__restrict__ window, const uint32_t *__restrict__ twiddle, const int *__restrict__ dc,
                                                 size_t q0, size_t nq, uint32_t *__restrict__ scratch)
{
	constexpr int N = 1 << M, TPF = N / 16;
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ql = gid / TPF;
	if (ql >= nq)
		return;
	const unsigned col = (unsigned)(gid % TPF);
	const size_t q = q0 + ql, pt = q / (size_t)nbpt;
	const int blk = (int)(q - pt * (size_t)nbpt);
	const size_t pass = pt / (size_t)tunes, tune = pt - pass * (size_t)tunes;
	const uint32_t *buf = (const uint32_t *)(in + pass * pass_stride + tune * tune_stride) + (size_t)blk * N;
	const uint32_t ave = pw_pack(dc[2 * pt], dc[2 * pt + 1]);
	uint32_t v[16];
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const uint32_t c = (uint32_t)window[col + r * TPF] & 0xffffu;
		v[r] = pw_pk_mul(pw_pk_sub(buf[col + r * TPF], ave), c | (c << 16));          // remove_dc + window, rtl_power.c:744-758
	}
	fft_pass<M, 0>(v, twiddle, col);
	uint32_t *dst = scratch + (ql << M);
#pragma unroll
	for (int r = 0; r < 16; r++)
		dst[col + r * TPF] = v[r];
}

// A further radix-16 pass through HBM for N > 2^16 (stages 4 PASS .. 4 PASS + 3 on the scratch copy, in place): after pass 0 the
// transform is 16 independent sub-transforms of N/16 points, after this one 256 of N/256, and so on until a sub-transform's N/16^H
// points fit the 256 threads of a workgroup (<= 4096 points: H = 1 up to 2^16, 2 up to 2^20, 3 for 2^21 -- the reference's limit,
// rtl_power.c:485).  A thread takes the 16 values of the pass's index field, 2^f(PASS) apart: consecutive threads, consecutive dwords.
template <int M, int PASS>
__global__ __launch_bounds__(256) void k_pwm_head_mid(uint32_t *__restrict__ scratch, size_t nq, const uint32_t *__restrict__ twiddle)
{
	typedef fft_geom<M> G;
	constexpr int TPF = (1 << M) / 16, F = G::f(PASS);
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ql = gid / TPF;
	if (ql >= nq)
		return;
	const unsigned tq = (unsigned)(gid % TPF);
	uint32_t *x = scratch + (ql << M) + ((size_t)(tq >> F) << (F + 4)) + (tq & ((1u << F) - 1u));
	uint32_t v[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
		v[r] = x[(size_t)r << F];
	fft_pass<M, PASS>(v, twiddle, tq);
#pragma unroll
	for (int r = 0; r < 16; r++)
		x[(size_t)r << F] = v[r];
}

// N >= 2^17 needs two radix-16 passes in front of the tail: done by k_pwm_head + k_pwm_head_mid the block crosses HBM twice between them
// (pass 0 out, pass 1 in).  Here both run in one launch: a thread takes the 16 values of pass 0 for TWO neighbouring columns (8-byte loads:
// sixteen threads cover a whole 128-byte line), the workgroup -- 16 values of the pass-1 field x 32 columns that share everything above
// and below it -- transposes 16 x 16 per column through LDS (pass 0 leaves a thread the top-field values of one pass-1 field value, pass 1
// wants the pass-1 field values of one top-field value), pass 1, natural-order store.  One trip through the scratch copy less.
template <int M>
__global__ __launch_bounds__(256) void k_pwm_head2(const int16_t *__restrict__ in, size_t tune_stride, size_t pass_stride, int tunes, int nbpt,
                                                  const int *__restrict__ window, const uint32_t *__restrict__ twiddle, const int *__restrict__ dc,
                                                  size_t q0, size_t nq, uint32_t *__restrict__ scratch)
{
	typedef fft_geom<M> G;
	constexpr int N = 1 << M, TPF = N / 16, F1 = G::f(1);           // F1 = M - 8: the bits below the pass-1 field
	static_assert(M >= 17 && F1 >= 5, "two head passes: N >= 2^17");
	constexpr unsigned WGB = (1u << F1) / 32u;                      // workgroups per block: 32 columns each
	constexpr int MAT = 16 * 17 + 1;                                // a column's 16 x 16 matrix, rows padded to 17, matrices one dword apart
	__shared__ uint32_t xch[32 * MAT];
	const size_t ql = blockIdx.x / WGB;
	if (ql >= nq)
		return;
	const unsigned a = threadIdx.x >> 4, l = threadIdx.x & 15u;
	const unsigned low = (blockIdx.x % WGB) * 32u + 2u * l;         // this thread's two columns: low, low + 1
	const size_t q = q0 + ql, pt = q / (size_t)nbpt;
	const int blk = (int)(q - pt * (size_t)nbpt);
	const size_t pass = pt / (size_t)tunes, tune = pt - pass * (size_t)tunes;
	const uint32_t *buf = (const uint32_t *)(in + pass * pass_stride + tune * tune_stride) + (size_t)blk * N;
	const uint32_t ave = pw_pack(dc[2 * pt], dc[2 * pt + 1]);
	const unsigned tq = (a << F1) | low;                            // pass 0: a = the pass-1 field's value; pass 1: a = the top field's
	uint32_t v0[16], v1[16];
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const uint2 x = *reinterpret_cast<const uint2 *>(buf + tq + r * TPF);
		const int2 w = *reinterpret_cast<const int2 *>(window + tq + r * TPF);
		const uint32_t c0 = (uint32_t)w.x & 0xffffu, c1 = (uint32_t)w.y & 0xffffu;
		v0[r] = pw_pk_mul(pw_pk_sub(x.x, ave), c0 | (c0 << 16));     // remove_dc + window, rtl_power.c:744-758
		v1[r] = pw_pk_mul(pw_pk_sub(x.y, ave), c1 | (c1 << 16));
	}
	fft_pass<M, 0>(v0, twiddle, tq);
	fft_pass<M, 0>(v1, twiddle, tq + 1);
#pragma unroll
	for (int r = 0; r < 16; r++) {
		xch[(2 * l) * MAT + a * 17 + r] = v0[r];
		xch[(2 * l + 1) * MAT + a * 17 + r] = v1[r];
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < 16; r++) {
		v0[r] = xch[(2 * l) * MAT + r * 17 + a];
		v1[r] = xch[(2 * l + 1) * MAT + r * 17 + a];
	}
	fft_pass<M, 1>(v0, twiddle, tq);
	fft_pass<M, 1>(v1, twiddle, tq + 1);
	uint32_t *dst = scratch + (ql << M) + ((size_t)a << (M - 4)) + low;
#pragma unroll
	for (int r = 0; r < 16; r++)
		*reinterpret_cast<uint2 *>(dst + ((size_t)r << F1)) = make_uint2(v0[r], v1[r]);
}

// the passes H .. P-1 of one thread's 16 values, an LDS transpose between two of them.  The thread's twiddles (15 per pass) do not
// change from one block of the launch to the next: they are loaded ONCE, in front of the pass loop, into twr[PASS - H] (round 4; they
// used to be 15 global loads per pass and block, waited for right behind each barrier)
template <int M, int PASS, int H>
__device__ __forceinline__ void pwm_tail_twiddles(uint32_t (&twr)[fft_geom<M>::P - H][15], const uint32_t *__restrict__ tw, unsigned tq)
{
	fft_tw_regs<M, PASS>(twr[PASS - H], tw, tq);
	if constexpr (PASS + 1 < fft_geom<M>::P)
		pwm_tail_twiddles<M, PASS + 1, H>(twr, tw, tq);
}
// A sub-transform has N / 16^(H+1) threads; they only ever exchange with each other (the later passes pair indices inside the sub-transform, and
// fft_exchange's rows are the rows of the threads that read them next).  With at most 64 of them -- N = 2^14, 2^17, 2^18, 2^21 -- they are lanes of
// ONE wave and the four waves of the workgroup need not wait for each other three times per transform: wave-level ordering instead of s_barrier.
template <int M, int PASS, int H>
__device__ __forceinline__ void pwm_tail_passes(uint32_t (&v)[16], uint32_t *lds, const uint32_t (&twr)[fft_geom<M>::P - H][15], unsigned tq, unsigned row0, bool first)
{
	typedef fft_geom<M> G;
	constexpr int XT = ((1 << M) >> (4 * H)) / 16;
	if (!first)
		fft_sync_n<XT>();                                            // the transpose area: the reads of the pass before
	fft_pass_regs<M, PASS>(v, twr[PASS - H]);
	if constexpr (PASS + 1 < G::P) {
		fft_exchange<M, PASS, XT>(v, lds, tq, row0);
		pwm_tail_passes<M, PASS + 1, H>(v, lds, twr, tq, row0, false);
	}
}

// grid: x = (tune * nbpt + blk) * WPB + workgroup of the block (WPB = N/4096 workgroups of 256 threads per transform), y = group of
// passes; the launch's passes are p0 .. p0 + np.  H = radix-16 passes already done through HBM (k_pwm_head, k_pwm_head_mid).
// partial != NULL: the accumulators go, without atomics, to partial[((group * tunes + tune) * nbpt + blk) * N + bin] and k_pwm_reduce folds
// them into avg -- with one tune every pass of a sweep lands on the same N bins, and int64 atomics on a few thousand addresses were
// three quarters of this kernel's time
// 129-151 VGPRs: three waves per SIMD.  Forcing four (__launch_bounds__(256, 4): 128 VGPRs, 2-25 dwords of twiddles spilled to scratch) LOSES -- round 6,
// A/B of two builds on one box (tools/pw_big_time.py): N = 2^16 163 -> 128, 2^18 182 -> 164, 2^20 159 -> 129 G bins/s, 2^17 / 2^21 (two spilled dwords) level.
template <int M, bool PEAK, int H = 1>
__global__ __launch_bounds__(256) void k_pwm_tail(const uint32_t *__restrict__ scratch, int tunes, int nbpt, int p0, int np, int ppg,
                                                 const uint32_t *__restrict__ twiddle, i64 *__restrict__ avg, i64 *__restrict__ partial)
{
	typedef fft_geom<M> G;
	constexpr int N = 1 << M, WPB = N / 4096, F = G::f(H);
	static_assert(M - 4 * H <= 12 && H < G::P, "a sub-transform has to fit the workgroup");
	__shared__ __attribute__((aligned(16))) uint32_t lds[256 * G::XROW];
	const int tid = threadIdx.x;
	const unsigned sg = blockIdx.x % WPB, tb = blockIdx.x / WPB;         // tb = tune * nbpt + blk
	const unsigned tune = tb / (unsigned)nbpt, blk = tb - tune * (unsigned)nbpt;
	const unsigned tq = sg * 256u + (unsigned)tid;                   // this thread's index in the whole transform's N/16
	const unsigned row0 = sg * 256u;                                 // fft_exchange addresses rows by tq: this workgroup's are row0 ..
	uint32_t twr[G::P - H][15];
	pwm_tail_twiddles<M, H, H>(twr, twiddle, tq);
	i64 acc[16];
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = 0;
	const int pb = blockIdx.y * ppg, pe = min(np, pb + ppg);
	// block index inside the launch's scratch: ((pass * tunes + tune) * nbpt + blk); a thread's 16 values are the index field of pass H,
	// 2^F apart.  The next pass's values are on their way while this one is transformed.
	const size_t pass_step = ((size_t)tunes * (size_t)nbpt) << M;
	const uint32_t *src = scratch + ((((size_t)pb * tunes + tune) * (size_t)nbpt + blk) << M) + ((size_t)(tq >> F) << (F + 4)) + (tq & ((1u << F) - 1u));
	uint32_t nxt[16];
	if (pb < pe) {
#pragma unroll
		for (int x = 0; x < 16; x++)
			nxt[x] = src[(size_t)x << F];
	}
	for (int pl = pb; pl < pe; pl++) {
		uint32_t v[16];
#pragma unroll
		for (int x = 0; x < 16; x++)
			v[x] = nxt[x];
		src += pass_step;
		if (pl + 1 < pe) {
#pragma unroll
			for (int x = 0; x < 16; x++)
				nxt[x] = src[(size_t)x << F];
		}
		fft_sync_n<((1 << M) >> (4 * H)) / 16>();                      // the previous transform's reads of the transpose area
		pwm_tail_passes<M, H, H>(v, lds, twr, tq, row0, true);
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const i64 pw = (i64)pw_norm(v[r]);
			acc[r] = PEAK ? (pw > acc[r] ? pw : acc[r]) : acc[r] + pw;
		}
	}
	(void)p0;
	if (partial) {
		// in the ORDER THE THREADS HOLD THEM (thread tq's sixteen values side by side: whole 128-byte lines) -- k_pwm_reduce<true> applies the bit
		// reversal when it folds the groups into avg.  Stored at their bins, a workgroup's values lie N/4096 bins apart: 8-byte pieces of as many
		// lines, which cost the write path four times their bytes (PMC: 268 MB written per 67 MB of spectra at N = 2^14)
		i64 *dst = partial + ((((size_t)blockIdx.y * tunes + tune) * (size_t)nbpt + blk) << M) + ((size_t)tq << 4);
#pragma unroll
		for (int r = 0; r < 16; r += 2)
			*reinterpret_cast<longlong2 *>(dst + r) = make_longlong2(acc[r], acc[r + 1]);
		return;
	}
	i64 *avg_t = avg + ((size_t)tune << M);
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const unsigned bin = __brev((tq << 4) | (unsigned)r) >> (32 - M);
		if (PEAK) atomicMax((long long *)&avg_t[bin], acc[r]);
		else if (acc[r]) atomicAdd((unsigned long long *)&avg_t[bin], (unsigned long long)acc[r]);
	}
}

// avg[tune][bin] (+= | max=) over the groups and blocks of partial: x over (tune, bin), y over 16 slices of the groups (a thread's
// loads are independent and coalesced across the workgroup; 16 atomics per bin instead of one per pass)
// PERM: partial holds every spectrum in the tail kernel's thread order (entry i = the value of bin bit_reverse(i)): thread i still reads entry i of
// every group -- coalesced -- and the permutation goes into the ONE atomic per bin and launch
template <bool PERM>
__global__ void k_pwm_reduce(const i64 *__restrict__ partial, int tunes, int nbpt, int groups, int bin_e, int peak_hold, i64 *__restrict__ avg)
{
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= ((size_t)tunes << bin_e))
		return;
	const size_t tune = gid >> bin_e, bin = gid & (((size_t)1 << bin_e) - 1);
	const size_t out = PERM ? (tune << bin_e) + (size_t)(__brev((unsigned)bin) >> (32 - bin_e)) : gid;
	const int per = (groups + (int)gridDim.y - 1) / (int)gridDim.y, g0 = (int)blockIdx.y * per, g1 = min(groups, g0 + per);
	if (g0 >= g1)
		return;
	// (every entry is a sum or a maximum of |X|^2: none is negative, 0 is the identity of both folds)
	i64 a = 0;
	const size_t gstep = ((size_t)tunes * (size_t)nbpt) << bin_e;
	const i64 *src = partial + ((((size_t)g0 * tunes + tune) * (size_t)nbpt) << bin_e) + bin;
	int g = g0;
	// eight groups' entries on their way at once: one dependent load per group left a thread waiting a memory latency per group (25 of the
	// 366 us of an N = 2^18 launch went to this kernel)
	for (; g + 8 <= g1; g += 8, src += 8 * gstep)
		for (int b = 0; b < nbpt; b++) {
			i64 v[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
				v[u] = __builtin_nontemporal_load(src + (size_t)u * gstep + ((size_t)b << bin_e));
#pragma unroll
			for (int u = 0; u < 8; u++)
				a = peak_hold ? (v[u] > a ? v[u] : a) : a + v[u];
		}
	for (; g < g1; g++, src += gstep)
		for (int b = 0; b < nbpt; b++) {
			const i64 v = src[(size_t)b << bin_e];
			a = peak_hold ? (v > a ? v : a) : a + v;
		}
	if (peak_hold) atomicMax((long long *)&avg[out], a);
	else if (a) atomicAdd((unsigned long long *)&avg[out], (unsigned long long)a);
}

// eff_len a multiple of 2^(bin_e+1), bin_e = 14 .. 21; scratch: cap_blocks * 2^bin_e dwords; dc: 2 ints per (pass, tune)
extern "C" int rxk_pw_fft_mid(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes,
                              int bin_e, int eff_len, const int *window, const uint32_t *twiddle, int peak_hold,
                              uint32_t *scratch, size_t cap_blocks, int *dc, long long *avg, long long *partial, size_t partial_cap, int dc_sums_done,
                              int *samples, int samples_add)
{
	hipStream_t s = (hipStream_t)stream;
	const size_t n = (size_t)1 << bin_e;
	const int nbpt = (int)((size_t)eff_len / (2 * n));
	const size_t per_pass = (size_t)tunes * (size_t)nbpt;
	if (bin_e < 14 || bin_e > 21 || !nbpt || (size_t)eff_len % (2 * n) || cap_blocks < per_pass)
		return -1;
	const uint32_t *tw2 = twiddle + (n >> 1);                        // the doubled half of rxgpu_twiddle_table (bfly_pk)
	/* dc_sums_done: the producer of `in` (rxk_pw_fifth_regn4) left remove_dc's sums where pwb_dc would have put them (rxk_pw_dc_sums) -- only the division is left */
	if (dc_sums_done) {
		const size_t npt = (size_t)passes * tunes;
		hipLaunchKernelGGL(k_pwb_dc_fin, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, s, (const i64 *)rxk_pw_dc_sums(dc, npt), (int)npt, eff_len, dc,
		                   samples, tunes, samples_add);
	} else {
		pwb_dc(s, in, tune_stride, pass_stride, passes, tunes, eff_len, dc, samples, samples_add);
	}
	const int max_np = (int)(cap_blocks / per_pass);
	for (int p0 = 0; p0 < passes; p0 += max_np) {
		const int np = passes - p0 < max_np ? passes - p0 : max_np;
		const size_t q0 = (size_t)p0 * per_pass, nq = (size_t)np * per_pass;
		const unsigned g_head = (unsigned)((nq * (n / 16) + 255) / 256);
		/* enough workgroups to fill the chip, few enough that the int64 accumulators amortise the atomics */
		const unsigned wg_x = (unsigned)(per_pass * (n / 4096));
		int groups = (int)((RXK_PWM_TARGET_WG + wg_x - 1) / wg_x);
		if (groups > np) groups = np;
		const int ppg = (np + groups - 1) / groups;
		groups = (np + ppg - 1) / ppg;
		i64 *part = (partial && (size_t)groups * per_pass * n <= partial_cap) ? (i64 *)partial : nullptr;
#define HEAD0(MM) hipLaunchKernelGGL((k_pwm_head<MM>), dim3(g_head), dim3(256), 0, s, in, tune_stride, pass_stride, tunes, nbpt, window, tw2, dc, q0, nq, scratch)
#define HEADN(MM, PP) hipLaunchKernelGGL((k_pwm_head_mid<MM, PP>), dim3(g_head), dim3(256), 0, s, scratch, nq, tw2)
#define TAIL(MM, HH) do { \
		if (peak_hold) hipLaunchKernelGGL((k_pwm_tail<MM, true, HH>), dim3(wg_x, (unsigned)groups), dim3(256), 0, s, scratch, tunes, nbpt, p0, np, ppg, tw2, (i64 *)avg, part); \
		else hipLaunchKernelGGL((k_pwm_tail<MM, false, HH>), dim3(wg_x, (unsigned)groups), dim3(256), 0, s, scratch, tunes, nbpt, p0, np, ppg, tw2, (i64 *)avg, part); } while (0)
		/* N >= 2^17: the first two passes in one launch (k_pwm_head2); input that is not 8-byte aligned (its loads) takes one launch each */
		const bool head2 = (((size_t)in & 7u) | (tune_stride & 3u) | (pass_stride & 3u)) == 0;
#define HEAD2(MM) hipLaunchKernelGGL((k_pwm_head2<MM>), dim3((unsigned)(nq * ((n >> 8) / 32))), dim3(256), 0, s, in, tune_stride, pass_stride, tunes, nbpt, window, tw2, dc, q0, nq, scratch)
#define HEAD01(MM) do { if (head2) HEAD2(MM); else { HEAD0(MM); HEADN(MM, 1); } } while (0)
		switch (bin_e) {
		case 14: HEAD0(14); TAIL(14, 1); break;
		case 15: HEAD0(15); TAIL(15, 1); break;
		case 16: HEAD0(16); TAIL(16, 1); break;
		case 17: HEAD01(17); TAIL(17, 2); break;
		case 18: HEAD01(18); TAIL(18, 2); break;
		case 19: HEAD01(19); TAIL(19, 2); break;
		case 20: HEAD01(20); TAIL(20, 2); break;
		default: HEAD01(21); HEADN(21, 2); TAIL(21, 3); break;
		}
#undef HEAD01
#undef HEAD2
#undef TAIL
#undef HEADN
#undef HEAD0
		if (part) {
			/* the permuted fold ends in SCATTERED atomics (bit-reversed bins): one slice of the groups per 2^18 of them -- sixteen slices of a
			 * 2^18-point spectrum were 4 M scattered atomics per launch and cost more than the linear stores had saved */
			const unsigned slices = bin_e >= 18 ? 1u : (16u >> (bin_e > 14 ? bin_e - 14 : 0));
			hipLaunchKernelGGL(k_pwm_reduce<true>, dim3((unsigned)((((size_t)tunes << bin_e) + 255) / 256), slices), dim3(256), 0, s, part, tunes, nbpt, groups,
			                   bin_e, peak_hold, (i64 *)avg);
		}
	}
	LAUNCH_RET();
}

// scratch: cap_blocks * 2^bin_e dwords; dc: 2 ints per (pass, tune)
extern "C" int rxk_pw_fft_big(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes,
                              int bin_e, int eff_len, const int *window, const uint32_t *twiddle, int peak_hold,
                              uint32_t *scratch, size_t cap_blocks, int *dc, long long *avg)
{
	hipStream_t s = (hipStream_t)stream;
	const size_t n = (size_t)1 << bin_e;
	const int nbpt = (int)(((size_t)eff_len + 2 * n - 1) / (2 * n));
	const size_t total = (size_t)passes * (size_t)tunes * (size_t)nbpt;
	pwb_dc(s, in, tune_stride, pass_stride, passes, tunes, eff_len, dc);
	for (size_t q0 = 0; q0 < total; q0 += cap_blocks) {
		const size_t nq = total - q0 < cap_blocks ? total - q0 : cap_blocks;
		const unsigned g_full = (unsigned)((nq * n + 255) / 256), g_half = (unsigned)((nq * (n / 2) + 255) / 256);
		hipLaunchKernelGGL(k_pwb_load, dim3(g_full), dim3(256), 0, s, in, tune_stride, pass_stride, tunes, nbpt, bin_e, eff_len, window, dc,
		                   q0, nq, scratch);
		for (int st = 0; st < bin_e; st++)
			hipLaunchKernelGGL(k_pwb_stage, dim3(g_half), dim3(256), 0, s, scratch, nq, bin_e, st, twiddle);
		hipLaunchKernelGGL(k_pwb_acc, dim3(g_full), dim3(256), 0, s, scratch, tunes, nbpt, bin_e, q0, nq, peak_hold, (i64 *)avg);
	}
	LAUNCH_RET();
}

extern "C" int rxk_pw_fft(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes,
                          int bin_e, int eff_len, int dc_len, const int *window, const uint32_t *twiddle, int peak_hold,
                          int passes_per_group, long long *avg, long long *partial, size_t partial_cap)
{
	(void)dc_len;
	const int n = 1 << bin_e;
	const size_t shm = (size_t)n * 4 + 16 * 8;
	const int groups = (passes + passes_per_group - 1) / passes_per_group;
	dim3 grid((unsigned)tunes, (unsigned)groups);
	hipStream_t s = (hipStream_t)stream;
	const int fpw = n >= 4096 ? 1 : 4096 / n;                       /* transforms side by side in a k_pw_fftR workgroup */
	const bool k4096 = bin_e == 12 && (eff_len == 8192 || eff_len == 16384 || eff_len == 32768);
	i64 *part = (partial && bin_e >= 5 && bin_e <= 13 && eff_len % (2 * n) == 0 &&
	             (size_t)groups * tunes * fpw * (size_t)n <= partial_cap) ? (i64 *)partial : nullptr;
	if (k4096) {
		const int nb = eff_len / 8192;
#define GO4K_(NB, TWL) do { if (peak_hold) hipLaunchKernelGGL((k_pw_fft4096<NB, true, TWL>), grid, dim3(256), 0, s, in, tune_stride, pass_stride, passes, window, twiddle + 2048, passes_per_group, (i64 *)avg, part); \
		else hipLaunchKernelGGL((k_pw_fft4096<NB, false, TWL>), grid, dim3(256), 0, s, in, tune_stride, pass_stride, passes, window, twiddle + 2048, passes_per_group, (i64 *)avg, part); } while (0)
#define GO4K(NB) GO4K_(NB, true)
		if (nb == 1) GO4K(1); else if (nb == 2) GO4K(2); else GO4K(4);
#undef GO4K
		if (part)
			hipLaunchKernelGGL(k_pwm_reduce<false>, dim3((unsigned)((((size_t)tunes << bin_e) + 255) / 256), 16), dim3(256), 0, s, part, tunes, 1, groups, bin_e,
			                   peak_hold, (i64 *)avg);
		LAUNCH_RET();
	}
	if (bin_e >= 8 && bin_e <= 13 && eff_len % (2 * n) == 0) {
		/* register-blocked kernel for every power of two from 256 to 8192.  2^14 was tried (round 2): N/16 = 1024 threads leave 128
		 * VGPRs per lane, the transform wants ~200, and the spilling build ran 3.3x slower than the LDS radix-2 kernel below */
		const int nb_total = eff_len / (2 * n);
		const int T = (n / 16 > 256) ? n / 16 : 256;
		const size_t lds_bytes = (size_t)(bin_e <= 12 ? 2 : 1) * T * RXK_FFT_XROW * 4 + 32 * 8 + (size_t)8 * ((n >> 4) + 8) * 4;   /* transposes (XROW), red, twiddle copy */
#define GOR(MM) do { \
		if (lds_bytes > 64 * 1024) { \
			(void)hipFuncSetAttribute((const void *)k_pw_fftR<MM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			(void)hipFuncSetAttribute((const void *)k_pw_fftR<MM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); } \
		if (peak_hold) hipLaunchKernelGGL((k_pw_fftR<MM, true>), grid, dim3(T), lds_bytes, s, in, tune_stride, pass_stride, passes, nb_total, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); \
		else hipLaunchKernelGGL((k_pw_fftR<MM, false>), grid, dim3(T), lds_bytes, s, in, tune_stride, pass_stride, passes, nb_total, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); } while (0)
		/* N = 256 ... 2048 with one, two or four groups of side-by-side transforms per buffer: the buffer in registers, read once (k_pw_fftR2) */
		const int ng = (nb_total % fpw == 0) ? nb_total / fpw : 0;
#define GOR2_(MM, NGG) do { \
		if (lds_bytes > 64 * 1024) { \
			(void)hipFuncSetAttribute((const void *)k_pw_fftR2<MM, NGG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			(void)hipFuncSetAttribute((const void *)k_pw_fftR2<MM, NGG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); } \
		if (peak_hold) hipLaunchKernelGGL((k_pw_fftR2<MM, NGG, true>), grid, dim3(T), lds_bytes, s, in, tune_stride, pass_stride, passes, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); \
		else hipLaunchKernelGGL((k_pw_fftR2<MM, NGG, false>), grid, dim3(T), lds_bytes, s, in, tune_stride, pass_stride, passes, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); } while (0)
#define GOR2(MM) do { if (ng == 1) GOR2_(MM, 1); else if (ng == 2) GOR2_(MM, 2); else GOR2_(MM, 4); } while (0)
		if ((ng == 1 || ng == 2 || ng == 4) && bin_e != 12) {                /* (N = 4096 with these buffers is k_pw_fft4096's, above) */
			switch (bin_e) { case 8: GOR2(8); break; case 9: GOR2(9); break; case 10: GOR2(10); break; case 11: GOR2(11); break; default: GOR2(13); break; }
		} else
		switch (bin_e) {
		case 8: GOR(8); break; case 9: GOR(9); break; case 10: GOR(10); break; case 11: GOR(11); break;
		case 12: GOR(12); break; default: GOR(13); break;
		}
#undef GOR2
#undef GOR2_
#undef GOR
		if (part)
			hipLaunchKernelGGL(k_pwm_reduce<false>, dim3((unsigned)((((size_t)tunes << bin_e) + 255) / 256), 16), dim3(256), 0, s, part, tunes, fpw, groups, bin_e,
			                   peak_hold, (i64 *)avg);
		LAUNCH_RET();
	}
	if (bin_e >= 5 && bin_e <= 7 && eff_len % (2 * n) == 0) {
		/* N = 32 ... 128 (coarse bins: the classic -f 88M:108M:125k): the same register-blocked kernel, 128 ... 32 transforms side by side in a workgroup */
		const int nb_total = eff_len / (2 * n), ng = (nb_total % fpw == 0) ? nb_total / fpw : 0;
		const size_t lds_bytes = (size_t)2 * 256 * RXK_FFT_XROW * 4 + 32 * 8 + (size_t)8 * ((n >> 4) + 8) * 4;
#define GOS_(MM, NGG) do { \
		if (peak_hold) hipLaunchKernelGGL((k_pw_fftR2<MM, NGG, true>), grid, dim3(256), lds_bytes, s, in, tune_stride, pass_stride, passes, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); \
		else hipLaunchKernelGGL((k_pw_fftR2<MM, NGG, false>), grid, dim3(256), lds_bytes, s, in, tune_stride, pass_stride, passes, window, twiddle + (1 << (MM - 1)), passes_per_group, (i64 *)avg, part); } while (0)
#define GOS(MM) do { if (ng == 1) GOS_(MM, 1); else if (ng == 2) GOS_(MM, 2); else GOS_(MM, 4); } while (0)
		if (ng == 1 || ng == 2 || ng == 4) {
			if (bin_e == 5) GOS(5); else if (bin_e == 6) GOS(6); else GOS(7);
			if (part)
				hipLaunchKernelGGL(k_pwm_reduce<false>, dim3((unsigned)((((size_t)tunes << bin_e) + 255) / 256), 16), dim3(256), 0, s, part, tunes, fpw, groups, bin_e,
				                   peak_hold, (i64 *)avg);
			LAUNCH_RET();
		}
#undef GOS
#undef GOS_
	}
#define GO(A) do { \
		if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_pw_fft<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
		hipLaunchKernelGGL((k_pw_fft<A>), grid, dim3(256), shm, s, in, tune_stride, pass_stride, passes, bin_e, eff_len, \
		                   window, twiddle, peak_hold, passes_per_group, (i64 *)avg); } while (0)
	if (n <= 256) GO(1);
	else if (n <= 512) GO(2);
	else if (n <= 1024) GO(4);
	else if (n <= 2048) GO(8);
	else if (n <= 4096) GO(16);
	else GO(0);
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_pw_gather_rows(void *stream, const void *const *d_rows, int rows, size_t row_bytes, int16_t *out)
{
	const unsigned units = (unsigned)(row_bytes / 16);
	hipLaunchKernelGGL(k_pw_gather_rows, dim3((units + 1023) / 1024, (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
	                   (const uint4 *const *)d_rows, (uint4 *)out, units);
	LAUNCH_RET();
}

extern "C" int rxk_pw_merge_rows(void *stream, void *const *d_rows, int rows, size_t row_bytes, long long *acc, int peak_hold, int host_rows_zero)
{
	const unsigned units = (unsigned)(row_bytes / 16);
	if (host_rows_zero)
		hipLaunchKernelGGL(k_pw_merge_rows<true>, dim3((units + 1023) / 1024, (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
		                   (uint4 *const *)d_rows, (uint4 *)acc, units, peak_hold);
	else
		hipLaunchKernelGGL(k_pw_merge_rows<false>, dim3((units + 1023) / 1024, (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
		                   (uint4 *const *)d_rows, (uint4 *)acc, units, peak_hold);
	LAUNCH_RET();
}

extern "C" int rxk_pw_samples(void *stream, int *samples, int tunes, int add)
{
	hipLaunchKernelGGL(k_pw_samples, dim3((tunes + 255) / 256), dim3(256), 0, (hipStream_t)stream, samples, tunes, add);
	LAUNCH_RET();
}

extern "C" int rxk_pw_boxcar(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int buf_len, int ds, int n_write)
{
	const int nc = buf_len / 2;
	if (n_write > nc || n_write <= 0)
		n_write = nc;
	const u64 total = (u64)n_bufs * n_write;
	hipLaunchKernelGGL(k_pw_boxcar, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t *)in, (uint32_t *)out, n_bufs, nc, ds, n_write);
	LAUNCH_RET();
}

extern "C" int rxk_pw_fifth(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int n_in, int in_stride, int out_stride)
{
	const u64 total = (u64)n_bufs * ((n_in + 1) / 2);
	hipLaunchKernelGGL(k_pw_fifth, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t *)in, (uint32_t *)out, n_bufs, n_in, in_stride, out_stride);
	LAUNCH_RET();
}

extern "C" int rxk_pw_droop(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int n, int stride, const int *fir)
{
	if (n >= 16 && (n & 3) == 0 && (stride & 3) == 0 && (((size_t)in | (size_t)out) & 15u) == 0) {
		/* four outputs per thread: twelve samples in registers instead of 4 x 9 loads, one 16-byte store, the buffer index from the grid */
		hipLaunchKernelGGL(k_pw_droop4, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)n_bufs), dim3(256), 0, (hipStream_t)stream,
		                   (const uint32_t *)in, (uint32_t *)out, n, stride, fir);
		LAUNCH_RET();
	}
	const u64 total = (u64)n_bufs * n;
	hipLaunchKernelGGL(k_pw_droop, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t *)in, (uint32_t *)out, n_bufs, n, stride, fir);
	LAUNCH_RET();
}

extern "C" int rxk_pw_rms_sums(void *stream, const int16_t *in, size_t n_bufs, int buf_len, long long *t, long long *p)
{
	hipLaunchKernelGGL(k_pw_rms_sums, dim3((unsigned)n_bufs), dim3(256), 0, (hipStream_t)stream, in, n_bufs, buf_len, (i64 *)t, (i64 *)p);
	LAUNCH_RET();
}

extern "C" int rxk_pw_rms_apply(void *stream, const long long *t, const long long *p, int passes, int tunes, int buf_len,
                                int peak_hold, long long *avg, int *samples)
{
	hipLaunchKernelGGL(k_pw_rms_apply, dim3((tunes + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (const i64 *)t, (const i64 *)p, passes, tunes, buf_len, peak_hold, (i64 *)avg, samples);
	LAUNCH_RET();
}
