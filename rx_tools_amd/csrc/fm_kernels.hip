// fm_kernels.hip -- gfx950 kernels for the rx_fm stream pre-stage + full_demod() chain.
//
// Reference behaviour (file:line under /root/reference/src):
//   F0 scale        rtl_fm.c:845-848     F1 rotate16_90   rtl_fm.c:309-327
//   F2 low_pass     rtl_fm.c:351-371     F3 fifth_order   rtl_fm.c:411-440, 764-769
//   F5 fm_demod     rtl_fm.c:584-615     F6 polar_disc_fast/fast_atan2 rtl_fm.c:485-513
//   F8 deemph       rtl_fm.c:667-682     F9 low_pass_real rtl_fm.c:389-409
//   F12 generic_fir rtl_fm.c:442-465
//
// All arithmetic is integer and bit-exact with the C reference except where the reference
// itself goes through libm (polar_discriminant, rtl_fm.c:476-483): there the device
// evaluates in fp64 and flags results too close to a truncation boundary for the host to
// re-evaluate with the same libm the reference uses.
//
// Data layout: the IQ stream is int16 I,Q interleaved (cs16) exactly as SoapySDR delivers
// it; one complex sample = one dword.  A lane reads 4 consecutive samples with one
// global_load_dwordx4 (1 KiB per wave instruction, fully coalesced).  Decimated IQ is kept
// packed (I | Q<<16) mod 2^16, which is exactly the int16 truncation the reference applies
// when it stores int sums back into lowpassed[].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "fft_device.h"

typedef unsigned long long u64;
typedef long long i64;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DEC_THREADS 256
#define DEC_WAVE_SPAN (RXK_DEC_SPAN / 4)     // samples per wave
#define DEC_TILE 256                         // samples per wave load (64 lanes x 4)
#define DEC_TILES (DEC_WAVE_SPAN / DEC_TILE)
#define DEC_BATCH 8

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
	s16x2 r = __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b);   // v_pk_add_u16
	return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
	s16x2 r = __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b);   // v_pk_sub_u16
	return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_iq(int i, int q) { return ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16); }
__device__ __forceinline__ int lo16(uint32_t w) { return (int)(short)(w & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t w) { return (int)w >> 16; }

// Tiled layout of the demodulated stream for the lane-per-chunk audio kernels (k_fm_deemph_scan_t / _apply_rs_t):
// a tile is 64 chunks of CH = 2^chl2 samples; inside a tile the 16-byte unit u (8 samples) of chunk c sits at unit
// u*64 + c, so that 64 lanes walking 64 consecutive chunks read 1 KiB contiguous per instruction -- no LDS transpose,
// no staging, full occupancy.  Sample m = tile*64*CH + c*CH + k  ->  tile*64*CH + (k/8)*512 + c*8 + k%8.  chl2 == 0: linear.
__device__ __forceinline__ u64 pcm_index(u64 m, int chl2)
{
	if (!chl2)
		return m;
	const unsigned k = (unsigned)m & ((1u << chl2) - 1u), c = (unsigned)(m >> chl2) & 63u;
	return (m & ~(((u64)64 << chl2) - 1)) | ((u64)(k >> 3) << 9) | (c << 3) | (k & 7u);
}

// F0, rtl_fm.c:846: (int16)(x / 32767.0 * 128.0 + 0.4) in double, truncated.  One fp32
// fma reproduces it for all 65536 inputs (checked exhaustively in tests/test_scale.py and
// on the device in tests/test_gpu_fm.py): the result is never closer than 6.0e-6 to an
// integer while the single rounding error of fma at magnitude <= 128.4 is <= 3.9e-6 and
// the coefficient error contributes <= 3.8e-6 of the same sign budget.
__device__ __forceinline__ int scale_cs16(int x)
{
	return (int)__builtin_fmaf((float)x, (float)(128.0 / 32767.0), 0.4f);
}

// wave64 inclusive scan of packed int16 pairs with DPP (row_shr 1,2,4,8 then row_bcast 15/31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_step(uint32_t v)
{
	uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
	return pk_add(v, t);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
	v = dpp_step<0x111, 0xf>(v);
	v = dpp_step<0x112, 0xf>(v);
	v = dpp_step<0x114, 0xf>(v);
	v = dpp_step<0x118, 0xf>(v);
	v = dpp_step<0x142, 0xa>(v);
	v = dpp_step<0x143, 0xc>(v);
	return v;
}

// The same scan as ONE 32-bit add per step (v_add_u32_dpp) for values whose low halves are small and non-negative:
// with the callback's scaling (values in [-127, 128]) a lane's I sum lies in [-510, 510] when rotated and in
// [-508, 512] when not; biased by 511 every low half is positive and the 64 of them add up to at most 65 472, so no
// carry ever reaches the Q half and the packed sum is exact.  The caller removes 511*(lane+1).
__device__ __forceinline__ uint32_t wave_scan_incl_biased(uint32_t v)
{
	uint32_t t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v += t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v += t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v += t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v += t;
	// rows keep their value where the mask excludes them: dst = dst + dpp(dst) in place
	asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
	return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two cs16 components through F0 (scale_cs16) with the signs of rotate16_90 folded in -- trunc(-r) == -trunc(r) --
// packed as int16 pair (lo from a, hi from b): v_cvt_f32_i32 (SDWA) x2, one v_pk_fma_f32, v_cvt_i32_f32 x2 (the second into the high half)
// the two constant pairs of the scale live in VGPRs for the whole kernel (made opaque once by scale_consts: left as literals the
// compiler rebuilds both pairs -- two v_mov_b64 and four s_mov -- in front of every tile)
struct scale_k { f32x2 cc, hh; };
__device__ __forceinline__ scale_k scale_consts()
{
	scale_k k;
	k.cc = (f32x2){(float)(128.0 / 32767.0), (float)(128.0 / 32767.0)};
	k.hh = (f32x2){0.4f, 0.4f};
	asm volatile("" : "+v"(k.cc), "+v"(k.hh));
	return k;
}

template <int SA, int SB>
__device__ __forceinline__ uint32_t scale_pk(int a, int b, const scale_k &K)
{
	const f32x2 x = {(float)a, (float)b};
	const f32x2 cc = K.cc, hh = K.hh;
	f32x2 r;
	// the sign pairs are source modifiers of the one constant pair (the compiler would materialise four)
	if (SA > 0 && SB > 0)
		asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(cc), "v"(hh));
	else if (SA < 0 && SB > 0)
		asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,1]" : "=v"(r) : "v"(x), "v"(cc), "v"(hh));
	else if (SA > 0 && SB < 0)
		asm("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(cc), "v"(hh));
	else
		asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,1] neg_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(cc), "v"(hh));
	// |value| <= 128: the second conversion writes its low half straight into the high half of the pair (SDWA destination select),
	// no v_cvt_pk_i16_i32
	uint32_t out = (uint32_t)(int)r.x;
	asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(out) : "v"(r.y));
	return out;
}

// a lane's four samples as packed (I,Q) contributions to the running sums: rotate16_90 (rtl_fm.c:309-327) multiplies
// sample n of the block by j^n, and a lane's samples sit at phases 0..3
template <bool PRESCALED, bool ROTATE>
__device__ __forceinline__ void dec_contrib(const u32x4 v, uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3, const scale_k &K)
{
	if (!PRESCALED) {
		s0 = scale_pk<1, 1>(lo16(v.x), hi16(v.x), K);
		if (ROTATE) {
			s1 = scale_pk<-1, 1>(hi16(v.y), lo16(v.y), K);         // (-q1,  i1)
			s2 = scale_pk<-1, -1>(lo16(v.z), hi16(v.z), K);        // (-i2, -q2)
			s3 = scale_pk<1, -1>(hi16(v.w), lo16(v.w), K);         // ( q3, -i3)
		} else {
			s1 = scale_pk<1, 1>(lo16(v.y), hi16(v.y), K);
			s2 = scale_pk<1, 1>(lo16(v.z), hi16(v.z), K);
			s3 = scale_pk<1, 1>(lo16(v.w), hi16(v.w), K);
		}
	} else {
		s0 = v.x;
		if (ROTATE) {
			s1 = pack_iq(-hi16(v.y), lo16(v.y));
			s2 = pk_sub(0u, v.z);
			s3 = pack_iq(hi16(v.w), -lo16(v.w));
		} else {
			s1 = v.y; s2 = v.z; s3 = v.w;
		}
	}
}

// ------------------------------------------------------------------ F0+F1+F2 fused

// One workgroup = RXK_DEC_SPAN consecutive complex samples of the stream, 4 waves of
// DEC_WAVE_SPAN each.  Every lane turns its 4 samples into rotated, scaled partial sums,
// a wave-level DPP scan gives the running (I,Q) prefix, and the lane that holds the last
// sample of a boxcar window drops the prefix at that point into an LDS slot.  After a
// barrier, output j = slot[j] - slot[j-1].  The window that straddles the workgroup start
// is finished by rxk_fm_disc from head[]/tail[].
template <bool DEN24 = false>
__device__ __forceinline__ int fast_atan2_dev(int y, int x);
__device__ __forceinline__ void mul_conj_pk(uint32_t a, uint32_t b, int &cr, int &cj);

// the packed prefix up to the end of window j of a span: slot[j] plus the totals of the waves before the one that wrote it
// (b1, b2, b3 = totals of waves 0, 0..1, 0..2); the window's last sample, span-relative, names the writer.  Range tests on
// the position, not a switch on the wave number: the compiler turns the latter into a table in scratch memory.
__device__ __forceinline__ uint32_t dec_prefix(const uint32_t *slot, unsigned j, unsigned ds, unsigned e_off, uint32_t b1, uint32_t b2, uint32_t b3)
{
	const unsigned e1 = __umul24(j, ds) + e_off;             // j < 2^13, ds < 2^14
	uint32_t base = e1 >= DEC_WAVE_SPAN ? b1 : 0u;
	base = e1 >= 2 * DEC_WAVE_SPAN ? b2 : base;
	base = e1 >= 3 * DEC_WAVE_SPAN ? b3 : base;
	return pk_add(slot[j], base);
}

// WIDE slots: the main loop leaves the SELECTION of the prefix inside the lane (which of its four samples ends the window) to the
// reader -- once per output instead of once per lane and tile.  A record holds X = the biased prefix through the lane's four
// samples and the three suffix sums behind samples 0, 1, 2 of the lane; the window's last sample e1 names wave, tile, lane and
// sample, and with them what to take off: the suffix behind that sample and the scan's bias (511 per lane, 64 * 511 per tile).
__device__ __forceinline__ uint32_t dec_prefix_wide(const uint4 *slot4, unsigned j, unsigned ds, unsigned e_off, uint32_t b1, uint32_t b2, uint32_t b3)
{
	const unsigned e1 = __umul24(j, ds) + e_off;
	uint32_t base = e1 >= DEC_WAVE_SPAN ? b1 : 0u;
	base = e1 >= 2 * DEC_WAVE_SPAN ? b2 : base;
	base = e1 >= 3 * DEC_WAVE_SPAN ? b3 : base;
	const uint4 rec = slot4[j];
	const unsigned pos = e1 & (DEC_WAVE_SPAN - 1), r = pos & 3u;
	const unsigned unb = (511u * (((pos >> 2) & 63u) + 1u) + 32704u * (pos >> 8)) & 0xffffu;
	uint32_t suf = r == 0 ? rec.y : r == 1 ? rec.z : rec.w;
	suf = r == 3 ? 0u : suf;
	return pk_add(pk_sub(pk_sub(rec.x, unb), suf), base);
}

// DISC: also run the -A fast discriminator (F5/F6) for every output whose predecessor was completed
// by this workgroup too (all but its first two), while the sums are still in LDS/registers; the
// two seam outputs per workgroup and each block's libm sample are left to k_fm_disc.
// DCS (rx_power's boxcar in front of a large transform, rtl_power.c:723-733 then 609-624): remove_dc's sums of the decimated buffers ride along --
// every wave leaves the sums of the outputs it stored (int I, int Q) at dc_sums[(span * 4 + wave)] (here: an int2 array, one entry per wave of every
// span); the seam kernel adds a span's four, its own output, and does the buffer's atomics.  The transform then only divides (no pass of its own
// over the block).
template <bool PRESCALED, bool ROTATE, bool DISC, bool DIV24, bool WIDE = false, bool DCS = false>
__global__ __launch_bounds__(DEC_THREADS) void k_fm_decimate(
	const u32x4 *__restrict__ iq, u64 T, int ds, int p0, unsigned magic, unsigned magic24,
	uint32_t *__restrict__ lp_raw, uint32_t *__restrict__ head, uint32_t *__restrict__ tail, unsigned slot_cap,
	int lp_sparse, int16_t *__restrict__ pcm, int pcm_chl2, i64 *__restrict__ dc_sums = nullptr, unsigned outs_per_buf = 1)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	static_assert(!(WIDE && PRESCALED), "wide slots ride on the biased scan of the raw path");
	uint32_t *slot = lds;
	uint4 *slot4 = reinterpret_cast<uint4 *>(lds);              // WIDE: 16-byte records (dec_prefix_wide)
	uint32_t *wtot = lds + (WIDE ? 4 : 1) * slot_cap;

	// workgroup b runs on XCD b % 8 (observed placement): give every XCD one contiguous eighth of the stream so that
	// the partial output lines of neighbouring spans meet in the same L2 instead of being written back one by one
	const unsigned per = gridDim.x >> 3;
	const unsigned wgi = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	const u64 wg0 = (u64)wgi * RXK_DEC_SPAN;
	const u64 left = T - wg0;
	const unsigned span = left < (u64)RXK_DEC_SPAN ? (unsigned)left : (unsigned)RXK_DEC_SPAN;
	const u64 t0 = wg0 + (unsigned)p0;
	const u64 m_base = t0 / (unsigned)ds;                  // outputs completed before this span
	const unsigned ph = (unsigned)(t0 - m_base * (unsigned)ds);
	const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const u32x4 *src = iq + (wg0 >> 2);

	uint32_t run = 0;                                       // wave-uniform running prefix
	const uint32_t unbias = 511u * (lane + 1);              // what the biased scan added to this lane's I prefix (< 2^16)
	const scale_k K = scale_consts();
	const unsigned dummy = slot_cap - 1;                    // where lanes without a window end put their (unused) prefix
#pragma unroll
	for (int half = 0; half < DEC_TILES / DEC_BATCH; half++) {      // DEC_BATCH loads in flight per lane
		u32x4 v[DEC_BATCH];
#pragma unroll
		for (int u = 0; u < DEC_BATCH; u++) {
			unsigned rel = wave * DEC_WAVE_SPAN + (half * DEC_BATCH + u) * DEC_TILE + lane * 4;
			v[u] = rel < span ? __builtin_nontemporal_load(src + (rel >> 2)) : (u32x4)(0u);
		}
#pragma unroll
		for (int u = 0; u < DEC_BATCH; u++) {
			const unsigned rel = wave * DEC_WAVE_SPAN + (half * DEC_BATCH + u) * DEC_TILE + lane * 4;
			uint32_t s0, s1, s2, s3;
			dec_contrib<PRESCALED, ROTATE>(v[u], s0, s1, s2, s3, K);
			const uint32_t s23 = pk_add(s2, s3), s123 = pk_add(s1, s23);           // WIDE: the suffix sums the reader selects from
			const uint32_t c1 = s0, c2 = pk_add(c1, s1), c3 = pk_add(c2, s2), c4 = WIDE ? pk_add(s0, s123) : pk_add(c3, s3);
			uint32_t incl;
			if (WIDE)
				incl = wave_scan_incl_biased(pk_add(c4, 511u));                    // stays biased: the reader takes the bias off
			else if (!PRESCALED)
				incl = pk_sub(wave_scan_incl_biased(pk_add(c4, 511u)), unbias);
			else
				incl = wave_scan_incl(c4);
			// does a window end inside this lane?  windows end (exclusive) at (j+1)*ds - ph:
			// k = floor(qn / ds) windows end at or before this lane's last sample, the k-th after cnt of its samples
			const unsigned qn = rel + 4 + ph;
			unsigned k;
			int cnt;
			if (DIV24) {                                    // qn * ds < 2^24: two full-rate 24-bit multiplies
				k = (unsigned)(((unsigned long long)((qn << 8) & 0xffffffu) * (unsigned long long)(magic24 & 0xffffffu)) >> 32);
				asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cnt) : "v"(k), "s"(ds), "v"(4u - qn));
			} else {
				k = __umulhi(qn, magic);
				cnt = (int)(k * (unsigned)ds + 4u - qn);
			}
			// cnt = 1..4 samples of this lane belong to window k-1: the prefix up to there goes to its slot.  No branch: every lane
			// stores (the others into one spare slot), so that the tiles of a batch form one basic block
			const bool hit = cnt > 0 && rel < span;
			if (WIDE) {
				slot4[hit ? k - 1 : dummy] = make_uint4(pk_add(run, incl), s123, s23, s3);
			} else {
				const uint32_t sel = cnt <= 1 ? c1 : cnt == 2 ? c2 : cnt == 3 ? c3 : c4;
				slot[hit ? k - 1 : dummy] = pk_add(pk_add(run, pk_sub(incl, c4)), sel);
			}
			run = pk_add(run, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
		}
	}
	if (lane == 0)
		wtot[wave] = WIDE ? pk_sub(run, (uint32_t)((DEC_TILES * 32704u) & 0xffffu)) : run;    // WIDE: lane 63's bias of every tile
	__syncthreads();

	const uint32_t b1 = wtot[0], b2 = pk_add(b1, wtot[1]), b3 = pk_add(b2, wtot[2]);
	const uint32_t total = pk_add(b3, wtot[3]);
	const unsigned n_b = (span + ph) / (unsigned)ds;
	// P(j): the packed prefix up to the end of window j = slot[j] + the totals of the waves before the one that wrote it
	// (the window's last sample, span-relative e1 = j*ds + ds - ph - 1, names that wave)
	const unsigned e_off = (unsigned)ds - ph - 1u;
#define PREFIX_AT(J) (WIDE ? dec_prefix_wide(slot4, (J), (unsigned)ds, e_off, b1, b2, b3) : dec_prefix(slot, (J), (unsigned)ds, e_off, b1, b2, b3))
	// Every wave takes a contiguous quarter of the outputs, 64 consecutive ones per turn: output j = P(j) - P(j-1) and the
	// discriminator also wants P(j-2) -- both sit in the neighbouring lanes (two wave-wide DPP shifts), the two values
	// that cross a turn travel in SGPRs.  One LDS read and one prefix selection per output instead of three.
	const unsigned per_wave = (n_b + 3) / 4;
	const unsigned j_lo = wave * per_wave, j_hi = min(n_b, j_lo + per_wave);
	uint32_t c1 = 0, c2 = 0;                                 // P(j-1), P(j-2) for the turn's first lane
	int dsi = 0, dsq = 0;                                    // DCS: this lane's share of the buffer's sums (a few hundred int16 at most)
	if (j_lo < j_hi) {
		if (j_lo >= 1) c1 = PREFIX_AT(j_lo - 1);
		if (j_lo >= 2) c2 = PREFIX_AT(j_lo - 2);
	}
	for (unsigned j0 = j_lo; j0 < j_hi; j0 += 64) {
		const unsigned j = j0 + lane;
		const uint32_t pj = PREFIX_AT(j);                                    // lanes past j_hi read slots nobody wrote: never stored
		const uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp((int)c1, (int)pj, 0x138, 0xf, 0xf, false);    // wave_shr:1, lane 0 keeps c1
		const uint32_t pmm = (uint32_t)__builtin_amdgcn_update_dpp((int)c2, (int)pm, 0x138, 0xf, 0xf, false);
		c1 = (uint32_t)__builtin_amdgcn_readlane((int)pj, 63);
		c2 = (uint32_t)__builtin_amdgcn_readlane((int)pj, 62);
		if (j >= j_hi)
			continue;
		const uint32_t a = pk_sub(pj, pm);
		if (!lp_sparse && j)
			lp_raw[m_base + j] = a;
		if (DCS && j) {
			dsi += lo16(a);
			dsq += hi16(a);
		}
		if (DISC && j >= 2) {
			int cr, cj;
			mul_conj_pk(a, pk_sub(pm, pmm), cr, cj);
			const int16_t v = (int16_t)fast_atan2_dev(cj, cr);
			if (pcm_chl2)
				pcm[pcm_index(m_base + j, pcm_chl2)] = v;        // 16-byte pieces of a line arrive from different turns: let L2 merge them
			else
				__builtin_nontemporal_store(v, &pcm[m_base + j]);
		}
	}
	if (DCS) {
		// (a workgroup lives a few microseconds: no LDS shuffles, no 64-bit division in its tail -- six DPP adds per sum, the buffer from the span number)
#define DPP_ADD(V, CTRL, ROWS) V += __builtin_amdgcn_update_dpp(0, V, CTRL, ROWS, 0xf, true)
		DPP_ADD(dsi, 0x111, 0xf); DPP_ADD(dsq, 0x111, 0xf); DPP_ADD(dsi, 0x112, 0xf); DPP_ADD(dsq, 0x112, 0xf);
		DPP_ADD(dsi, 0x114, 0xf); DPP_ADD(dsq, 0x114, 0xf); DPP_ADD(dsi, 0x118, 0xf); DPP_ADD(dsq, 0x118, 0xf);
		DPP_ADD(dsi, 0x142, 0xa); DPP_ADD(dsq, 0x142, 0xa); DPP_ADD(dsi, 0x143, 0xc); DPP_ADD(dsq, 0x143, 0xc);
#undef DPP_ADD
		// a plain store per wave (k_pw_boxcar_seams adds the four and does the atomics: int64 atomics HERE kept every wave resident for their round
		// trip -- 743 against 620 us for the launch's decimator, more than the dc pass they replaced)
		if (lane == 63)
			reinterpret_cast<int2 *>(dc_sums)[(size_t)wgi * 4 + wave] = make_int2(dsi, dsq);
	}
	// the first window ending in the span goes out in head/tail form; with a sparse lowpassed[] only the entries the seam kernel
	// reads are kept: the span's second output and its last one
	if (threadIdx.x == 0 && n_b)
		head[wgi] = PREFIX_AT(0);
	if (lp_sparse && threadIdx.x >= 64 && threadIdx.x < 66) {
		const unsigned j = threadIdx.x == 64 ? 1u : n_b - 1;
		if (j >= 1 && j < n_b)
			lp_raw[m_base + j] = pk_sub(PREFIX_AT(j), PREFIX_AT(j - 1));
	}
	if (threadIdx.x == 0) {
		uint32_t last = 0;
		if (n_b) {
			last = PREFIX_AT(n_b - 1);
		} else {
			head[wgi] = 0;
		}
		tail[wgi] = pk_sub(total, last);
	}
}

#undef PREFIX_AT

// ------------------------------------------------------------------ F0+F1+F2(+F5/F6) for small decimation, direct form

// -M wbfm's own decimation is 6 (rtl_fm.c:968 with rate_in 170 kHz), BASELINE configs[0] has 5.  There the prefix-scan
// decimator above spends as many instructions on slots, scans and prefix differences as on the samples themselves, and the
// whole chain is bound by VALU issue (rocprofv3: 0.256 wave-instructions per SIMD-cycle at ds = 6).  For ds <= 32 a window is
// shorter than a wave's row of samples, so: stage a span of SCALED, ROTATED samples in LDS once (same coalesced 16-byte loads,
// same packed fp32-fma scale), then one thread per output sums its ds consecutive LDS words -- no scan, no slots, no seams:
// a span also stages the ds samples in front of it and the ds - 1 behind it (0.2 % re-read at ds = 6), so every window that
// STARTS in a span is complete there, and the predecessor the discriminator needs is the neighbouring lane's sum (lane 0 of a
// wave sums it again).  What this kernel cannot know stays with k_fm_disc(sparse, seams = 2): the run's first two outputs (the
// carried now_r/now_j and pre_r/pre_j live in a struct that a later kernel of the previous run is still writing), every
// callback block's first output (libm) and the carries out.
#define DSM_SPAN_MAX 4608                                       // samples of a span at most (see dsm_span)
#define DSM_HALO 32                                            // >= ds: whole 16-byte vectors on either side

// sum of the ds staged samples from LDS word r: the reads are issued together, in groups of eight with a uniform bound (a loop with a
// run-time trip count would wait for each LDS read in turn)
__device__ __forceinline__ uint32_t dsm_window(const uint32_t *sm, int r, int ds)
{
	uint32_t a = 0;
	for (int i0 = 0; i0 < ds; i0 += 8) {
		uint32_t v[8];
#pragma unroll
		for (int i = 0; i < 8; i++)
			v[i] = sm[r + i0 + i];                            // past the window: staged neighbours (or the pad), dropped below
#pragma unroll
		for (int i = 0; i < 8; i++)
			a = pk_add(a, i0 + i < ds ? v[i] : 0u);
	}
	return a;
}

// span of a workgroup in samples: a multiple of 256 * ds (64 * ds for ds > 18) not above 4608, so that every span holds the same
// whole number of windows -- whatever the phase p0 -- and the four waves get whole 64-output turns of them
static inline unsigned dsm_span(int ds)
{
	const unsigned unit = (ds <= 18 ? 256u : 64u) * (unsigned)ds;
	return (DSM_SPAN_MAX / unit) * unit;
}

// NR: rounds of 256 vectors that cover the staged range (4 or 5)
template <bool ROTATE, int NR>
__global__ __launch_bounds__(256) void k_fm_decimate_small(
	const u32x4 *__restrict__ iq, u64 T, int ds, int p0, u64 M, int16_t *__restrict__ pcm, int pcm_chl2, unsigned span, unsigned span_windows)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t dsm_sm[];   // DSM_HALO + span + DSM_HALO + 8 (the generic window sum's over-read)
	uint32_t *const sm = dsm_sm;
	// XCD-contiguous spans, like k_fm_decimate (neighbouring spans write pieces of the same lines of the tiled output: they have to
	// meet in one L2); the grid is rounded up to a multiple of 8, the few workgroups past the last span leave
	const unsigned per = gridDim.x >> 3;
	const unsigned wgi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	const u64 wg0 = (u64)wgi * span;
	if (wg0 >= T)
		return;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const scale_k K = scale_consts();
	// stage [wg0 - HALO, wg0 + span + HALO): vector v holds samples wg0 - HALO + 4v .. +3 (their rotation phases are 0..3: wg0
	// and HALO are multiples of 4); outside the run: zeros (never used by a window that is produced here)
	const int NV = (int)(span + 2 * DSM_HALO) / 4;             // <= 1168 vectors: five rounds of 256
	u32x4 w[NR];
	if (wg0 >= DSM_HALO && wg0 + span + DSM_HALO <= T) {      // an inner span: no bounds to test per load
		const u32x4 *base = iq + ((wg0 - DSM_HALO) >> 2);
#pragma unroll
		for (int u = 0; u < NR; u++)                          // all in flight: vectors past the staged range re-read its last one, unused
			w[u] = __builtin_nontemporal_load(base + min(tid + 256 * u, NV - 1));
	} else {
#pragma unroll
		for (int u = 0; u < NR; u++) {
			const int v = tid + 256 * u;
			const i64 pos = (i64)wg0 - DSM_HALO + 4 * (i64)v;
			w[u] = (v < NV && pos >= 0 && (u64)pos < T) ? __builtin_nontemporal_load(iq + (pos >> 2)) : (u32x4)(0u);
		}
	}
#pragma unroll
	for (int u = 0; u < NR; u++) {
		const int v = tid + 256 * u;
		if (v < NV) {
			uint32_t s0, s1, s2, s3;
			dec_contrib<false, ROTATE>(w[u], s0, s1, s2, s3, K);
			*reinterpret_cast<uint4 *>(&sm[4 * v]) = make_uint4(s0, s1, s2, s3);
		}
	}
	__syncthreads();
	// windows that START in this span: m*ds - p0 in [wg0, wg0 + span).  The span is a whole number of windows (dsm_span), so the
	// first one is wgi * (span / ds), plus one when the run starts inside a window -- no division.  All per-output arithmetic
	// below is 32-bit, relative to the span's first window m_lo and to the tile of the output stream it falls in.
	const unsigned wps = span_windows;
	const u64 m_lo = (u64)wgi * wps + (p0 ? 1u : 0u);
	u64 m_hi64 = m_lo + wps;
	if (m_hi64 > M)
		m_hi64 = M;
	if (m_hi64 <= m_lo)
		return;
	const unsigned n_out = (unsigned)(m_hi64 - m_lo);                            // span / ds, fewer at the end of the run
	const int rel0 = (p0 ? ds - p0 : 0) + DSM_HALO;                              // LDS word of window m_lo's first sample
	const int rel_max = (int)span + DSM_HALO - 1;
	const unsigned tile_mask = pcm_chl2 ? ((64u << pcm_chl2) - 1u) : 0u;
	int16_t *const pcm_tile = pcm + (m_lo & ~(u64)tile_mask);
	const unsigned mt0 = (unsigned)(m_lo & (u64)tile_mask);
	const unsigned keep = ~tile_mask | 7u;
	// every wave takes a contiguous quarter of the span's outputs, 64 consecutive ones per turn: the previous output the
	// discriminator needs is the neighbouring lane's sum, the one that crosses a turn travels in an SGPR, and only the wave's very
	// first predecessor is summed again (its window starts ds samples earlier, inside the left halo at worst)
	const unsigned per_wave = (n_out + 3) >> 2;
	const unsigned j_lo = (unsigned)wave * per_wave, j_hi = min(n_out, j_lo + per_wave);
	if (j_lo >= j_hi)
		return;
	uint32_t b0 = 0;
	if (lane == 0)
		b0 = dsm_window(sm, rel0 + (int)__umul24(j_lo, (unsigned)ds) - ds, ds);
	b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
	for (unsigned j0 = j_lo; j0 < j_hi; j0 += 64) {
		const unsigned j = j0 + lane;
		// lanes past the last output stay inside the staged range and are not stored
		const int rel = rel0 + (int)__umul24(j, (unsigned)ds);
		const uint32_t a = dsm_window(sm, min(rel, rel_max), ds);
		const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp((int)b0, (int)a, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps b0
		b0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
		if (j >= j_hi)
			continue;
		int cr, cj;
		mul_conj_pk(a, b, cr, cj);
		// |lowpassed| <= 128 * ds: for ds <= 16 the discriminator's denominator stays below 2^24
		const int16_t v = (int16_t)fast_atan2_dev<false>(cj, cr);
		if (pcm_chl2) {
			// 32-bit form of pcm_index (two bit-field moves): the bits above a tile pass through, so an output that runs into the
			// next tile lands there
			const unsigned mt = mt0 + j;
			unsigned idx = mt & keep;
			idx |= __builtin_amdgcn_ubfe(mt, 3u, (unsigned)pcm_chl2 - 3u) << 9;
			idx |= __builtin_amdgcn_ubfe(mt, (unsigned)pcm_chl2, 6u) << 3;
			pcm_tile[idx] = v;
		} else {
			__builtin_nontemporal_store(v, &pcm_tile[j]);
		}
	}
}

// ------------------------------------------------------------------ F0+F1+F2+F5/F6 for small decimation, a lane owns whole windows
//
// The barrier-free form of the kernel above (round 6).  A lane owns L = W * ds consecutive samples -- W = 2 windows for even ds, 4 for
// odd, so that L is a whole number of 16-byte vectors -- starting at the 16-byte boundary A(G) = L G - p0 + RP, RP = p0 % 4: the
// rotation phase of every register (rotate16_90, rtl_fm.c:309-327: position mod 4) and its place in a window are compile-time constants.
// A wave brings a tile of 64 L samples in with NP = L / 4 LDS-DMA instructions (whole lines, wave-private LDS, no barrier anywhere),
// every lane reads its NP vectors with ds_read_b128 (lane stride 4 NP dwords: conflict-free for odd NP), scales them with scale_pk and adds
// them straight into W + 1 packed accumulators: the head of window W G (its first RP samples sit in the lane to the left), W - 1 whole
// windows, and the RP samples at the end that begin window W (G + 1).  That tail and the lane's last window travel one lane to the right
// (wave_shr:1) to complete the neighbour's first window and to be its discriminator's predecessor; what crosses a tile goes through two
// SGPRs, and the first lane of a wave's first tile is a halo lane (1 / (64 tw) of the input read twice).  A lane stores its W consecutive
// int16 results in one 4- or 8-byte store (never across a 16-byte unit of the tiled layout).  What this kernel cannot know stays with
// k_fm_disc(sparse, seams = 2) exactly as for k_fm_decimate_small: the run's first two outputs, every block's libm sample, the carries.
#define DL_HALO 8
template <int DS> struct dl_geom {
	static constexpr int W = (DS & 1) ? 4 : 2, L = W * DS, NP = L / 4;
	// tiles in flight per wave (the ring of LDS stages): two while a workgroup's ring stays within 64 KiB
	static constexpr int NS = NP <= 8 ? 2 : 1;
};

// s_waitcnt vmcnt takes an immediate; the ring below needs one of three values per (NS, NP)
template <int N>
__device__ __forceinline__ void dl_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// NS: depth of the wave's ring of LDS stages = tiles in flight per wave
template <bool ROTATE, int DS, int RP, int NS = dl_geom<DS>::NS>
__global__ __launch_bounds__(256) void k_fm_decimate_lane(const uint32_t *__restrict__ iq, u64 T, int p0, u64 M, int16_t *__restrict__ pcm, int pcm_chl2,
                                                          unsigned tw, unsigned n_waves, int g_a)
{
	constexpr int W = dl_geom<DS>::W, L = dl_geom<DS>::L, NP = dl_geom<DS>::NP;
	static_assert(RP < DS && RP < 4, "the head of a lane's first window lies in the lane itself");
	static_assert((NS - 1) * NP + NS < 64, "vmcnt is a 6-bit counter");
	extern __shared__ __attribute__((aligned(16))) u32x4 dl_stage[];       // [4][NS][64 * NP] (+ the launcher's occupancy pad)
	const unsigned lane = threadIdx.x & 63u;
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned per = gridDim.x >> 3;                     // XCD-contiguous order, as everywhere: the pieces of a line of the tiled pcm meet in one L2
	const unsigned wgi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	const unsigned wave_id = wgi * 4u + wv;
	if (wave_id >= n_waves)
		return;
	u32x4 *const ring = dl_stage + wv * (NS * 64 * NP);
	const scale_k K = scale_consts();
	// the wave's lanes: G0 .. G0 + 64 tw - 1, the first DL_HALO of them halo lanes (their results belong to the wave before; one would do, eight
	// keep every wave's first lane at g_a mod 8 -- the lane number at which a tile starts on a 128-byte line, see dl_launch)
	const i64 G0 = (i64)wave_id * (i64)(64u * tw - DL_HALO) + (g_a - DL_HALO);
	const i64 Gtot = (i64)((M + W - 1) / W);
	const i64 T4 = (i64)T - 4;
	// tiles this wave walks: while the tile's first lane exists (the launcher's n_waves guarantees the first one does, and a tile's first
	// non-halo lane always stores -- the vmcnt arithmetic below counts ONE store instruction per tile)
	const i64 left = Gtot - G0;
	const unsigned nt = left >= (i64)(64u * tw) ? tw : (unsigned)((left + 63) >> 6);

	auto fetch = [&](i64 G, unsigned sidx) {
		const i64 A = L * G - p0 + RP;
		u32x4 *const stage = ring + sidx * (64 * NP);
		if (A >= 0 && A + 64 * L <= (i64)T) {                // an inner tile: one uniform base, the lane's offset a constant
			const uint32_t *base = iq + A;
#pragma unroll
			for (int h = 0; h < NP; h++)
				__builtin_amdgcn_global_load_lds((const void *)(base + 4 * (64 * h + (int)lane)), (__attribute__((address_space(3))) void *)(stage + 64 * h), 16, 0, 2);
		} else {                                             // the run's two ends: vectors outside it repeat its first / last one; they only reach
#pragma unroll                                               // windows that k_fm_disc writes (the first two) or that do not exist (past M)
			for (int h = 0; h < NP; h++) {
				i64 sp = A + 4 * (64 * h + (int)lane);
				sp = sp < 0 ? 0 : (sp > T4 ? T4 : sp);
				__builtin_amdgcn_global_load_lds((const void *)(iq + sp), (__attribute__((address_space(3))) void *)(stage + 64 * h), 16, 0, 2);
			}
		}
	};

	const unsigned tile_mask = pcm_chl2 ? ((64u << pcm_chl2) - 1u) : 0u;
	const unsigned keep = ~tile_mask | 7u;
	uint32_t carry_t = 0, carry_a = 0;                       // the tail and the last window of the lane left of lane 0
	i64 G = G0;
#pragma unroll
	for (int k = 0; k < NS; k++)
		if ((unsigned)k < nt)
			fetch(G0 + 64 * k, (unsigned)k);
	unsigned sidx = 0;
#pragma unroll 1
	for (unsigned it = 0; it < nt; it++, G += 64) {
		// VMEM instructions retire in issue order.  Behind tile it's NP loads were issued: the loads of tiles it+1 .. it+NS-1 and one store
		// per tile computed since -- min(it, NS) of them; at the end of the walk, where fewer tiles are in flight, wait for everything.
		if (it + NS <= nt) {
			if (it >= NS) dl_wait_vm<(NS - 1) * NP + NS>(); else dl_wait_vm<(NS - 1) * NP>();
		} else {
			dl_wait_vm<0>();
		}
		u32x4 raw[NP];
		const u32x4 *const stage = ring + sidx * (64 * NP);
#pragma unroll
		for (int j = 0; j < NP; j++)
			raw[j] = stage[NP * lane + j];
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		// the stage is free as soon as its vectors sit in registers: tile it + NS goes into it
		if (it + NS < nt)
			fetch(G + 64 * NS, sidx);
		sidx = sidx + 1 == NS ? 0 : sidx + 1;
		uint32_t acc[W + 1];
#pragma unroll
		for (int k = 0; k <= W; k++)
			acc[k] = 0;
#pragma unroll
		for (int j = 0; j < NP; j++) {
			uint32_t x[4];
			dec_contrib<false, ROTATE>(raw[j], x[0], x[1], x[2], x[3], K);
#pragma unroll
			for (int q = 0; q < 4; q++) {
				constexpr int H = DS - RP;                       // samples of the lane's first window that lie in the lane
				const int pos = 4 * j + q, seg = pos < H ? 0 : (pos - H) / DS + 1;
				// the first sample of a segment starts its sum
				const bool first = pos == 0 || pos == H || (pos > H && (pos - H) % DS == 0);
				acc[seg] = first ? x[q] : pk_add(acc[seg], x[q]);
			}
		}
		uint32_t a[W];
		if (RP)
			a[0] = pk_add(acc[0], (uint32_t)__builtin_amdgcn_update_dpp((int)carry_t, (int)acc[W], 0x138, 0xf, 0xf, false));   // wave_shr:1, lane 0 keeps the carry
		else
			a[0] = acc[0];
#pragma unroll
		for (int k = 1; k < W; k++)
			a[k] = acc[k];
		const uint32_t pred = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_a, (int)a[W - 1], 0x138, 0xf, 0xf, false);
		if (RP)
			carry_t = (uint32_t)__builtin_amdgcn_readlane((int)acc[W], 63);
		carry_a = (uint32_t)__builtin_amdgcn_readlane((int)a[W - 1], 63);
		// |lowpassed| <= 128 ds: for ds <= 16 the discriminator's denominator stays below 2^24
		int r[W];
#pragma unroll
		for (int k = 0; k < W; k++) {
			int cr, cj;
			mul_conj_pk(a[k], k ? a[k - 1] : pred, cr, cj);
			r[k] = fast_atan2_dev<(DS <= 16)>(cj, cr);
		}
		// results W (G + lane) .. + W - 1; tiled layout: the bits above a tile pass through (keep), so a wave that runs into the next tile lands there
		const i64 Mt = W * G;                                // uniform; -W for the run's very first (halo) lane
		const i64 Bt = Mt & ~(i64)tile_mask;
		const unsigned mt = (unsigned)(Mt - Bt) + W * lane;
		const i64 m0 = Mt + (i64)(W * lane);
		unsigned idx = mt;
		if (pcm_chl2) {
			idx = mt & keep;
			idx |= __builtin_amdgcn_ubfe(mt, 3u, (unsigned)pcm_chl2 - 3u) << 9;
			idx |= __builtin_amdgcn_ubfe(mt, (unsigned)pcm_chl2, 6u) << 3;
		}
		int16_t *dst = pcm + Bt + idx;
		// not the halo lane, not the lanes past the run's last window; a last lane that holds fewer than W windows stores W all the same (the
		// buffer ends in a spare tile, nothing reads past M): one store instruction per tile, whatever the lane
		if ((it > 0 || lane >= DL_HALO || wave_id == 0) && m0 >= 0 && m0 < (i64)M) {
			if constexpr (W == 2)
				*reinterpret_cast<uint32_t *>(dst) = (uint32_t)(uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16);
			else
				*reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)(uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16), (uint32_t)(uint16_t)r[2] | ((uint32_t)(uint16_t)r[3] << 16));
		}
	}
}


// One sample, scaled and rotated by its position in its block (exact int)
template <bool PRESCALED>
__device__ __forceinline__ void load_rot(const uint32_t *iq, u64 pos, unsigned phase, int &ri, int &rq)
{
	const uint32_t w = iq[pos];
	int i = lo16(w), q = hi16(w);
	if (!PRESCALED) { i = scale_cs16(i); q = scale_cs16(q); }
	switch (phase & 3) {
	case 0: ri = i; rq = q; break;
	case 1: ri = -q; rq = i; break;
	case 2: ri = -i; rq = -q; break;
	default: ri = q; rq = -i; break;
	}
}

// Generic form: one thread per output, any ds, any block length (rotation phase = position
// in the block, rtl_fm.c:315).
template <bool PRESCALED>
__global__ void k_fm_decimate_generic(const uint32_t *__restrict__ iq, u64 T, int ds, int p0, u64 n_per_block,
                                      int rotate, const rxk_fm_dev *__restrict__ dev, uint32_t *__restrict__ lp, u64 M)
{
	const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= M)
		return;
	u64 start = m ? m * (u64)ds - (u64)p0 : 0;
	const u64 end = (m + 1) * (u64)ds - (u64)p0;
	int si = 0, sq = 0;
	if (m == 0) { si = dev->in_now_r; sq = dev->in_now_j; }
	u64 inblk = start % n_per_block;
	for (u64 pos = start; pos < end; pos++) {
		int ri, rq;
		load_rot<PRESCALED>(iq, pos, rotate ? (unsigned)inblk : 0u, ri, rq);
		si += ri; sq += rq;
		if (++inblk == n_per_block) inblk = 0;
	}
	lp[m] = pack_iq(si, sq);
}

// ------------------------------------------------------------------ F5/F6 discriminator

// C's truncating int32 division.  Fast path: the quotient from one fp32 reciprocal (v_rcp_f32, 1 ulp) is within 2^-21
// relative of the true one, so for |quotient| < 2^20 its truncation is off by at most one, which the exact 32-bit
// remainder settles (|den| < 2^30 keeps that remainder inside an int32).  Anything else -- quotients the int32 wrap of
// fast_atan2's numerator can produce -- takes one correctly rounded fp64 division: a non-integer quotient of two int32
// lies at least 1/|den| away from the next integer while the rounding error is below 2^-22/|den|.
__device__ __forceinline__ int div_trunc(int num, int den)
{
	const unsigned un = num < 0 ? 0u - (unsigned)num : (unsigned)num;
	const unsigned ud = den < 0 ? 0u - (unsigned)den : (unsigned)den;
	const float fq = (float)un * __builtin_amdgcn_rcpf((float)ud);
	if (__builtin_expect(ud >= (1u << 30) || !(fq < 1048576.0f), 0))
		return (int)((double)num / (double)den);
	unsigned q = (unsigned)fq;
	const int r = (int)(un - q * ud);
	q = r < 0 ? q - 1 : ((unsigned)r >= ud ? q + 1 : q);
	return ((num ^ den) < 0) ? -(int)q : (int)q;
}

// rtl_fm.c:485-506 with the int32 wrap of `pi4 * (x -/+ yabs)` and C's truncating division.  DEN24: the caller guarantees
// |x| + |y| < 2^24 (no wrap of the denominator, and the remainder's product is one full-rate 24-bit multiply instead of the
// quarter-rate v_mul_lo_u32)
template <bool DEN24>
__device__ __forceinline__ int fast_atan2_dev(int y, int x)
{
	if (x == 0 && y == 0)
		return 0;
	const unsigned ux = (unsigned)x;
	const unsigned ay = y < 0 ? 0u - (unsigned)y : (unsigned)y;
	// x >= 0: pi/4 - pi/4 * (x - |y|) / (x + |y|);  x < 0: 3pi/4 - pi/4 * (x + |y|) / (|y| - x)  -- one division site,
	// so lanes of both kinds do not walk the division twice
	const bool neg = x < 0;
	const int num = (int)(4096u * (neg ? ux + ay : ux - ay));
	const int den = (int)(neg ? ay - ux : ux + ay);
	// den = |x| + |y| is positive unless that sum wrapped: one test sends a wrapped (or huge) denominator and an oversized
	// quotient to the exact fp64 division, everything else needs no |den| and no sign of den
	int q;
	const unsigned un = num < 0 ? 0u - (unsigned)num : (unsigned)num;
	const float fq = (float)un * __builtin_amdgcn_rcpf((float)(unsigned)den);
	if (__builtin_expect((!DEN24 && (unsigned)den >= (1u << 30)) || !(fq < 1048576.0f), 0)) {
		q = (int)((double)num / (double)den);
	} else {
		unsigned uq = (unsigned)fq;
		const int r = (int)(un - (DEN24 ? __umul24(uq, (unsigned)den) : uq * (unsigned)den));
		uq = r < 0 ? uq - 1 : ((unsigned)r >= (unsigned)den ? uq + 1 : uq);
		q = num < 0 ? -(int)uq : (int)uq;
	}
	const int ang = (neg ? 12288 : 4096) - q;
	return y < 0 ? -ang : ang;
}

// multiply(a, conj(b)) on packed (re, im) int16 pairs, rtl_fm.c:470-474 via 480/511, wrapping like -fwrapv:
// cr = ar*br + aj*bj is one v_dot2_i32_i16, cj = aj*br - ar*bj two v_mad_i32_i16 (op_sel picks the halves) and a subtraction
__device__ __forceinline__ void mul_conj_pk(uint32_t a, uint32_t b, int &cr, int &cj)
{
	cr = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), 0, false);
	int t, u;
	asm("v_mad_i32_i16 %0, %2, %3, 0 op_sel:[1,0,0,0]\n\tv_mad_i32_i16 %1, %2, %3, 0 op_sel:[0,1,0,0]" : "=&v"(t), "=&v"(u) : "v"(a), "v"(b));
	cj = (int)((unsigned)t - (unsigned)u);
}

// rtl_fm.c:528-564 on the already multiplied (cr, cj); atan_lut from atan_lut_init (515-526, host libm)
__device__ __forceinline__ int polar_disc_lut_dev(int cr, int cj, const int *__restrict__ lut)
{
	if (cr == 0 || cj == 0) {
		if (cr == 0 && cj == 0) return 0;
		if (cr == 0 && cj > 0) return 1 << 13;
		if (cr == 0 && cj < 0) return -(1 << 13);
		if (cj == 0 && cr > 0) return 0;
		return 1 << 14;
	}
	const int x = (int)((unsigned)cj << 8) / cr;
	const int xa = x < 0 ? -x : x;
	if (xa >= 131072)
		return cj > 0 ? 1 << 13 : -(1 << 13);
	if (x > 0)
		return cj > 0 ? lut[x] : lut[x] - (1 << 14);
	return cj > 0 ? (1 << 14) - lut[-x] : -lut[-x];
}

// rtl_fm.c:566-582, wrapping like -fwrapv
__device__ __forceinline__ int esbensen_dev(int ar, int aj, int br, int bj)
{
	const int dr = (int)(((unsigned)br - (unsigned)ar) * 2u);
	const int dj = (int)(((unsigned)bj - (unsigned)aj) * 2u);
	const int cj = (int)((unsigned)bj * (unsigned)dr - (unsigned)br * (unsigned)dj);
	const int den = (int)((unsigned)ar * (unsigned)ar + (unsigned)aj * (unsigned)aj + 1u);
	return (int)(2608u * (unsigned)cj) / den;
}

// window m = stream samples [m*ds - p0, (m+1)*ds - p0), summed again from the capture (exact int, then int16 like low_pass's
// store); only for windows that lie completely inside this run
template <bool PRESCALED>
__device__ __forceinline__ uint32_t lp_brute(const uint32_t *iq, u64 m, int ds, int p0, u64 n_per_block, int rotate)
{
	const u64 start = m * (u64)ds - (u64)p0;
	int si = 0, sq = 0;
#pragma unroll 8
	for (int i = 0; i < ds; i++) {                            // independent loads: eight in flight
		int ri, rq;
		// the fast decimator only takes blocks of a multiple of 4 samples, so the rotation phase (position in the
		// block, rtl_fm.c:315) is the stream position mod 4
		load_rot<PRESCALED>(iq, start + (u64)i, rotate ? (unsigned)(start + (u64)i) : 0u, ri, rq);
		si += ri; sq += rq;
	}
	return pack_iq(si, sq);
}

// Is window m one of the two entries per span that rxk_fm_decimate(lp_sparse) stored -- the second window ending in
// its span, or the last one?  (The first is in head/tail form, handled before this is asked.)
__device__ __forceinline__ bool lp_sparse_stored(u64 m, int ds, int p0, u64 M)
{
	const u64 e1 = (m + 1) * (u64)ds - (u64)p0 - 1;                   // last sample of the window
	const u64 g = e1 >> RXK_DEC_SPAN_LOG2;
	const u64 m_g = ((g << RXK_DEC_SPAN_LOG2) + (u64)p0) / (u64)ds;   // the window holding the span's first sample
	u64 m_next = (((g + 1) << RXK_DEC_SPAN_LOG2) + (u64)p0) / (u64)ds;
	if (m_next > M)
		m_next = M;
	return m == m_g + 1 || m + 1 == m_next;
}

// stored: lp_raw[m] is known to have been written (< 0: find out); otherwise (lp_sparse) the window is summed again
template <bool PRESCALED>
__device__ __forceinline__ uint32_t lp_final(u64 m, int ds, int p0, int seams, const uint32_t *lp_raw,
                                             const uint32_t *head, const uint32_t *tail, uint32_t carry,
                                             int stored, u64 M, const uint32_t *iq, u64 n_per_block, int rotate, bool *brute)
{
	*brute = false;
	if (!seams)
		return lp_raw[m];
	if (seams == 2) {
		// after k_fm_decimate_small with a sparse lowpassed[]: nothing is stored, every window this kernel needs is summed again;
		// the run's first window may begin in the previous run -- those samples are the carried now_r/now_j
		*brute = true;
		const i64 w0 = (i64)(m * (u64)ds) - (i64)p0;
		if (w0 >= 0)
			return lp_brute<PRESCALED>(iq, m, ds, p0, n_per_block, rotate);
		int si = 0, sq = 0;
		for (i64 pos = 0; pos < w0 + ds; pos++) {
			int ri, rq;
			load_rot<PRESCALED>(iq, (u64)pos, rotate ? (unsigned)pos : 0u, ri, rq);
			si += ri; sq += rq;
		}
		return pk_add(carry, pack_iq(si, sq));
	}
	// window m covers stream samples [m*ds - p0, (m+1)*ds - p0); it is the one the decimator left
	// in head/tail form iff it is the first window ENDING inside its workgroup span, i.e. iff it
	// starts at or before that span's first sample
	const i64 w0 = (i64)(m * (u64)ds) - (i64)p0;
	const u64 g = ((u64)(w0 + ds - 1)) >> RXK_DEC_SPAN_LOG2;
	if (w0 <= (i64)(g << RXK_DEC_SPAN_LOG2))
		return pk_add(g ? tail[g - 1] : carry, head[g]);
	if (stored > 0 || (stored < 0 && lp_sparse_stored(m, ds, p0, M)))
		return lp_raw[m];
	*brute = true;
	return lp_brute<PRESCALED>(iq, m, ds, p0, n_per_block, rotate);
}

// $RXGPU_FLAG_ALL (test hook): 1 and 2 hand EVERY libm sample to the host (2 stores a wrong value first, so that only the host's
// re-evaluation can make it right), 3 does what 2 does to a pseudo-random eighth of them -- a pipelined sequence then mixes runs with
// and without fix-ups
__device__ __forceinline__ bool flag_forced(int flag_all, u64 m)
{
	return flag_all == 3 ? (((unsigned)m * 2654435761u) >> 29) == 0u : flag_all != 0;
}

// grid: ceil(M/256) blocks for the outputs (+1 block for the exact low_pass tail sums)
template <bool PRESCALED>
__global__ __launch_bounds__(256) void k_fm_disc(
	const uint32_t *__restrict__ iq, u64 T, int ds, int p0, u64 n_per_block, int rotate, int seams,
	const uint32_t *lp_raw, const uint32_t *__restrict__ head, const uint32_t *__restrict__ tail,
	uint32_t *lp /* may alias lp_raw: seam entries are finished in place */, u64 M, int first_mode, u64 uniform_k, int custom_atan, int do_tail,
	int16_t *__restrict__ pcm, rxk_fm_dev *__restrict__ dev, rxk_flag_rec *__restrict__ flag_list, int *__restrict__ flag_cnt,
	unsigned out_blocks, int sparse, u64 n_wg, u64 n_blocks, const int *__restrict__ atan_lut, int lp_sparse, int flag_all, int pcm_chl2)
{
	if (blockIdx.x >= out_blocks) {
		// ---- low_pass carry: exact int32 sums of the samples after the last complete window
		__shared__ int red[2][4];
		const u64 done = M ? M * (u64)ds - (u64)p0 : 0;        // samples consumed by complete windows
		int si = 0, sq = 0;
		for (u64 pos = done + threadIdx.x; pos < T; pos += 256) {
			int ri, rq;
			load_rot<PRESCALED>(iq, pos, rotate ? (unsigned)(pos % n_per_block) : 0u, ri, rq);
			si += ri; sq += rq;
		}
		for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
		if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = si; red[1][threadIdx.x >> 6] = sq; }
		__syncthreads();
		if (threadIdx.x == 0) {
			si = red[0][0] + red[0][1] + red[0][2] + red[0][3];
			sq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
			if (M == 0) { si += dev->in_now_r; sq += dev->in_now_j; }
			dev->out_now_r = si;
			dev->out_now_j = sq;
			dev->out_prev_index = (int)((u64)p0 + T - M * (u64)ds);
		}
		return;
	}
	u64 m = (u64)blockIdx.x * 256 + threadIdx.x;
	int a_stored = lp_sparse ? -1 : 1, b_stored = a_stored;  // lp_sparse: only a span's second and last outputs are in lp_raw (-1: look)
	if (sparse) {
		// only what k_fm_decimate<DISC> could not finish: the first two windows ending in each
		// workgroup span, the first window of each callback block (libm), and the very last one (carry)
		const u64 t = m;
		if (t < 2 * n_wg) {
			const u64 g = t >> 1;
			m = ((g << RXK_DEC_SPAN_LOG2) + (u64)p0) / (u64)ds + (t & 1);
			if (m >= M || (((m + 1) * (u64)ds - (u64)p0 - 1) >> RXK_DEC_SPAN_LOG2) != g)
				return;
			a_stored = 1;                            // t odd: the span's second output; t even: head/tail form
			b_stored = 1;                            // t odd: head/tail form; t even: the previous span's last output
		} else if (t < 2 * n_wg + n_blocks) {
			m = ((t - 2 * n_wg) * n_per_block + (u64)p0) / (u64)ds;
		} else if (t == 2 * n_wg + n_blocks) {
			m = M - 1;
		} else {
			return;
		}
	}
	if (m >= M)
		return;
	const uint32_t carry = pack_iq(dev->in_now_r, dev->in_now_j);
	bool brute;
	const uint32_t a = lp_final<PRESCALED>(m, ds, p0, seams, lp_raw, head, tail, carry, a_stored, M, iq, n_per_block, rotate, &brute);
	if (seams)
		lp[m] = a;
	if (!pcm) {                                  // lp_only: squelch / another demodulator comes next
		if (m == M - 1) { dev->out_pre_r = dev->in_pre_r; dev->out_pre_j = dev->in_pre_j; }
		return;
	}
	int br, bj;
	if (m) {
		const uint32_t b = lp_final<PRESCALED>(m - 1, ds, p0, seams, lp_raw, head, tail, carry, b_stored, M, iq, n_per_block, rotate, &brute);
		if (brute)
			lp[m - 1] = b;                           // the host re-reads both for a flagged libm sample
		br = lo16(b); bj = hi16(b);
	} else {
		br = dev->in_pre_r; bj = dev->in_pre_j;
	}
	const int ar = lo16(a), aj = hi16(a);
	// multiply(a, conj(b)), rtl_fm.c:470-474 via 480/511, wrapping like -fwrapv (pre_r/pre_j are full ints: plain products)
	const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);

	// first output of a callback block (rtl_fm.c:588-590)?
	bool first;
	if (first_mode == RXK_FIRST_UNIFORM) {
		first = (uniform_k & (uniform_k - 1)) ? (m % uniform_k) == 0 : (m & (uniform_k - 1)) == 0;
	} else {
		// the window ends in block b; it is that block's first iff it starts at or before the block
		const i64 w0 = (i64)(m * (u64)ds) - (i64)p0;
		const u64 e1 = (u64)(w0 + ds - 1);
		const u64 b = (n_per_block & (n_per_block - 1)) ? e1 / n_per_block : e1 >> (63 - __clzll((long long)n_per_block));
		first = w0 <= (i64)(b * n_per_block);
	}
	int out;
	if (first || custom_atan == 0) {
		// polar_discriminant, rtl_fm.c:476-483: (int)(atan2(cj,cr) / 3.14159 * (1<<14))
		const double ang = atan2((double)cj, (double)cr);
		const double v = ang / 3.14159 * 16384.0;
		out = (int)v;
		if (v != 0.0 && (flag_forced(flag_all, m) || fabs(v - rint(v)) < RXK_LIBM_WINDOW)) {
			const int idx = atomicAdd(flag_cnt, 1);
			if (idx < RXK_FLAG_CAP) {
				rxk_flag_rec r;
				r.m = m; r.ar = ar; r.aj = aj; r.br = br; r.bj = bj;
				flag_list[idx] = r;
			}
			if (flag_all > 1)
				out += 77;                               // test hook: only the host's re-evaluation can make this sample right
		}
	} else if (custom_atan == 1) {
		out = fast_atan2_dev(cj, cr);
	} else if (custom_atan == 2) {
		out = polar_disc_lut_dev(cr, cj, atan_lut);
	} else {
		out = esbensen_dev(ar, aj, br, bj);
	}
	pcm[pcm_index(m, pcm_chl2)] = (int16_t)out;
	if (m == M - 1) {
		dev->out_pre_r = ar;
		dev->out_pre_j = aj;
	}
}

// ------------------------------------------------------------------ F2 + F5/F6 of ONE callback block (the drop-in's latency path)

// full_demod() on a single block through rxgpu_fm_stream_run costs a dozen launches over four streams, carries up and down
// (84 us of host time for 1 MiB, round 3).  One block of the plain chain -- low_pass (rtl_fm.c:351-371) on the block the callback
// pre-staged, then fm_demod (584-615) -- is one launch here: a wave per decimated sample sums its window (and its predecessor's,
// again: there is no neighbour to wait for), lane 0 stores lowpassed[m] and the discriminator's sample; the wave behind the last
// window leaves low_pass's carry.  All carries come in as arguments and go out in `out`, which the host reads back together with
// the rows; the de-emphasis / resampler seeds are passed on to k_ch_audio (one workgroup, the next launch) through audio_in.
// blk: the block as the callback left it (scaled, rotated); n complex samples; p0 = prev_index.
__device__ __forceinline__ void blk_window(const uint32_t *__restrict__ blk, long s0, long e0, unsigned lane, int &si, int &sq)
{
	si = 0; sq = 0;
	for (long k = s0 + lane; k < e0; k += 64) {
		const uint32_t w = blk[k];
		si += lo16(w); sq += hi16(w);
	}
	for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
}

__global__ __launch_bounds__(256) void k_fm_block_dd(const uint32_t *__restrict__ blk, unsigned n, int ds, int p0, int now_r, int now_j,
                                                     int pre_r, int pre_j, int custom_atan, int flag_all, uint32_t *__restrict__ lp,
                                                     uint32_t *__restrict__ lp_host, int16_t *__restrict__ pcm, int16_t *__restrict__ keep, rxk_blk_out *__restrict__ out,
                                                     int *__restrict__ audio_in, int avg, int now_lpr, int prev_lpr_index)
{
	const unsigned lane = threadIdx.x & 63u;
	const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
	const long M = ((long)p0 + (long)n) / ds;
	if (gw == 0 && lane == 0) {
		audio_in[0] = avg; audio_in[1] = now_lpr; audio_in[2] = prev_lpr_index;
	}
	for (long m = gw; m <= M; m += nw) {
		// window m: samples [m ds - p0, (m + 1) ds - p0) of the block, the first one on top of the carried sums
		const long s0 = m * ds - p0;
		int si, sq;
		blk_window(blk, s0 < 0 ? 0 : s0, m == M ? (long)n : s0 + ds, lane, si, sq);
		if (m == 0) { si += now_r; sq += now_j; }
		if (m == M) {                                             // what is left behind the last complete window
			if (lane == 0) {
				out->now_r = si; out->now_j = sq; out->prev_index = (int)((long)p0 + (long)n - M * ds);
				if (M == 0) { out->pre_r = pre_r; out->pre_j = pre_j; }
			}
			continue;
		}
		int br = pre_r, bj = pre_j;
		if (m) {
			int ti, tq;
			blk_window(blk, s0 - ds < 0 ? 0 : s0 - ds, s0, lane, ti, tq);
			if (m == 1) { ti += now_r; tq += now_j; }
			br = (int16_t)ti; bj = (int16_t)tq;                 // lowpassed[] is int16 (rtl_fm.c:363-364)
		}
		if (lane)
			continue;
		const int ar = (int16_t)si, aj = (int16_t)sq;
		lp[m] = pack_iq(ar, aj);
		lp_host[m] = pack_iq(ar, aj);                            // lowpassed[] for the caller, straight into the page-locked mirror
		const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
		const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
		int v;
		if (m == 0 || custom_atan == 0) {
			// polar_discriminant, rtl_fm.c:476-483: (int)(atan2(cj,cr) / 3.14159 * (1<<14)); undecided within RXK_LIBM_WINDOW: the host's libm
			const double ang = atan2((double)cj, (double)cr);
			const double r = ang / 3.14159 * 16384.0;
			v = (int)r;
			if (r != 0.0 && (flag_forced(flag_all, (u64)m) || fabs(r - rint(r)) < RXK_LIBM_WINDOW)) {
				const int idx = atomicAdd(&out->flag_cnt, 1);
				if (idx < RXK_BLK_FLAGS) {
					rxk_flag_rec rec;
					rec.m = (u64)m; rec.ar = ar; rec.aj = aj; rec.br = br; rec.bj = bj;
					out->rec[idx] = rec;
				}
				if (flag_all > 1)
					v += 77;                                        // test hook: only the host's re-evaluation can make this sample right
			}
		} else if (custom_atan == 1) {
			v = fast_atan2_dev(cj, cr);
		} else {
			v = esbensen_dev(ar, aj, br, bj);
		}
		pcm[m] = (int16_t)v;
		keep[m] = (int16_t)v;                                   // the row as demodulated: what the audio stages start again from after a host fix-up
		if (m == M - 1) { out->pre_r = ar; out->pre_j = aj; }
	}
}

// ------------------------------------------------------------------ squelch, am/usb/lsb, dc block

__device__ __forceinline__ void block_range(const rxk_fm_blocks &g, u64 b, u64 &m0, u64 &m1)
{
	if (g.first_mode == RXK_FIRST_UNIFORM) {
		m0 = b * g.k; m1 = m0 + g.k;
	} else {
		m0 = (b * g.n + (u64)g.p0) / (u64)g.ds;
		m1 = ((b + 1) * g.n + (u64)g.p0) / (u64)g.ds;
	}
	if (g.post > 1) {                  // after low_pass_simple: every block's count is a multiple of post
		m0 /= (u64)g.post;
		m1 /= (u64)g.post;
	}
}

// floor(sqrt(v)) for v >= 0, exact whatever the last bit of the device sqrt does
__device__ __forceinline__ i64 isqrt_floor(double v)
{
	i64 r = (i64)sqrt(v);
	while (r > 0 && (double)(r * r) > v) r--;
	while ((double)((r + 1) * (r + 1)) <= v) r++;
	return r;
}

// rtl_fm.c:781-790 with rms() 739-757 (step 1, over both components): one workgroup per callback block
__global__ __launch_bounds__(256) void k_fm_squelch(uint32_t *__restrict__ lp, rxk_fm_blocks g, int level, int *__restrict__ below, int *__restrict__ sr_out)
{
	__shared__ i64 red[8];
	__shared__ int quiet;
	const u64 b = blockIdx.x;
	u64 m0, m1;
	block_range(g, b, m0, m1);
	i64 t = 0, p = 0;
	for (u64 m = m0 + threadIdx.x; m < m1; m += 256) {
		const uint32_t w = lp[m];
		const i64 i = lo16(w), q = hi16(w);
		t += i + q;
		p += i * i + q * q;
	}
	for (int off = 32; off; off >>= 1) { t += __shfl_down(t, off); p += __shfl_down(p, off); }
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = t; red[4 + (threadIdx.x >> 6)] = p; }
	__syncthreads();
	if (threadIdx.x == 0) {
		t = red[0] + red[1] + red[2] + red[3];
		p = red[4] + red[5] + red[6] + red[7];
		const int len = (int)(2 * (m1 - m0));
		const double dc = (double)t / (double)len;          // (double)(t*step)/(double)len, step == 1
		const double lhs = (double)(t * 2) * dc;
		const double rhs = dc * dc * (double)len;
		const double v = ((double)p - (lhs - rhs)) / (double)len;
		const int sr = v >= 0.0 ? (int)isqrt_floor(v) : 0;
		quiet = sr < level;
		below[b] = quiet;
		sr_out[b] = sr;                                      // the `sr` of full_demod (rtl_fm.c:781): what -L prints (rtl_fm.c:792-807)
	}
	__syncthreads();
	if (quiet)
		for (u64 m = m0 + threadIdx.x; m < m1; m += 256)
			lp[m] = 0;
}

// am_demod / usb_demod / lsb_demod, rtl_fm.c:617-656
__global__ void k_fm_simple_demod(const uint32_t *__restrict__ lp, u64 M, int mode, int output_scale, int16_t *__restrict__ pcm)
{
	const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= M)
		return;
	const uint32_t w = lp[m];
	const int i = lo16(w), q = hi16(w);
	int v;
	if (mode == 1) {
		const int pw = (int)((unsigned)(i * i) + (unsigned)(q * q));    // may wrap at full scale, like the reference's int
		// (int16_t)sqrt(pcm): sqrt of a negative int is NaN, which the x86 conversion turns into 0
		v = pw < 0 ? 0 : (int)(short)isqrt_floor((double)pw);
	} else {
		v = (int)(short)(mode == 2 ? i + q : i - q);
	}
	pcm[m] = (int16_t)(v * output_scale);
}

// dc_block_audio_filter, rtl_fm.c:684-697: sum per block ...
__global__ __launch_bounds__(256) void k_fm_dc_sums(const int16_t *__restrict__ y, rxk_fm_blocks g, i64 *__restrict__ sums)
{
	__shared__ i64 red[4];
	u64 m0, m1;
	block_range(g, blockIdx.x, m0, m1);
	i64 t = 0;
	for (u64 m = m0 + threadIdx.x; m < m1; m += 256)
		t += y[m];
	for (int off = 32; off; off >>= 1) t += __shfl_down(t, off);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
	__syncthreads();
	if (threadIdx.x == 0)
		sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ... the recursion avg = (sum/len + dc_avg*c) / (c+1) block after block (C truncating divisions) ...
__global__ void k_fm_dc_scan(const i64 *__restrict__ sums, rxk_fm_blocks g, int c, int *__restrict__ avgs, rxk_fm_dev *__restrict__ dev)
{
	if (threadIdx.x || blockIdx.x)
		return;
	int dc = dev->in_dc_avg;
	for (u64 b = 0; b < g.n_blocks; b++) {
		u64 m0, m1;
		block_range(g, b, m0, m1);
		int avg = (int)(sums[b] / (i64)(int)(m1 - m0));
		avg = (int)((unsigned)avg + (unsigned)dc * (unsigned)c) / (c + 1);
		avgs[b] = avg;
		dc = avg;
	}
	dev->out_dc_avg = dc;
}

// ... and the subtraction with int16 wrap
__global__ void k_fm_dc_apply(int16_t *__restrict__ y, u64 M, rxk_fm_blocks g, const int *__restrict__ avgs)
{
	const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= M)
		return;
	const u64 md = g.post > 1 ? m * (u64)g.post : m;        // after -o: sample m came from demodulated samples md .. md+post-1
	u64 b;
	if (g.first_mode == RXK_FIRST_UNIFORM)
		b = md / g.k;
	else
		b = ((md + 1) * (u64)g.ds - (u64)g.p0 - 1) / g.n;
	y[m] = (int16_t)(y[m] - avgs[b]);
}

// ------------------------------------------------------------------ F8 de-emphasis

// avg += trunc((d +- a/2) / a), rtl_fm.c:674-679  ==  avg += sign(d) * floor((|d| + a/2) / a)
__device__ __forceinline__ int deemph_step(int avg, int x, int a, int h, unsigned magic)
{
	const int d = x - avg;
	const unsigned t = (unsigned)(d < 0 ? -d : d) + (unsigned)h;
	const unsigned q = a == 1 ? t : __umulhi(t, magic);
	return avg + (d < 0 ? -(int)q : (int)q);
}

// The same map with the sign handling folded into one biased unsigned division:
//   d > 0 : floor((d + h) / a)            d <= 0 : -floor((-d + h) / a) = floor((d - h + a - 1) / a)
// and a - 1 - h == h for odd a, h - 1 for even a.  With xb = x + h + bias*a staged once per sample,
//   avg' = avg + floor((xb - avg - (EVEN && x <= avg)) / a) - bias,
// 3 VALU ops per sample for odd a (13 at 170 kHz / 75 us, 19 at 240 kHz, 9 for 50 us).
// Valid for int16 x and avg (rxgpu_fm.c sends anything else to the serial kernel): the dividend is
// in [0, 2^18) and magic = floor(2^32/a) + 1 is exact there for a < 2^14.
template <bool EVEN>
__device__ __forceinline__ int deemph_step_b(int avg, int xb, int x, unsigned magic, int bias)
{
	unsigned t = (unsigned)(xb - avg);
	if (EVEN)
		t -= (x <= avg) ? 1u : 0u;
	return avg + (int)__umulhi(t, magic) - bias;
}

// The recurrence is a non-linear integer IIR, but the per-sample map avg -> avg' is monotone
// with slope 0 or 1.  So (1) trajectories started from the two ends of the possible state
// range sandwich the true one, and their gap shrinks by at least floor(gap/a) per sample:
// after `warm` samples fewer than GS (>= a) start states remain possible for a chunk; and
// (2) the map of a whole chunk restricted to those states is a table of <= GS entries.
// Tables compose associatively, which turns the serial recurrence into a tree scan:
//   k_fm_deemph_scan   one lane per chunk (the lowest candidate + a merge mask, see deemph_track), 64 chunks per
//                      workgroup; the workgroup composes its 64 chunk tables (tree level 0) and stores every
//                      chunk's start state for each candidate of the workgroup
//   k_fm_deemph_up     composite of DEEMPH_FAN tables of one level -> next level (long runs)
//   k_fm_deemph_top    one workgroup walks the top level from the carried state (sqrt split)
//   k_fm_deemph_down   start state of every table one level down
//   k_fm_deemph_apply  picks each chunk's start state and replays the chunk once -> output
#define DEEMPH_FAN 16

// Division by a: D24 (5 <= a < 256, dividend < 2^18) takes two full-rate 24-bit multiplies,
// floor(u/a) = ((u << 6) * (2^26/a + 1)) >> 32 exactly because u*a < 2^26; otherwise the 32-bit magic.
template <bool D24>
__device__ __forceinline__ unsigned deemph_div(unsigned t, unsigned magic)
{
	if (D24)
		return (unsigned)(((unsigned long long)((t << 6) & 0xffffffu) * (unsigned long long)(magic & 0xffffffu)) >> 32);
	return __umulhi(t, magic);
}
template <bool D24>
__device__ __forceinline__ unsigned deemph_mul(unsigned q, unsigned a)
{
	if (D24) {
		unsigned r;
		asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(q), "v"(a));
		return r;
	}
	return q * a;
}
template <bool EVEN, bool D24>
__device__ __forceinline__ int deemph_step_d(int avg, int xb, int x, unsigned magic, int bias)
{
	unsigned t = (unsigned)(xb - avg);
	if (EVEN)
		t -= (x <= avg) ? 1u : 0u;
	return avg + (int)deemph_div<D24>(t, magic) - bias;
}

// One LANE per chunk.  The still-possible start states of a chunk are consecutive integers lo..lo+g and
// stay consecutive (slope 0 or 1), and per sample at most ONE adjacent pair of them merges: the pair
// (v, v+1) collapses iff x - v sits on an edge of the rounding, i.e. (odd a) iff v == x - a/2 - 1 (mod a).
// So instead of carrying every candidate through the chunk, the lane carries the lowest one plus a
// bit mask of which original neighbours are still distinct: T[k] = lo_end + popcount(mask & ((1<<k)-1)).
// The warm-up before the chunk tracks just the two extreme trajectories.
template <int GS> struct deemph_mask { typedef u64 type; };
template <> struct deemph_mask<16> { typedef uint32_t type; };
__device__ __forceinline__ int deemph_popc(u64 v) { return __popcll(v); }
__device__ __forceinline__ int deemph_popc(uint32_t v) { return __popc(v); }

template <bool EVEN, bool D24, typename MASK>
__device__ __forceinline__ void deemph_track(int &lo, int &cnt, MASK &mask, int x, int a, int xoff, unsigned magic, int bias)
{
	const unsigned u = (unsigned)(x + xoff - lo);                  // x - lo + a/2 + bias*a  >= 0
	unsigned q = deemph_div<D24>(u, magic);
	const unsigned r1 = u - deemph_mul<D24>(q, (unsigned)a);       // u mod a
	int r;
	if (!EVEN) {
		// a = 2h+1: (x - h - 1 - lo) == u (mod a)
		r = (int)r1 + 1 < cnt ? (int)r1 : -1;
	} else {
		// a = 2h: edges at D == h (mod a) for D > 0 and D == h+1 (mod a) for D <= 0, D = x - v;
		// (x - h - lo) == u and (x - h - 1 - lo) == u - 1 (mod a)
		const unsigned r2 = r1 ? r1 - 1 : (unsigned)a - 1;
		r = -1;
		if ((int)r1 + 2 <= cnt && x - lo - (int)r1 > 0)
			r = (int)r1;
		else if ((int)r2 + 2 <= cnt && x - lo - (int)r2 <= 0)
			r = (int)r2;
		if (x <= lo && r1 == 0)
			q--;                                                   // the lowest candidate divides u - 1
	}
	if (__builtin_expect(r >= 0, 0)) {
		// distinct values r and r+1 become one: clear the r-th set bit of the mask
		MASK m2 = mask;
		for (int i = 0; i < r; i++)
			m2 &= m2 - 1;
		mask &= ~(m2 & (~m2 + 1));
		cnt--;
	}
	lo += (int)q - bias;
}

// Stage 64 consecutive chunks (+ the `warm` samples before the first) of pcm into LDS, coalesced; row r holds
// chunk first-1+r, rows are chunk/8+1 sixteen-byte units apart (odd: lane-per-row b128 reads hit every bank once).
__device__ __forceinline__ void deemph_stage(const int16_t *__restrict__ pcm, u64 M, int chunk_l2, int warm, u64 first,
                                             uint4 *lds, int lane)
{
	const int row_u = (1 << (chunk_l2 - 3)) + 1, ul2 = chunk_l2 - 3;
	const u64 p0 = first ? (first << chunk_l2) - (u64)warm : 0;
	u64 p1 = (first + 64) << chunk_l2;
	if (p1 > M)
		p1 = M;
	const uint4 *src = reinterpret_cast<const uint4 *>(pcm + p0);
	const unsigned nfull = (unsigned)((p1 - p0) >> 3);                    // whole 16-byte units; p0 % 8 == 0
	const unsigned u0 = (unsigned)((p0 >> 3) - ((first ? first - 1 : 0) << ul2));   // unit index of p0 counted from row 0 (row 1 if first == 0)
	const unsigned rbase = first ? 0 : 1;
	for (unsigned base = 0; base < nfull; base += 8 * 64) {
		uint4 w[8];
		unsigned g[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {                                     // eight loads in flight per lane
			unsigned u = base + j * 64 + lane;
			u = u < nfull ? u : nfull - 1;                                // past the end: repeat the last unit (same data, same slot)
			w[j] = src[u];
			g[j] = u + u0;
		}
#pragma unroll
		for (int j = 0; j < 8; j++)
			lds[((g[j] >> ul2) + rbase) * row_u + (g[j] & ((1u << ul2) - 1))] = w[j];
	}
	const unsigned tail = (unsigned)((p1 - p0) & 7);
	if (tail && lane == 0) {                                              // the ragged end of the run
		uint32_t ww[4] = {0, 0, 0, 0};
		const u64 p = p0 + ((u64)nfull << 3);
		for (unsigned k = 0; k < tail; k++)
			ww[k >> 1] |= (uint32_t)(uint16_t)pcm[p + k] << ((k & 1) * 16);
		const unsigned g = nfull + u0;
		lds[((g >> ul2) + rbase) * row_u + (g & ((1u << ul2) - 1))] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
	}
}

// Fast form of the step for odd a in 9..255 (the D24 range): the state is kept as N = ((a/2 - avg) << 6) + 32, so that
//   t6 = ((x - avg + a/2) << 6) + 32          one v_mad_i32_i16 straight from the packed sample (x * 64 + N),
//   q  = floor((x - avg + a/2) / a)            one signed 24-bit multiply-high, (t6 * (2^26/a + 1)) >> 32 -- the +32 (half
//                                              a unit) keeps it exact for negative dividends: |t6| < 2^23, error < 2^-9 < 1/(2a),
//   N' = N - 64 q                              one v_mad_i32_i24,
// and, for the scan, r6 = t6 - q * 64a = ((x - avg + a/2) mod a) * 64 + 32 -- the merge test's remainder.  For odd a
// the reference's truncating (d +- a/2) / a is floor((d + a/2) / a) for either sign of d (rtl_fm.c:675-679).
__device__ __forceinline__ int de_state(int avg, int h) { return ((h - avg) << 6) + 32; }
__device__ __forceinline__ int de_avg(int N, int h) { return h - (N >> 6); }

template <int HALF>
__device__ __forceinline__ void de_step(uint32_t w, int &N, unsigned m, int n64)
{
	int t, q;
	if (HALF == 0)
		asm("v_mad_i32_i16 %[t], %[w], 64, %[N]\n\tv_mul_hi_i32_i24 %[q], %[t], %[m]\n\tv_mad_i32_i24 %[N], %[q], %[n64], %[N]"
		    : [N] "+v"(N), [t] "=&v"(t), [q] "=&v"(q) : [w] "v"(w), [m] "s"(m), [n64] "s"(n64));
	else
		asm("v_mad_i32_i16 %[t], %[w], 64, %[N] op_sel:[1,0,0,0]\n\tv_mul_hi_i32_i24 %[q], %[t], %[m]\n\tv_mad_i32_i24 %[N], %[q], %[n64], %[N]"
		    : [N] "+v"(N), [t] "=&v"(t), [q] "=&v"(q) : [w] "v"(w), [m] "s"(m), [n64] "s"(n64));
}

template <int HALF>
__device__ __forceinline__ int de_step_r(uint32_t w, int &N, unsigned m, int n64, int na6)
{
	int t, q, r;
	if (HALF == 0)
		asm("v_mad_i32_i16 %[t], %[w], 64, %[N]\n\tv_mul_hi_i32_i24 %[q], %[t], %[m]\n\tv_mad_i32_i24 %[N], %[q], %[n64], %[N]\n\t"
		    "v_mad_i32_i24 %[r], %[q], %[na6], %[t]"
		    : [N] "+v"(N), [t] "=&v"(t), [q] "=&v"(q), [r] "=&v"(r) : [w] "v"(w), [m] "s"(m), [n64] "s"(n64), [na6] "s"(na6));
	else
		asm("v_mad_i32_i16 %[t], %[w], 64, %[N] op_sel:[1,0,0,0]\n\tv_mul_hi_i32_i24 %[q], %[t], %[m]\n\tv_mad_i32_i24 %[N], %[q], %[n64], %[N]\n\t"
		    "v_mad_i32_i24 %[r], %[q], %[na6], %[t]"
		    : [N] "+v"(N), [t] "=&v"(t), [q] "=&v"(q), [r] "=&v"(r) : [w] "v"(w), [m] "s"(m), [n64] "s"(n64), [na6] "s"(na6));
	return r;
}

// distinct candidates r and r+1 have become one: clear the r-th set bit of the mask
template <typename MASK>
__device__ __forceinline__ void de_merge(MASK &mask, int r)
{
	MASK m2 = mask;
	for (int i = 0; i < r; i++)
		m2 &= m2 - 1;
	mask &= ~(m2 & (~m2 + 1));
}

#define DEEMPH_WG_CHUNKS 64          // chunks per workgroup of scan/apply = fan of the first tree level

template <int GS, bool EVEN, bool D24>
__global__ __launch_bounds__(64) void k_fm_deemph_scan(
	const int16_t *__restrict__ pcm, u64 M, int a, unsigned magic, int bias, int chunk_l2, int warm, int lo0, int hi0,
	int *__restrict__ pre, int *__restrict__ p_tab, int *__restrict__ p_lo, int *__restrict__ p_gap,
	rxk_fm_dev *__restrict__ dev)
{
	extern __shared__ uint4 de_lds[];
	constexpr bool FAST = !EVEN && D24;                                 // odd a in 5..255: the three-instruction step
	const int lane = threadIdx.x, chunk = 1 << chunk_l2, row_u = chunk / 8 + 1;
	int *ltab = reinterpret_cast<int *>(de_lds + 65 * row_u);        // this workgroup's 64 chunk tables, then lo and gap
	int *llo = ltab + 64 * GS, *lgap = llo + 64;
	const u64 n_chunks = (M + chunk - 1) >> chunk_l2;
	const u64 first = (u64)blockIdx.x * 64;
	deemph_stage(pcm, M, chunk_l2, warm, first, de_lds, lane);
	__syncthreads();
	const u64 c = first + lane;
	if (c < n_chunks) {
		const u64 c0 = c << chunk_l2;
		const int n = (int)((M - c0) < (u64)chunk ? (M - c0) : (u64)chunk);
		const int xoff = a / 2 + bias * a;
		int lo, hi;
		if (c == 0) {                                     // the run's carried state
			lo = hi = dev->in_deemph_avg;
		} else {
			lo = lo0;
			hi = hi0;
			const uint4 *row = de_lds + lane * row_u;
			uint4 w = row[(chunk - warm) >> 3];
			if (FAST) {
				int nl = de_state(lo, a / 2), nh = de_state(hi, a / 2);
				for (int u = (chunk - warm) >> 3; u < (chunk >> 3); u++) {
					const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
					w = row[u + 1];
#pragma unroll
					for (int k = 0; k < 4; k++) {
						de_step<0>(ww[k], nl, magic, -64); de_step<0>(ww[k], nh, magic, -64);
						de_step<1>(ww[k], nl, magic, -64); de_step<1>(ww[k], nh, magic, -64);
					}
				}
				lo = de_avg(nl, a / 2);
				hi = de_avg(nh, a / 2);
			} else {
				for (int u = (chunk - warm) >> 3; u < (chunk >> 3); u++) {
					const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
					w = row[u + 1];
#pragma unroll
					for (int k = 0; k < 8; k++) {
						const int x = (k & 1) ? hi16(ww[k >> 1]) : lo16(ww[k >> 1]);
						lo = deemph_step_d<EVEN, D24>(lo, x + xoff, x, magic, bias);
						hi = deemph_step_d<EVEN, D24>(hi, x + xoff, x, magic, bias);
					}
				}
			}
		}
		int gap = hi - lo;
		if (gap >= GS) {                                  // excluded by `warm` (rxgpu_fm.c); checked anyway
			atomicExch(&dev->err, 1);
			gap = GS - 1;
		}
		typedef typename deemph_mask<GS>::type MASK;
		const int lo_start = lo;
		int cnt = gap + 1;                                // a lone candidate never passes the merge test: no special case
		MASK mask = (MASK)(((MASK)1 << gap) - 1);
		const uint4 *row = de_lds + (lane + 1) * row_u;
		const int nu = n >> 3;
		uint4 w = row[0];
		if (FAST) {
			int N = de_state(lo, a / 2), cm6 = gap << 6;  // r6 < cm6  <=>  remainder + 1 < number of distinct candidates
			const int na6 = -(a << 6);
			for (int u = 0; u < nu; u++) {
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				w = row[u + 1];                           // next unit (the pad unit after the last) while this one is walked
#pragma unroll
				for (int k = 0; k < 4; k++) {
					int r6 = de_step_r<0>(ww[k], N, magic, -64, na6);
					if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
					r6 = de_step_r<1>(ww[k], N, magic, -64, na6);
					if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
				}
			}
			if (n & 7) {                                  // the ragged end of the run
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				for (int k = 0; k < (n & 7); k++) {
					const int r6 = (k & 1) ? de_step_r<1>(ww[k >> 1], N, magic, -64, na6) : de_step_r<0>(ww[k >> 1], N, magic, -64, na6);
					if (r6 < cm6) { de_merge(mask, r6 >> 6); cm6 -= 64; }
				}
			}
			lo = de_avg(N, a / 2);
		} else {
			for (int u = 0; u < nu; u++) {
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				w = row[u + 1];
#pragma unroll
				for (int k = 0; k < 8; k++) {
					const int x = (k & 1) ? hi16(ww[k >> 1]) : lo16(ww[k >> 1]);
					deemph_track<EVEN, D24>(lo, cnt, mask, x, a, xoff, magic, bias);
				}
			}
			if (n & 7) {                                  // the ragged end of the run
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				for (int k = 0; k < (n & 7); k++) {
					const int x = (k & 1) ? hi16(ww[k >> 1]) : lo16(ww[k >> 1]);
					deemph_track<EVEN, D24>(lo, cnt, mask, x, a, xoff, magic, bias);
				}
			}
		}
		for (int k = 0; k <= gap; k++)
			ltab[lane * GS + k] = lo + deemph_popc((MASK)(mask & (MASK)(((MASK)1 << k) - 1)));
		llo[lane] = lo_start;
		lgap[lane] = gap;
	}
	__syncthreads();
	// first tree level: the composite of this workgroup's chunk tables, one lane per candidate of the first chunk;
	// on the way, every chunk's start state for each of those candidates (k_fm_deemph_apply picks one)
	const int nc = (int)((n_chunks - first) < 64 ? (n_chunks - first) : 64);
	for (int k = lane; k < GS; k += 64) {
		const int g0 = lgap[0];
		int v = llo[0] + (k < g0 ? k : g0);
		for (int i = 0; i < nc; i++) {
			pre[(first + i) * GS + k] = v;
			v = ltab[i * GS + (v - llo[i])];
		}
		p_tab[(u64)blockIdx.x * GS + k] = v;
		if (k == 0) { p_lo[blockIdx.x] = llo[0]; p_gap[blockIdx.x] = g0; }
	}
}

// one thread group of gs lanes per parent table: walk DEEMPH_FAN children (global, dependent)
__global__ void k_fm_deemph_up(u64 n_child, int gs, const int *__restrict__ tab, const int *__restrict__ lo,
                               const int *__restrict__ gap, int *__restrict__ p_tab, int *__restrict__ p_lo,
                               int *__restrict__ p_gap)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 parent = gid / gs;
	const int k = (int)(gid % gs);
	const u64 first = parent * DEEMPH_FAN;
	if (first >= n_child)
		return;
	const int cnt = (int)((n_child - first) < DEEMPH_FAN ? (n_child - first) : DEEMPH_FAN);
	const int g0 = gap[first];
	int v = lo[first] + (k < g0 ? k : g0);
	for (int i = 0; i < cnt; i++)
		v = tab[(first + i) * gs + (v - lo[first + i])];
	p_tab[parent * gs + k] = v;
	if (k == 0) { p_lo[parent] = lo[first]; p_gap[parent] = g0; }
}

// single workgroup: n tables staged in LDS, walked in sqrt(n) segments
__global__ __launch_bounds__(256) void k_fm_deemph_top(int n, int gs, const int *__restrict__ tab,
                                                         const int *__restrict__ lo, const int *__restrict__ gap,
                                                         int *__restrict__ start, rxk_fm_dev *__restrict__ dev)
{
	extern __shared__ __attribute__((aligned(16))) int sm[];
	int R = 1;
	while (R * R < n) R++;
	const int nseg = (n + R - 1) / R;
	int *t = sm;                      // [n][gs]
	int *l = t + n * gs;              // [n]
	int *g = l + n;                   // [n]
	int *segtab = g + n;              // [nseg][gs]
	int *segstart = segtab + nseg * gs;   // [nseg]
	for (int i = threadIdx.x; i < n * gs; i += blockDim.x) t[i] = tab[i];
	for (int i = threadIdx.x; i < n; i += blockDim.x) { l[i] = lo[i]; g[i] = gap[i]; }
	__syncthreads();
	for (int w = threadIdx.x; w < nseg * gs; w += blockDim.x) {
		const int seg = w / gs, k = w % gs;
		const int first = seg * R, last = min(n, first + R);
		int v = l[first] + (k < g[first] ? k : g[first]);
		for (int i = first; i < last; i++)
			v = t[i * gs + (v - l[i])];
		segtab[w] = v;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int s = dev->in_deemph_avg;
		for (int seg = 0; seg < nseg; seg++) {
			segstart[seg] = s;
			s = segtab[seg * gs + (s - l[seg * R])];
		}
		dev->out_deemph_avg = s;      // state after the last sample of the run
	}
	__syncthreads();
	for (int seg = threadIdx.x; seg < nseg; seg += blockDim.x) {
		const int first = seg * R, last = min(n, first + R);
		int s = segstart[seg];
		for (int i = first; i < last; i++) {
			start[i] = s;
			s = t[i * gs + (s - l[i])];
		}
	}
}

// start states one level down: thread per parent walks its DEEMPH_FAN children
__global__ void k_fm_deemph_down(u64 n_child, int gs, const int *__restrict__ tab, const int *__restrict__ lo,
                                 const int *__restrict__ p_start, int *__restrict__ start)
{
	const u64 parent = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 first = parent * DEEMPH_FAN;
	if (first >= n_child)
		return;
	const int cnt = (int)((n_child - first) < DEEMPH_FAN ? (n_child - first) : DEEMPH_FAN);
	int s = p_start[parent];
	for (int i = 0; i < cnt; i++) {
		start[first + i] = s;
		s = tab[(first + i) * gs + (s - lo[first + i])];
	}
}

// every chunk replayed once from its exact start state: staged like the scan, filtered in place in LDS by
// one lane per chunk, written back coalesced
template <bool EVEN, bool D24>
__global__ __launch_bounds__(64) void k_fm_deemph_apply(
	const int16_t *__restrict__ pcm, u64 M, int a, unsigned magic, int bias, int chunk_l2, int gs,
	const int *__restrict__ pre, const int *__restrict__ p_lo, const int *__restrict__ p_start, int16_t *__restrict__ y)
{
	extern __shared__ uint4 de_lds[];
	const int lane = threadIdx.x, chunk = 1 << chunk_l2, row_u = chunk / 8 + 1;
	const u64 n_chunks = (M + chunk - 1) >> chunk_l2;
	const u64 first = (u64)blockIdx.x * 64;
	const int cand = p_start[blockIdx.x] - p_lo[blockIdx.x];         // which candidate of the workgroup's first chunk was the true state
	deemph_stage(pcm, M, chunk_l2, 0, first, de_lds, lane);
	__syncthreads();
	const u64 c = first + lane;
	if (c < n_chunks) {
		const u64 c0 = c << chunk_l2;
		const int n = (int)((M - c0) < (u64)chunk ? (M - c0) : (u64)chunk);
		const int xoff = a / 2 + bias * a;
		int s = pre[c * gs + cand];
		uint4 *row = de_lds + (lane + 1) * row_u;
		if (!EVEN && D24) {
			int N = de_state(s, a / 2);
			const uint32_t hh = (uint32_t)(a / 2) * 0x00010001u;
			for (int u = 0; u * 8 < n; u++) {           // past-the-end samples of the last unit are never stored
				const uint4 w = row[u];
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				uint32_t o[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					de_step<0>(ww[k], N, magic, -64);
					const int z0 = N >> 6;                 // a/2 - avg
					de_step<1>(ww[k], N, magic, -64);
					const int z1 = N >> 6;
					o[k] = pk_sub(hh, __builtin_amdgcn_perm((uint32_t)z1, (uint32_t)z0, 0x05040100u));
				}
				row[u] = make_uint4(o[0], o[1], o[2], o[3]);
			}
		} else {
			for (int u = 0; u * 8 < n; u++) {
				const uint4 w = row[u];
				const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
				uint32_t o[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int x0 = lo16(ww[k]), x1 = hi16(ww[k]);
					s = deemph_step_d<EVEN, D24>(s, x0 + xoff, x0, magic, bias);
					const int y0 = s;
					s = deemph_step_d<EVEN, D24>(s, x1 + xoff, x1, magic, bias);
					o[k] = pack_iq(y0, s);
				}
				row[u] = make_uint4(o[0], o[1], o[2], o[3]);
			}
		}
	}
	__syncthreads();
	const u64 p0 = first << chunk_l2;
	u64 p1 = (first + 64) << chunk_l2;
	if (p1 > M)
		p1 = M;
	const unsigned nfull = (unsigned)((p1 - p0) >> 3), ul2 = chunk_l2 - 3;
	uint4 *dst = reinterpret_cast<uint4 *>(y + p0);
	for (unsigned u = lane; u < nfull; u += 64)
		dst[u] = de_lds[((u >> ul2) + 1) * row_u + (u & ((1u << ul2) - 1))];
	const unsigned tail = (unsigned)((p1 - p0) & 7);
	if (tail && lane == 0) {
		const uint4 w = de_lds[((nfull >> ul2) + 1) * row_u + (nfull & ((1u << ul2) - 1))];
		const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
		for (unsigned k = 0; k < tail; k++)
			y[p0 + ((u64)nfull << 3) + k] = (int16_t)((k & 1) ? hi16(ww[k >> 1]) : lo16(ww[k >> 1]));
	}
}

__global__ void k_fm_deemph_serial(const int16_t *__restrict__ pcm, u64 M, int a, int16_t *__restrict__ y,
                                   rxk_fm_dev *__restrict__ dev)
{
	if (threadIdx.x || blockIdx.x)
		return;
	int s = dev->in_deemph_avg;
	const int h = a / 2;
	for (u64 i = 0; i < M; i++) {
		const int d = pcm[i] - s;
		s += d > 0 ? (d + h) / a : (d - h) / a;
		y[i] = (int16_t)s;
	}
	dev->out_deemph_avg = s;
}


// ------------------------------------------------------------------ F8 + F9 on the tiled stream (the small-decimation chain)

// With -M wbfm's default decimation (ds = 6; ds = 5 for BASELINE configs[0]) the audio stages see a sixth of the capture
// rate and the LDS-staged kernels above -- one 64-lane workgroup per 16 KiB of staging, two waves per SIMD, loads, compute
// and stores one after the other -- are latency-bound at a fraction of the VALU rate.  Here the decimator hands the
// demodulated samples over in the tiled layout (pcm_index), every lane streams its chunk with coalesced 16-byte loads
// straight into registers, there is no LDS staging and no barrier, eight waves per SIMD:
//   k_fm_deemph_scan_t     per chunk: warm-up on the previous chunk's tail (the lowest trajectory + a bound), then the lowest
//                          candidate + merge mask (deemph_track's idea, three-instruction step) -> a COMPACT chunk table
//                          {lo_start, lo_end | gap << 16, mask} of 16 bytes
//   k_fm_deemph_up0/down0  the tree's first level on compact tables (16 chunks per parent); the levels above are the
//                          kernels of the LDS-staged path (k_fm_deemph_up / _top / _down)
//   k_fm_deemph_apply_rs_t every chunk replayed from its exact start state, and low_pass_real (rtl_fm.c:389-409) run
//                          INLINE on the filtered samples: de-emphasised audio never goes to HBM.  A lane owns the
//                          resampler windows that START in its chunk and walks on into the next chunk to finish the last
//                          one; outputs are staged per wave in a few KiB of LDS and leave coalesced.
// Odd a in 9..255 only (the three-instruction step), 2 <= rate_out / rate_out2 <= 32; everything else keeps the other path.

__device__ __forceinline__ const uint4 *tile_unit(const int16_t *pcm_t, u64 chunk, int chl2)
{
	return reinterpret_cast<const uint4 *>(pcm_t + ((chunk >> 6) << (6 + chl2))) + (chunk & 63);
}

__device__ __forceinline__ int ctab_apply(const uint4 t, int v)
{
	const int idx = v - (int)t.x;                                  // 0 .. gap
	const u64 mask = (u64)t.z | ((u64)t.w << 32);
	const u64 below = idx >= 64 ? ~0ull : (((u64)1 << idx) - 1);
	return (int)(short)(t.y & 0xffffu) + __popcll(mask & below);
}

template <int GS, int CHL2>
__global__ __launch_bounds__(256) void k_fm_deemph_scan_t(
	const int16_t *__restrict__ pcm_t, u64 M, int a, unsigned magic, int warm, int lo0, int gap_w,
	uint4 *__restrict__ ctab, rxk_fm_dev *__restrict__ dev)
{
	constexpr int CH = 1 << CHL2, UPC = CH / 8;
	const u64 c = ((u64)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
	const u64 n_chunks = (M + CH - 1) >> CHL2;
	if (c >= n_chunks)
		return;
	const u64 c0 = c << CHL2;
	const int n = (int)((M - c0) < (u64)CH ? (M - c0) : (u64)CH);
	const int h = a / 2;
	const uint4 *row = tile_unit(pcm_t, c, CHL2);
	// four units (64 bytes per lane, 4 KiB per wave) in flight: the walk is a dependent chain, the loads must not be
	uint4 cur[4];
#pragma unroll
	for (int j = 0; j < 4; j++)
		cur[j] = row[(size_t)j * 64];                         // UPC >= 16
	// Warm-up on the previous chunk's tail.  Only the LOWEST trajectory is walked: for two states g apart one step leaves them
	// at most g - floor(g / a) apart (floor(u - v) <= floor(u) - floor(v)), a bound that does not depend on the samples and is
	// non-decreasing in g -- so after `warm` steps from the whole state range the true state lies in [lo, lo + gap_w], gap_w < a
	// being the host's iterate of that recurrence (rxgpu_fm.c, deemph_geometry).  Candidates the upper trajectory would have
	// excluded are carried as candidates like any other; the merge tracking below thins them out.
	int lo, gap;
	if (c == 0) {                                             // the run's carried state
		lo = dev->in_deemph_avg;
		gap = 0;
	} else {
		const uint4 *prow = tile_unit(pcm_t, c - 1, CHL2);
		int nl = de_state(lo0, h);
		const int u0 = (CH - warm) >> 3;
		uint4 wq[4];
#pragma unroll
		for (int j = 0; j < 4; j++)
			wq[j] = prow[(size_t)min(u0 + j, UPC - 1) * 64];
		for (int ub = u0; ub < UPC; ub += 4) {
			uint4 nx[4];
#pragma unroll
			for (int j = 0; j < 4; j++)
				nx[j] = prow[(size_t)min(ub + 4 + j, UPC - 1) * 64];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (ub + j < UPC) {
					const uint32_t ww[4] = {wq[j].x, wq[j].y, wq[j].z, wq[j].w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						de_step<0>(ww[k], nl, magic, -64);
						de_step<1>(ww[k], nl, magic, -64);
					}
				}
			}
#pragma unroll
			for (int j = 0; j < 4; j++)
				wq[j] = nx[j];
		}
		lo = de_avg(nl, h);
		gap = gap_w;
	}
	if (gap >= GS) {                                          // excluded by the host (gap_w < a <= GS); checked anyway
		atomicExch(&dev->err, 1);
		gap = GS - 1;
	}
	typedef typename deemph_mask<GS>::type MASK;
	const int lo_start = lo;
	MASK mask = (MASK)(((MASK)1 << gap) - 1);
	int N = de_state(lo, h), cm6 = gap << 6;                  // r6 < cm6  <=>  remainder + 1 < number of distinct candidates
	const int na6 = -(a << 6);
	const int nu = n >> 3;
	for (int ub = 0; ub < nu; ub += 4) {
		uint4 nx[4];
#pragma unroll
		for (int j = 0; j < 4; j++)
			nx[j] = row[(size_t)min(ub + 4 + j, UPC - 1) * 64];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			if (ub + j < nu) {
				const uint32_t ww[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
				for (int k = 0; k < 4; k++) {
					int r6 = de_step_r<0>(ww[k], N, magic, -64, na6);
					if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
					r6 = de_step_r<1>(ww[k], N, magic, -64, na6);
					if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 4; j++)
			cur[j] = nx[j];
	}
	if (n & 7) {                                              // the ragged end of the run: unit nu
		const uint4 w = row[(size_t)nu * 64];
		const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
		for (int k = 0; k < (n & 7); k++) {
			const int r6 = (k & 1) ? de_step_r<1>(ww[k >> 1], N, magic, -64, na6) : de_step_r<0>(ww[k >> 1], N, magic, -64, na6);
			if (r6 < cm6) { de_merge(mask, r6 >> 6); cm6 -= 64; }
		}
	}
	const int lo_end = de_avg(N, h);
	const u64 m64 = (u64)mask;
	ctab[c] = make_uint4((uint32_t)lo_start, ((uint32_t)lo_end & 0xffffu) | ((uint32_t)gap << 16), (uint32_t)m64, (uint32_t)(m64 >> 32));
}

// The same scan for 128-sample chunks with every byte of the stream read ONCE.  k_fm_deemph_scan_t reads a chunk's tail twice:
// as the right-hand neighbour's warm-up and as the chunk's own samples, several microseconds apart -- at the -M wbfm rates that is
// 0.6 GB more per 8 GiB of capture, and the kernel runs at memory speed.  Here a lane loads its chunk's 16 units into registers
// first, and the warm-up of lane L walks the registers of lane L - 1 (one wave_shr:1 DPP move per dword).  A wave therefore
// covers 63 chunks: its lane 0 only supplies the chunk in front of them (1.6 % of the stream is loaded by two waves).
template <int GS, int CHL2 = 7>
__global__ __launch_bounds__(256) void k_fm_deemph_scan_r(
	const int16_t *__restrict__ pcm_t, u64 M, int a, unsigned magic, int warm, int lo0, int gap_w,
	uint4 *__restrict__ ctab, rxk_fm_dev *__restrict__ dev)
{
	constexpr int CH = 1 << CHL2, UPC = CH / 8;                // CHL2 = 8 (round 4): 256-sample chunks, 128 VGPRs of samples per lane
	const int lane = threadIdx.x & 63;
	const u64 wv = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
	const u64 n_chunks = (M + CH - 1) >> CHL2;
	if (wv * 63 >= n_chunks)
		return;
	const i64 cs = (i64)(wv * 63) + lane - 1;                 // lane 0: the chunk in front of the wave's 63
	const bool valid = lane >= 1 && (u64)cs < n_chunks;
	const u64 c = cs < 0 ? 0 : ((u64)cs < n_chunks ? (u64)cs : n_chunks - 1);      // lanes without a chunk load a neighbour's (unused)
	const u64 c0 = c << CHL2;
	const int n = (int)((M - c0) < (u64)CH ? (M - c0) : (u64)CH);
	const int h = a / 2;
	const uint4 *row = tile_unit(pcm_t, c, CHL2);
	uint4 own[UPC];
#pragma unroll
	for (int u = 0; u < UPC; u++)
		own[u] = row[(size_t)u * 64];
	// warm-up over the previous chunk's tail: lane L - 1's registers (see k_fm_deemph_scan_t for the one-trajectory argument)
	int nl = de_state(lo0, h);
	const int u0 = (CH - warm) >> 3;
#pragma unroll
	for (int u = 0; u < UPC; u++) {
		if (u >= u0) {
			const uint32_t ww[4] = {
				(uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[u].x, 0x138, 0xf, 0xf, false),
				(uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[u].y, 0x138, 0xf, 0xf, false),
				(uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[u].z, 0x138, 0xf, 0xf, false),
				(uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[u].w, 0x138, 0xf, 0xf, false)};
#pragma unroll
			for (int k = 0; k < 4; k++) {
				de_step<0>(ww[k], nl, magic, -64);
				de_step<1>(ww[k], nl, magic, -64);
			}
		}
	}
	int lo = de_avg(nl, h), gap = gap_w;
	if (cs <= 0) {                                            // the run's first chunk: the carried state, no candidates
		lo = dev->in_deemph_avg;
		gap = 0;
	}
	if (gap >= GS) {                                          // excluded by the host (gap_w < a <= GS); checked anyway
		atomicExch(&dev->err, 1);
		gap = GS - 1;
	}
	typedef typename deemph_mask<GS>::type MASK;
	const int lo_start = lo;
	MASK mask = (MASK)(((MASK)1 << gap) - 1);
	int N = de_state(lo, h), cm6 = gap << 6;                  // r6 < cm6  <=>  remainder + 1 < number of distinct candidates
	const int na6 = -(a << 6);
	const int nu = n >> 3;
#pragma unroll
	for (int u = 0; u < UPC; u++) {
		const uint32_t ww[4] = {own[u].x, own[u].y, own[u].z, own[u].w};
		if (u < nu) {
#pragma unroll
			for (int k = 0; k < 4; k++) {
				int r6 = de_step_r<0>(ww[k], N, magic, -64, na6);
				if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
				r6 = de_step_r<1>(ww[k], N, magic, -64, na6);
				if (__builtin_expect(r6 < cm6, 0)) { de_merge(mask, r6 >> 6); cm6 -= 64; }
			}
		} else if (u == nu && (n & 7)) {                      // the ragged end of the run
			for (int k = 0; k < (n & 7); k++) {
				const int r6 = (k & 1) ? de_step_r<1>(ww[k >> 1], N, magic, -64, na6) : de_step_r<0>(ww[k >> 1], N, magic, -64, na6);
				if (r6 < cm6) { de_merge(mask, r6 >> 6); cm6 -= 64; }
			}
		}
	}
	if (!valid)
		return;
	const int lo_end = de_avg(N, h);
	const u64 m64 = (u64)mask;
	ctab[c] = make_uint4((uint32_t)lo_start, ((uint32_t)lo_end & 0xffffu) | ((uint32_t)gap << 16), (uint32_t)m64, (uint32_t)(m64 >> 32));
}

// first tree level on compact tables: thread (parent, k) walks the parent's 16 chunk tables
__global__ void k_fm_deemph_up0(u64 n_chunks, int gs, const uint4 *__restrict__ ctab, int *__restrict__ p_tab, int *__restrict__ p_lo,
                                int *__restrict__ p_gap)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 parent = gid / gs;
	const int k = (int)(gid % gs);
	const u64 first = parent * DEEMPH_FAN;
	if (first >= n_chunks)
		return;
	const int cnt = (int)((n_chunks - first) < DEEMPH_FAN ? (n_chunks - first) : DEEMPH_FAN);
	const uint4 t0 = ctab[first];
	const int g0 = (int)(t0.y >> 16);
	int v = (int)t0.x + (k < g0 ? k : g0);
	for (int i = 0; i < cnt; i++)
		v = ctab_apply(ctab[first + i], v);
	p_tab[parent * gs + k] = v;
	if (k == 0) { p_lo[parent] = (int)t0.x; p_gap[parent] = g0; }
}

// ... and back down: the exact start state of every chunk
__global__ void k_fm_deemph_down0(u64 n_chunks, const uint4 *__restrict__ ctab, const int *__restrict__ p_start, int *__restrict__ start)
{
	const u64 parent = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 first = parent * DEEMPH_FAN;
	if (first >= n_chunks)
		return;
	const int cnt = (int)((n_chunks - first) < DEEMPH_FAN ? (n_chunks - first) : DEEMPH_FAN);
	uint4 t[DEEMPH_FAN];                                    // all of the parent's tables on their way before the (dependent) walk starts
#pragma unroll
	for (int i = 0; i < DEEMPH_FAN; i++)
		t[i] = ctab[first + (i < cnt ? i : cnt - 1)];
	int s = p_start[parent];
#pragma unroll
	for (int i = 0; i < DEEMPH_FAN; i++)
		if (i < cnt) {
			start[first + i] = s;
			s = ctab_apply(t[i], s);
		}
}

// floor(num / den) for num < 2^52 (declared with the resampler below)
__device__ __forceinline__ u64 div_floor(u64 num, u64 den);

// One sample of low_pass_real run inline behind the de-emphasis step, five instructions and a masked LDS store.  The window
// sum is kept as acc = -64 * sum: the de-emphasis state N = ((a/2 - avg) << 6) + 32 joins it as N - (64 * (a/2) + 32) = -64 * avg
// in one v_add3.  p is the resampler phase BEFORE the sample (rtl_fm.c:397-399): the sample completes a window iff
// p + slow >= fast, i.e. p >= thr = fast - slow, and then the phase moves by slow - fast instead of slow.  A completed window
// leaves as its raw int32 sum; the division by the ratio happens once per OUTPUT when the wave's staging is written out.
// Branch-free bookkeeping: lanes are at different phases, some lane of the wave emits at nearly every sample.  A lane that
// starts in the middle of somebody else's window emits that window's (partial, wrong) sum into the slot BEFORE its own first
// one: slot -1 of the staging for a wave's first lane, otherwise the slot its left neighbour fills in afterwards, when it
// walks on past its chunk to finish the window it owns -- same wave, later in program order, LDS operations retire in order.
typedef __attribute__((address_space(3))) int lds_int;         // a pointer the compiler KNOWS to be LDS: ds_write with a 32-bit address
__device__ __forceinline__ void lpr_step(int N, int nK, int &acc, int &p, lds_int *&slot, int thr, int slow, int d_emit)
{
	acc = acc + N + nK;
	if (p >= thr) {                              // everything an emission changes sits in the one exec-masked region of the store
		*slot++ = acc;                           // slot IS the LDS address of the lane's next output
		acc = 0;
		p += d_emit - slow;                      // = -fast; the unconditional step below completes it
	}
	p += slow;
}

template <int CHL2>
__global__ __launch_bounds__(256) void k_fm_deemph_apply_rs_t(
	const int16_t *__restrict__ pcm_t, u64 M, int a, unsigned magic, const int *__restrict__ start,
	int fast, int slow, float rinv, int wcap, int16_t *__restrict__ out, rxk_fm_dev *__restrict__ dev)
{
	constexpr int CH = 1 << CHL2, UPC = CH / 8;
	extern __shared__ int stage_all[];                         // per wave: wcap window sums (as -64 * sum)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int *stage = stage_all + (size_t)wave * wcap;
	const u64 cw = ((u64)blockIdx.x * (blockDim.x >> 6) + wave) * 64;   // the wave's first chunk (1, 2 or 4 waves per workgroup)
	const u64 c = cw + lane;
	const u64 n_chunks = (M + CH - 1) >> CHL2;
	if (cw >= n_chunks)
		return;
	const bool valid = c < n_chunks;
	const int h = a / 2;
	const u64 p_in = (u64)dev->in_prev_lpr_index;
	// outputs completed before this chunk, and the resampler phase at its first sample (rtl_fm.c:397-399)
	const u64 c0 = valid ? c << CHL2 : M;
	const u64 num = p_in + c0 * (u64)slow;
	const u64 e0 = div_floor(num, (u64)fast);
	int p = (int)(num - e0 * (u64)fast);
	const int p_first = p;
	int acc = (c == 0) ? -64 * dev->in_now_lpr : 0;           // the run's first lane continues the carried window (as -64 * sum)
	const int nK = -(64 * h + 32), thr = fast - slow, d_emit = slow - fast;
	// does a window start exactly at this chunk's first sample (or is this the run's first lane)?  If not, the window under way
	// belongs to the lane on the left and completes after `skip` samples; this lane's own outputs start behind it.
	const bool owns_first = c == 0 || p < slow;
	const u64 j_first = e0 + (owns_first ? 0 : 1);
	const u64 jw0 = (u64)__builtin_amdgcn_readfirstlane((int)(unsigned)j_first) | ((u64)__builtin_amdgcn_readfirstlane((int)(unsigned)(j_first >> 32)) << 32);
	stage += 1;
	lds_int *const stage_l = (lds_int *)stage;
	lds_int *slot = stage_l + ((int)(j_first - jw0) - (owns_first ? 0 : 1));   // >= stage - 1: the staging has one spare element in front
	if (valid) {
		const int n = (int)((M - c0) < (u64)CH ? (M - c0) : (u64)CH);
		int N = de_state(start[c], h);
		const uint4 *row = tile_unit(pcm_t, c, CHL2);
		const int nu = n >> 3;
		// FOUR units on their way per lane (round 4; one before): eight samples are ~70 dependent instructions, a load under the next run's
		// decimator takes several times that, and the waves that could cover it are exactly what the two kernels compete for.  The chunk's
		// units all lie inside the tile (loads past a short last chunk's end read allocated, unused samples).
		uint4 q[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
			q[k] = row[(size_t)k * 64];
		uint4 w = q[0];
		for (int u0 = 0; u0 < nu; u0 += 4) {
#pragma unroll
			for (int k4 = 0; k4 < 4; k4++) {
				const int u = u0 + k4;
				w = q[k4];
				if (u + 4 < UPC)
					q[k4] = row[(size_t)(u + 4) * 64];
				if (u < nu) {
					const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						de_step<0>(ww[k], N, magic, -64);
						lpr_step(N, nK, acc, p, slot, thr, slow, d_emit);
						de_step<1>(ww[k], N, magic, -64);
						lpr_step(N, nK, acc, p, slot, thr, slow, d_emit);
					}
				}
			}
		}
		if (n & 7) {
			// the ragged end of the run (its last chunk only): unit nu once more, plainly
			w = row[(size_t)nu * 64];
			const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
			for (int k = 0; k < (n & 7); k++) {
				if (k & 1) de_step<1>(ww[k >> 1], N, magic, -64); else de_step<0>(ww[k >> 1], N, magic, -64);
				lpr_step(N, nK, acc, p, slot, thr, slow, d_emit);
			}
		}
		const u64 end = c0 + (u64)n;
		// a window this lane owns has started iff the one it found under way (if any) completed inside the chunk
		const int skip = owns_first ? 0 : (fast - p_first + slow - 1) / slow;
		const bool own = skip < n || (owns_first && n > 0);
		if (end == M) {
			// the run's last chunk: what is left is the carry (rtl_fm.c:150-151).  A window still open belongs either to
			// this lane (own) or to an earlier lane that walks on to M below and writes it; no window open: zero.
			if (own)
				dev->out_now_lpr = -(acc >> 6);
			else if (p < slow)
				dev->out_now_lpr = 0;
			dev->out_prev_lpr_index = p;
		} else if (own && p >= slow) {
			// a window this lane owns is still open: walk on into the following chunk(s) until it completes
			u64 i = end;
			bool open = true;
			while (open && i < M) {
				const uint4 wv = tile_unit(pcm_t, i >> CHL2, CHL2)[(size_t)((unsigned)(i & (CH - 1)) >> 3) * 64];
				const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
				for (int k = (int)(i & 7); k < 8 && open && i < M; k++, i++) {
					if (k & 1) de_step<1>(ww[k >> 1], N, magic, -64); else de_step<0>(ww[k >> 1], N, magic, -64);
					const lds_int *before = slot;
					lpr_step(N, nK, acc, p, slot, thr, slow, d_emit);
					open = slot == before;
				}
			}
			if (open)
				dev->out_now_lpr = -(acc >> 6);               // ran into the end of the run: this partial window is the carry
		}
	}
	// the wave's outputs [jw0, jw0 + cnt) leave coalesced; lanes own ascending, contiguous ranges
	int cnt = valid ? (int)(slot - stage_l) : 0;
	for (int off = 32; off; off >>= 1)
		cnt = max(cnt, __shfl_xor(cnt, off));
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	for (int idx = lane; idx < cnt; idx += 64)                // (int16)(sum / ratio), the truncating division by the reciprocal rounded up
		out[jw0 + (u64)idx] = (int16_t)(int)((float)(-(stage[idx] >> 6)) * rinv);
}

// ------------------------------------------------------------------ F9 low_pass_real

// rtl_fm.c:389-409 in closed form.  With phase p0 < fast and slow <= fast, after i inputs
// floor((p0 + i*slow)/fast) outputs exist, so output j sums inputs [E(j-1), E(j)) with
// E(j) = ceil(((j+1)*fast - p0) / slow), E(-1) = 0, and is (int16)(sum / (fast/slow)).
// floor(num / den) for num < 2^52 through one fp64 division and an exact correction
__device__ __forceinline__ u64 div_floor(u64 num, u64 den)
{
	u64 q = (u64)((double)num / (double)den);
	const i64 r = (i64)(num - q * den);
	if (r < 0) q--;
	else if ((u64)r >= den) q++;
	return q;
}
__device__ __forceinline__ u64 lpr_end(u64 j, u64 fast, u64 slow, u64 p0)
{
	return div_floor((j + 1) * fast - p0 + slow - 1, slow);
}

__global__ void k_fm_resample(const int16_t *__restrict__ y, u64 n, int fast, int slow, u64 J,
                              int16_t *__restrict__ out, rxk_fm_dev *__restrict__ dev)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 p0 = (u64)dev->in_prev_lpr_index;
	const int ratio = fast / slow;
	if (j < J) {
		// E(j) = floor(((j+1)*fast - p0 + slow - 1) / slow).  One 64-bit division per wave (its first output), then
		// E(j0 + l) = q0 + floor((r0 + (l+1)*fast) / slow) on 32-bit numerators through exact fp64 divisions
		u64 b, e;
		if (fast < (1 << 25)) {
			const u64 j0 = j - (threadIdx.x & 63);
			const i64 nb = (i64)(j0 * (u64)fast) - (i64)p0 + (i64)slow - 1;         // < 0 only for j0 == 0
			i64 q0 = nb >= 0 ? (i64)div_floor((u64)nb, (u64)slow) : -(i64)div_floor((u64)(-nb) + (u64)slow - 1, (u64)slow);
			const unsigned r0 = (unsigned)(nb - q0 * (i64)slow);
			const unsigned l = threadIdx.x & 63;
			const unsigned xb = r0 + l * (unsigned)fast, xe = xb + (unsigned)fast;
			const i64 bb = q0 + (i64)(unsigned)((double)xb / (double)slow);
			b = bb > 0 ? (u64)bb : 0;                                                // E(-1) = 0
			e = (u64)(q0 + (i64)(unsigned)((double)xe / (double)slow));
		} else {
			b = j ? lpr_end(j - 1, fast, slow, p0) : 0;
			e = lpr_end(j, fast, slow, p0);
		}
		int sum = j ? 0 : dev->in_now_lpr;
		for (u64 i = b; i < e; i += 8) {               // eight independent loads in flight, not one after the other
			int v[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
				v[k] = i + k < e ? y[i + k] : 0;
#pragma unroll
			for (int k = 0; k < 8; k++)
				sum += v[k];
		}
		// C's truncating division through fp64 (exact for |sum| < 2^31, see fast_atan2_dev)
		out[j] = (int16_t)(int)((double)sum / (double)ratio);
	}
	if (j == J) {                                 // one extra thread: the carries
		const u64 b = J ? lpr_end(J - 1, fast, slow, p0) : 0;
		int sum = J ? 0 : dev->in_now_lpr;
		for (u64 i = b; i < n; i++)
			sum += y[i];
		dev->out_now_lpr = sum;
		dev->out_prev_lpr_index = (int)(p0 + n * (u64)slow - J * (u64)fast);
	}
}

// pipelined runs: what the previous run left as carries-out is this run's carries-in; the audio-stage carries-in are
// also kept aside so that those stages can be redone after a host fix-up of a libm sample
__global__ void k_fm_carry_advance(rxk_fm_dev *dev, int advance, int *snap)
{
	if (advance) {
		dev->in_now_r = dev->out_now_r; dev->in_now_j = dev->out_now_j; dev->in_prev_index = dev->out_prev_index;
		dev->in_pre_r = dev->out_pre_r; dev->in_pre_j = dev->out_pre_j;
		dev->in_deemph_avg = dev->out_deemph_avg;
		dev->in_now_lpr = dev->out_now_lpr; dev->in_prev_lpr_index = dev->out_prev_lpr_index;
		dev->in_dc_avg = dev->out_dc_avg;
	}
	snap[0] = dev->in_deemph_avg; snap[1] = dev->in_now_lpr; snap[2] = dev->in_prev_lpr_index; snap[3] = dev->in_dc_avg;
}

__global__ void k_fm_audio_carry(rxk_fm_dev *dev, const int *snap)
{
	if (snap) {
		dev->in_deemph_avg = snap[0]; dev->in_now_lpr = snap[1]; dev->in_prev_lpr_index = snap[2]; dev->in_dc_avg = snap[3];
	} else {
		dev->in_deemph_avg = dev->out_deemph_avg;
		dev->in_now_lpr = dev->out_now_lpr; dev->in_prev_lpr_index = dev->out_prev_lpr_index;
		dev->in_dc_avg = dev->out_dc_avg;
	}
}

__global__ void k_fm_passthrough_carry(rxk_fm_dev *dev, int deemph_off, int resample_off)
{
	if (deemph_off)
		dev->out_deemph_avg = dev->in_deemph_avg;
	if (resample_off) {
		dev->out_now_lpr = dev->in_now_lpr;
		dev->out_prev_lpr_index = dev->in_prev_lpr_index;
	}
}

// ------------------------------------------------------------------ F3 fifth_order cascade

// One pass of rtl_fm.c:411-440 over every block at once.  On the strided sequence s_k of one
// component, output k of a block is
//   (s[2k-5] + 5(s[2k-4] + s[2k-1]) + 10(s[2k-3] + s[2k-2]) + s[2k]) >> 4      (int, then int16)
// where negative indices reach into the previous block's SAME-pass input: its samples
// 2K'-4 .. 2K' with K' = ceil(n/2)-1 the index of its last output (for even n the block's last
// sample is never consumed, rtl_fm.c:424-432).  Block 0 of a run takes them from the carried
// lp_*_hist[pass][1..5].
template <bool RAW, bool PRESCALED, bool ROTATE>
__device__ __forceinline__ void fifth_tap(const void *in, u64 blk, unsigned idx, unsigned in_stride, int &ri, int &rq)
{
	if (RAW) {
		const uint32_t *p = (const uint32_t *)in + blk * (u64)in_stride;
		const uint32_t w = p[idx];
		int i = lo16(w), q = hi16(w);
		if (!PRESCALED) { i = scale_cs16(i); q = scale_cs16(q); }
		if (ROTATE) {
			switch (idx & 3) {
			case 0: ri = i; rq = q; break;
			case 1: ri = -q; rq = i; break;
			case 2: ri = -i; rq = -q; break;
			default: ri = q; rq = -i; break;
			}
			// the reference stores the rotated value back into int16 (rtl_fm.c:316-325)
			ri = (int)(short)ri; rq = (int)(short)rq;
		} else {
			ri = i; rq = q;
		}
	} else {
		const uint32_t w = ((const uint32_t *)in)[blk * (u64)in_stride + idx];
		ri = lo16(w); rq = hi16(w);
	}
}

template <bool RAW, bool PRESCALED, bool ROTATE>
__global__ __launch_bounds__(256) void k_fm_fifth_pass(
	const void *__restrict__ in, u64 n_blocks, unsigned n_in, unsigned in_stride, uint32_t *__restrict__ out,
	unsigned out_stride, const int16_t *__restrict__ hist_in, int16_t *__restrict__ hist_out)
{
	const unsigned n_out = (n_in + 1) / 2;
	const unsigned k = blockIdx.x * 256 + threadIdx.x;        // grid: x over a block's outputs, y over the callback blocks
	if (k >= n_out)
		return;
	const unsigned kl = n_out - 1;                 // K' of any (equal-length) block
	for (u64 blk = blockIdx.y; blk < n_blocks; blk += gridDim.y) {
	int ti[6], tq[6];
#pragma unroll
	for (int t = 0; t < 6; t++) {
		const int idx = (int)(2 * k) - 5 + t;
		if (idx >= 0) {
			fifth_tap<RAW, PRESCALED, ROTATE>(in, blk, (unsigned)idx, in_stride, ti[t], tq[t]);
		} else if (blk == 0) {
			// carried history: s[-5..-1] = hist[1..5]
			ti[t] = hist_in[6 + idx];
			tq[t] = hist_in[6 + 6 + idx];
		} else {
			// previous block, same pass: s[-1] = its sample 2K', s[-5] = 2K'-4
			const int pidx = (int)(2 * kl) + 1 + idx;  // idx in -5..-1 -> 2K'-4 .. 2K'
			if (pidx >= 0) {
				fifth_tap<RAW, PRESCALED, ROTATE>(in, blk - 1, (unsigned)pidx, in_stride, ti[t], tq[t]);
			} else {
				ti[t] = 0; tq[t] = 0;              // blocks shorter than the filter: rejected by the host
			}
		}
	}
	const int oi = (ti[0] + (ti[1] + ti[4]) * 5 + (ti[2] + ti[3]) * 10 + ti[5]) >> 4;
	const int oq = (tq[0] + (tq[1] + tq[4]) * 5 + (tq[2] + tq[3]) * 10 + tq[5]) >> 4;
	out[blk * (u64)out_stride + k] = pack_iq(oi, oq);
	if (blk == n_blocks - 1 && k == kl) {
		// archive, rtl_fm.c:434-439: the last window
#pragma unroll
		for (int t = 0; t < 6; t++) {
			hist_out[t] = (int16_t)ti[t];
			hist_out[6 + t] = (int16_t)tq[t];
		}
	}
	}
}

// ------------------------------------------------------------------ F3 fused: first 1-3 passes in LDS

// For the raw cs16 stream the callback's scale bounds every sample by 128, each fifth_order pass
// has gain 2, so through three passes every tap sum stays below 2^15: the int arithmetic of
// rtl_fm.c:423/431 can be done on packed int16 pairs (I and Q at once) without changing a bit.
// A workgroup turns FF_RAW raw samples (+36 of left halo) into FF_RAW>>FUSE samples, keeping the
// intermediate levels in LDS.  Level-p sample i of block b is written V_p(b,i); negative i reach
// into the previous block through the seam rule (s[-q] = its sample n_p-1-q) -- those five values
// per block and level come precomputed from k_fm_fifth_seams, block 0 takes the carried hist.
#define FF_RAW 2048

typedef short ff_s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t fifth_pk(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	const ff_s16x2 va = __builtin_bit_cast(ff_s16x2, a), vb = __builtin_bit_cast(ff_s16x2, b), vc = __builtin_bit_cast(ff_s16x2, c);
	const ff_s16x2 vd = __builtin_bit_cast(ff_s16x2, d), ve = __builtin_bit_cast(ff_s16x2, e), vf = __builtin_bit_cast(ff_s16x2, f);
	const ff_s16x2 sum = (va + vf) + (vb + ve) * (ff_s16x2)(5) + (vc + vd) * (ff_s16x2)(10);
	return __builtin_bit_cast(uint32_t, sum >> (ff_s16x2)(4));
}

// the same window in the reference's int arithmetic (rtl_fm.c:423/431), for levels whose sums exceed int16.  The levels this
// is used on hold values below 2^14 in magnitude (raw samples are at most 128 after the scale, every pass doubles at most, six
// passes at most in front of it), so the three pair sums still fit packed int16; each component's 32-bit sum is then three
// v_mad_i32_i16 (op_sel picks the half), and shift + pack take three more: 12 instructions instead of ~30 of unpacking
__device__ __forceinline__ uint32_t fifth_pk32(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	const uint32_t af = pk_add(a, f), be = pk_add(b, e), cd = pk_add(c, d);
	int si, sq;
	asm("v_mad_i32_i16 %0, %2, 1, 0\n\tv_mad_i32_i16 %1, %2, 1, 0 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %3, 5, %0\n\tv_mad_i32_i16 %1, %3, 5, %1 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %4, 10, %0\n\tv_mad_i32_i16 %1, %4, 10, %1 op_sel:[1,0,0,0]"
	    : "=&v"(si), "=&v"(sq) : "v"(af), "v"(be), "v"(cd));
	// (si >> 4) & 0xffff | (sq >> 4) << 16
	return (uint32_t)__builtin_amdgcn_ubfe((unsigned)si, 4u, 16u) | (((unsigned)sq << 12) & 0xffff0000u);
}
// ... and for inputs of any magnitude (rx_power's buffers are raw int16)
__device__ __forceinline__ uint32_t fifth_int(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	// twelve v_mad_i32_i16 straight from the packed halves (op_sel picks Q), no unpacking: the 16-bit pre-additions of fifth_pk32 would wrap here
	int si, sq;
	asm("v_mad_i32_i16 %0, %2, 1, 0\n\tv_mad_i32_i16 %1, %2, 1, 0 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %7, 1, %0\n\tv_mad_i32_i16 %1, %7, 1, %1 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %3, 5, %0\n\tv_mad_i32_i16 %1, %3, 5, %1 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %6, 5, %0\n\tv_mad_i32_i16 %1, %6, 5, %1 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %4, 10, %0\n\tv_mad_i32_i16 %1, %4, 10, %1 op_sel:[1,0,0,0]\n\t"
	    "v_mad_i32_i16 %0, %5, 10, %0\n\tv_mad_i32_i16 %1, %5, 10, %1 op_sel:[1,0,0,0]"
	    : "=&v"(si), "=&v"(sq) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f));
	// (si >> 4) & 0xffff | (sq >> 4) << 16: the int16 stores of rtl_power.c:599-606
	return (uint32_t)__builtin_amdgcn_ubfe((unsigned)si, 4u, 16u) | (((unsigned)sq << 12) & 0xffff0000u);
}
// MODE 0: packed int16 (sums below 2^15), 1: 32-bit sums of values below 2^14, 2: 32-bit sums of anything
template <int MODE>
__device__ __forceinline__ uint32_t fifth_any(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	return MODE == 0 ? fifth_pk(a, b, c, d, e, f) : MODE == 1 ? fifth_pk32(a, b, c, d, e, f) : fifth_int(a, b, c, d, e, f);
}

template <bool ROTATE>
__device__ __forceinline__ uint32_t raw_scaled(uint32_t w, unsigned idx)
{
	const int i = scale_cs16(lo16(w)), q = scale_cs16(hi16(w));
	if (!ROTATE)
		return pack_iq(i, q);
	switch (idx & 3) {
	case 0: return pack_iq(i, q);
	case 1: return pack_iq(-q, i);
	case 2: return pack_iq(-i, -q);
	default: return pack_iq(q, -i);
	}
}

// the fused kernel's input element: raw cs16 (scale + rotate) in stage 1, an already packed level sample in stage 2
template <bool ROTATE, bool STAGE2>
__device__ __forceinline__ uint32_t leaf(uint32_t w, unsigned idx)
{
	return STAGE2 ? w : raw_scaled<ROTATE>(w, idx);
}

// level-P sample idx (>= 0, far enough from the block start that no tap is negative)
template <int P, bool ROTATE, bool STAGE2>
__device__ uint32_t level_val(const uint32_t *__restrict__ blk_raw, int idx)
{
	if constexpr (P == 0) {
		return leaf<ROTATE, STAGE2>(blk_raw[idx], (unsigned)idx);
	} else {
		uint32_t t[6];
#pragma unroll
		for (int k = 0; k < 6; k++)
			t[k] = level_val<P - 1, ROTATE, STAGE2>(blk_raw, 2 * idx - 5 + k);
		// raw input: level P - 1 holds values up to 128 << (P - 1), the tap sum reaches 32 times that -- 2^15 from the fifth level on
		// (P >= 4), where the packed int16 sum would wrap on a saturated capture: 32-bit sums there, like the register kernel's WIDE passes
		return fifth_any<STAGE2 ? 2 : (P >= 4 ? 1 : 0)>(t[0], t[1], t[2], t[3], t[4], t[5]);
	}
}

// seams[b][p][q] = V_p(b, -5+q), p < LV, q < 5; plus the carried-out histories of the fused passes
// LV = 3: the layout every fused kernel reads (15 dwords per block); LV = 4: 20 per block, for the four-pass register kernel
template <bool ROTATE, bool STAGE2, int LV = 3>
__global__ void k_fm_fifth_seams(const uint32_t *__restrict__ iq, u64 n_blocks, unsigned n, int fuse,
                                 const int16_t *__restrict__ hist_in, uint32_t *__restrict__ seams,
                                 int16_t *__restrict__ hist_out)
{
	constexpr int SL = LV == 3 ? 16 : 32;                  // slots (threads) per block, LV * 5 of them used (LV <= 5)
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 b = gid / SL;
	const int slot = (int)(gid % SL), p = slot / 5, q = slot % 5;
	auto level = [&](const uint32_t *blk_raw, int idx) -> uint32_t {
		if constexpr (LV >= 4) {
			if (p == 3)
				return level_val<3, ROTATE, STAGE2>(blk_raw, idx);
		}
		if constexpr (LV >= 5) {
			if (p == 4)
				return level_val<4, ROTATE, STAGE2>(blk_raw, idx);
		}
		return p == 0 ? level_val<0, ROTATE, STAGE2>(blk_raw, idx) : p == 1 ? level_val<1, ROTATE, STAGE2>(blk_raw, idx) : level_val<2, ROTATE, STAGE2>(blk_raw, idx);
	};
	if (b < n_blocks && slot < LV * 5 && p < fuse) {
		uint32_t v;
		if (b == 0) {
			// carried history: s[-5..-1] = hist[1..5] (rtl_fm.c:416-420)
			v = pack_iq(hist_in[p * 12 + 1 + q], hist_in[p * 12 + 6 + 1 + q]);
		} else {
			const uint32_t *prev = iq + (b - 1) * (u64)n;
			const int idx = (int)(n >> p) - 6 + q;          // s[-5+q] = previous block's sample n_p-1+(-5+q)
			v = level(prev, idx);
		}
		seams[(b * LV + p) * 5 + q] = v;
	}
	if (b == n_blocks && slot < LV * 5) {
		// archive (rtl_fm.c:434-439): pass p leaves its last window, samples n_p-7 .. n_p-2
		if (p < fuse) {
			const uint32_t *last = iq + (n_blocks - 1) * (u64)n;
			for (int t = q; t < 6; t += 5) {
				const int idx = (int)(n >> p) - 7 + t;
				const uint32_t v = level(last, idx);
				hist_out[p * 12 + t] = (int16_t)lo16(v);
				hist_out[p * 12 + 6 + t] = (int16_t)hi16(v);
			}
		}
	}
}

// Every level lives in LDS de-interleaved: E[k] = V[2k], O[k] = V[2k+1] (relative to the tile's base at
// that level), because output i of a pass needs V[2i-5..2i] = O[i-3], E[i-2], O[i-2], E[i-1], O[i-1], E[i]:
// consecutive outputs read consecutive words of each array, so a lane's four outputs come from two
// aligned b128 reads per array with no bank conflict (a stride-8-dword layout was 8-way conflicted).
// ev/od point at E[i-2] and O[i-3] for the quad's first output i.
template <int WIDE>
__device__ __forceinline__ uint4 fifth_quad_eo(const uint32_t *__restrict__ ev, const uint32_t *__restrict__ od)
{
	const uint4 e0 = *reinterpret_cast<const uint4 *>(ev), e1 = *reinterpret_cast<const uint4 *>(ev + 4);
	const uint4 o0 = *reinterpret_cast<const uint4 *>(od), o1 = *reinterpret_cast<const uint4 *>(od + 4);
	uint4 r;
	r.x = fifth_any<WIDE>(o0.x, e0.x, o0.y, e0.y, o0.z, e0.z);
	r.y = fifth_any<WIDE>(o0.y, e0.y, o0.z, e0.z, o0.w, e0.w);
	r.z = fifth_any<WIDE>(o0.z, e0.z, o0.w, e0.w, o1.x, e1.x);
	r.w = fifth_any<WIDE>(o0.w, e0.w, o1.x, e1.x, o1.y, e1.y);
	return r;
}

// LDS word offsets: level L keeps E_L[k] at le[k + FE_L] and O_L[k] at lo[k + FO_L]; chosen so that the quads
// (first output -16+4q, -8+4q, 4q at levels 1, 2, 3) read 16-byte aligned and the E pairs are written 8-byte aligned
#define FE0 18
#define FO0 19
#define FE1 10
#define FO1 11
#define FE2 6
#define FO2 7

// rx_power's stateless fifth_order (rtl_power.c:582-607): no history -- the first five outputs of every buffer and pass
// come from special ease-in formulas on the pass input's samples 0..8 (k_pw_fifth has them one by one)
__device__ __forceinline__ uint32_t ease_out(int k, const uint32_t *le, const uint32_t *lo, int fe, int fo)
{
	uint32_t s[9];
#pragma unroll
	for (int i = 0; i < 9; i++)
		s[i] = (i & 1) ? lo[i / 2 + fo] : le[i / 2 + fe];
	int oi, oq;
#define EASE_K(get, o) do { \
		const int a = get(s[0]), b_ = get(s[1]), c = get(s[2]), d = get(s[3]), e = get(s[4]), f = get(s[5]); \
		switch (k) { \
		case 0: o = ((a + b_) * 10 + (c + d) * 5 + d + f) >> 4; break; \
		case 1: o = ((b_ + c) * 10 + (a + d) * 5 + e + f) >> 4; break; \
		case 2: o = (a + (b_ + e) * 5 + (c + d) * 10 + f) >> 4; break; \
		case 3: o = (c + (d + f) * 5 + (e + f) * 10 + get(s[6])) >> 4; break; \
		default: o = (e + (f + get(s[7])) * 5 + (f + get(s[6])) * 10 + get(s[8])) >> 4; break; \
		} } while (0)
	EASE_K(lo16, oi);
	EASE_K(hi16, oq);
#undef EASE_K
	return pack_iq(oi, oq);
}

// outputs i .. i+3 of a buffer's first tile: the quad that holds any of outputs 0..4 takes them from the ease-in formulas
__device__ __forceinline__ void ease_fix(uint4 &o, int i, const uint32_t *le, const uint32_t *lo, int fe, int fo)
{
	if (i == 0) {
		o.x = ease_out(0, le, lo, fe, fo); o.y = ease_out(1, le, lo, fe, fo); o.z = ease_out(2, le, lo, fe, fo); o.w = ease_out(3, le, lo, fe, fo);
	} else if (i == 4) {
		o.x = ease_out(4, le, lo, fe, fo);
	}
}

// EASE: the buffers are independent and stateless (rx_power): no seams, ease-in at every buffer start; in_stride / out_stride:
// distance between buffers in samples of the input / of the group's output
template <int FUSE, bool ROTATE, bool STAGE2, bool EASE>
__global__ __launch_bounds__(256) void k_fm_fifth_fused(
	const uint32_t *__restrict__ iq, unsigned n, unsigned tiles_per_block, unsigned tpw, const uint32_t *__restrict__ seams,
	uint32_t *__restrict__ out, unsigned in_stride, unsigned out_stride)
{
	__shared__ __attribute__((aligned(16))) uint32_t le0[FF_RAW / 2 + FE0 + 14], lo0[FF_RAW / 2 + FO0 + 13];
	__shared__ __attribute__((aligned(16))) uint32_t le1[FF_RAW / 4 + FE1 + 14], lo1[FF_RAW / 4 + FO1 + 13];
	__shared__ __attribute__((aligned(16))) uint32_t le2[FF_RAW / 8 + FE2 + 10], lo2[FF_RAW / 8 + FO2 + 9];
	const unsigned wgs_per_block = tiles_per_block / tpw;
	const u64 blk = blockIdx.x / wgs_per_block;
	const unsigned tile0 = (blockIdx.x % wgs_per_block) * tpw;
	const uint32_t *braw = iq + blk * (u64)in_stride;
	const uint32_t *sm = EASE ? nullptr : seams + blk * 15;
	uint32_t *bout = out + blk * (u64)out_stride;
	const int tid = threadIdx.x;
	const scale_k K = scale_consts();
	constexpr int NV = (FF_RAW + 36) / 4;                  // 521 vectors of 4 samples: the tile + 36 of left halo
	// arithmetic of the passes: packed int16 on the raw stream, 32-bit sums behind it -- of bounded values in rx_fm's cascade
	// (passes 4-6), of anything in rx_power's buffers
	constexpr int FMODE = !STAGE2 ? 0 : (EASE ? 2 : 1);

	// a workgroup walks `tpw` consecutive tiles; the next tile's samples are in flight while this one is computed
	u32x4 w[3];
#pragma unroll
	for (int u = 0; u < 3; u++) {
		const int v4 = tid + 256 * u;
		const int rel = 4 * v4 - 36;
		const bool on = v4 < NV && !(tile0 == 0 && rel < 0);
		w[u] = on ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(braw + tile0 * FF_RAW + rel)) : (u32x4)(0u);
	}
	for (unsigned tt = 0; tt < tpw; tt++) {
		const unsigned tile = tile0 + tt, t0 = tile * FF_RAW;
		const bool first = tile == 0;
		// Tiles after the first of this workgroup's walk CONTINUE the stream: the five samples in front of a level are the
		// previous tile's last five, still in LDS -- one thread moves them to the front (the thread, or a lane of the wave, that
		// overwrites the tail later in program order), nothing of the halo is loaded, converted or filtered again, and every
		// pass is whole rounds of the workgroup: 512 vectors, 256 / 128 / 64 quads.
		const bool cont = tt > 0;
		// level 0: scale + rotate (t0 + rel is a multiple of 4: phases 0..3), de-interleaved into LDS.
		// vector v4 holds V0[rel..rel+3], rel = 4 v4 - 36: E_0[2 v4 - 18 + {0,1}], O_0[2 v4 - 18 + {0,1}]
		if (cont) {
			if (tid == 255) {                              // V0[-5..-1] = O[-3], E[-2], O[-2], E[-1], O[-1] <- O[1021], E[1022], O[1022], E[1023], O[1023]
				const uint32_t a0 = lo0[FF_RAW / 2 - 3 + FO0], a1 = le0[FF_RAW / 2 - 2 + FE0], a2 = lo0[FF_RAW / 2 - 2 + FO0],
				               a3 = le0[FF_RAW / 2 - 1 + FE0], a4 = lo0[FF_RAW / 2 - 1 + FO0];
				lo0[-3 + FO0] = a0; le0[-2 + FE0] = a1; lo0[-2 + FO0] = a2; le0[-1 + FE0] = a3; lo0[-1 + FO0] = a4;
			}
#pragma unroll
			for (int u = 0; u < 2; u++) {
				const int v4 = 9 + tid + 256 * u;              // rel = 4 (tid + 256 u) >= 0
				uint32_t s0, s1, s2, s3;
				if (STAGE2) { s0 = w[u].x; s1 = w[u].y; s2 = w[u].z; s3 = w[u].w; }
				else dec_contrib<false, ROTATE>(w[u], s0, s1, s2, s3, K);
				*reinterpret_cast<uint2 *>(&le0[2 * v4 - 18 + FE0]) = make_uint2(s0, s2);
				lo0[2 * v4 - 18 + FO0] = s1;
				lo0[2 * v4 - 17 + FO0] = s3;
			}
		} else {
#pragma unroll
			for (int u = 0; u < 3; u++) {
				const int v4 = tid + 256 * u;
				const int rel = 4 * v4 - 36;
				if (v4 < NV && !(first && rel < 0)) {
					uint32_t s0, s1, s2, s3;
					if (STAGE2) { s0 = w[u].x; s1 = w[u].y; s2 = w[u].z; s3 = w[u].w; }
					else dec_contrib<false, ROTATE>(w[u], s0, s1, s2, s3, K);    // the decimator's packed scale + rotate: 20 instructions per 4 samples
					*reinterpret_cast<uint2 *>(&le0[2 * v4 - 18 + FE0]) = make_uint2(s0, s2);
					lo0[2 * v4 - 18 + FO0] = s1;
					lo0[2 * v4 - 17 + FO0] = s3;
				}
			}
		}
		if (tt + 1 < tpw) {                                    // the next tile continues: its 512 vectors, no halo
#pragma unroll
			for (int u = 0; u < 2; u++)
				w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(braw + t0 + FF_RAW + 4 * (tid + 256 * u)));
		}
		if (!EASE && first && tid == 0) {                  // V0[-5..-1] = O[-3], E[-2], O[-2], E[-1], O[-1]
			lo0[-3 + FO0] = sm[0]; le0[-2 + FE0] = sm[1]; lo0[-2 + FO0] = sm[2]; le0[-1 + FE0] = sm[3]; lo0[-1 + FO0] = sm[4];
		}
		__syncthreads();

		// pass 0: outputs V1[i..i+3], i = -16 + 4q; they are E_1[i/2], O_1[i/2], E_1[i/2+1], O_1[i/2+1]
		if (FUSE >= 2 && cont && tid == 255) {                 // level 1's last five to its front, before this thread's quad overwrites them
			const uint32_t a0 = lo1[FF_RAW / 4 - 3 + FO1], a1 = le1[FF_RAW / 4 - 2 + FE1], a2 = lo1[FF_RAW / 4 - 2 + FO1],
			               a3 = le1[FF_RAW / 4 - 1 + FE1], a4 = lo1[FF_RAW / 4 - 1 + FO1];
			lo1[-3 + FO1] = a0; le1[-2 + FE1] = a1; lo1[-2 + FO1] = a2; le1[-1 + FE1] = a3; lo1[-1 + FO1] = a4;
		}
		for (int q = cont ? tid + 4 : tid; q < FF_RAW / 8 + 4; q += 256) {
			if (first && q < 4)
				continue;
			const int i = -16 + 4 * q;
			uint4 o = fifth_quad_eo<FMODE>(&le0[i - 2 + FE0], &lo0[i - 3 + FO0]);
			if (EASE && first)
				ease_fix(o, i, le0, lo0, FE0, FO0);
			if (FUSE == 1) {
				if (i >= 0)                                    // halo outputs belong to the previous tile
					*reinterpret_cast<uint4 *>(bout + t0 / 2 + i) = o;
			} else {
				*reinterpret_cast<uint2 *>(&le1[i / 2 + FE1]) = make_uint2(o.x, o.z);
				lo1[i / 2 + FO1] = o.y;
				lo1[i / 2 + 1 + FO1] = o.w;
			}
		}
		if (FUSE >= 2) {
			if (!EASE && first && tid == 0) {
				lo1[-3 + FO1] = sm[5]; le1[-2 + FE1] = sm[6]; lo1[-2 + FO1] = sm[7]; le1[-1 + FE1] = sm[8]; lo1[-1 + FO1] = sm[9];
			}
			__syncthreads();
			// pass 1: outputs V2[i..i+3], i = -8 + 4q
			if (FUSE >= 3 && cont && tid == 127) {             // level 2's last five to its front (thread 127 writes that tail below)
				const uint32_t a0 = lo2[FF_RAW / 8 - 3 + FO2], a1 = le2[FF_RAW / 8 - 2 + FE2], a2 = lo2[FF_RAW / 8 - 2 + FO2],
				               a3 = le2[FF_RAW / 8 - 1 + FE2], a4 = lo2[FF_RAW / 8 - 1 + FO2];
				lo2[-3 + FO2] = a0; le2[-2 + FE2] = a1; lo2[-2 + FO2] = a2; le2[-1 + FE2] = a3; lo2[-1 + FO2] = a4;
			}
			for (int q = cont ? tid + 2 : tid; q < FF_RAW / 16 + 2; q += 256) {
				if (first && q < 2)
					continue;
				const int i = -8 + 4 * q;
				uint4 o = fifth_quad_eo<FMODE>(&le1[i - 2 + FE1], &lo1[i - 3 + FO1]);
				if (EASE && first)
					ease_fix(o, i, le1, lo1, FE1, FO1);
				if (FUSE == 2) {
					if (i >= 0)
						*reinterpret_cast<uint4 *>(bout + t0 / 4 + i) = o;
				} else {
					*reinterpret_cast<uint2 *>(&le2[i / 2 + FE2]) = make_uint2(o.x, o.z);
					lo2[i / 2 + FO2] = o.y;
					lo2[i / 2 + 1 + FO2] = o.w;
				}
			}
		}
		if (FUSE >= 3) {
			if (!EASE && first && tid == 0) {
				lo2[-3 + FO2] = sm[10]; le2[-2 + FE2] = sm[11]; lo2[-2 + FO2] = sm[12]; le2[-1 + FE2] = sm[13]; lo2[-1 + FO2] = sm[14];
			}
			__syncthreads();
			// pass 2: outputs V3[i..i+3], i = 4q
			if (tid < FF_RAW / 32) {
				const int i = 4 * tid;
				uint4 o = fifth_quad_eo<FMODE>(&le2[i - 2 + FE2], &lo2[i - 3 + FO2]);
				if (EASE && first)
					ease_fix(o, i, le2, lo2, FE2, FO2);
				*reinterpret_cast<uint4 *>(bout + t0 / 8 + i) = o;
			}
		}
		if (FUSE == 1)
			__syncthreads();                               // level 0 is rewritten at the top of the next turn
	}
}

// The first three fifth_order passes on the raw capture WITHOUT LDS and WITHOUT barriers (round 3; the LDS-tiled kernel above
// spent 70 % of its wave-time parked: its HBM, VALU and LDS phases ran one after the other between four barriers per tile, and
// it issued ~21 instructions per sample).  A decimating FIR keeps its taps next door when every lane owns a CONTIGUOUS run (for three passes):
//   lane l holds 8 consecutive level-0 samples x[8l .. 8l+7] (two 16-byte loads, scale + rotate as in the decimator),
//   level 1:  y[4l+k] = W(x[8l+2k-5 .. 8l+2k]),  k < 4 -- its own registers plus the left neighbour's x3..x7,
//   level 2:  z[2l+k] = W(y[4l+2k-5 .. 4l+2k]),  k < 2 -- own, the neighbour's y0..y3 and the neighbour-but-one's y3,
//   level 3:  w[l]    = W(z[2l-5 .. 2l])              -- z of lanes l-3 .. l,
// and "the left neighbour's register" is one v_mov_b32_dpp wave_shr:1 (two or three of them for lanes l-2, l-3): 15 moves, 7
// windows of packed int16 arithmetic and 40 instructions of scale/rotate per lane and 8 samples -- ~14 per sample, no LDS, no
// s_barrier, no s_waitcnt but the loads'.  What wave_shr shifts into lane 0 is not data, so the first five lanes of a wave only
// feed the others: a wave yields 59 outputs from 472 new samples and re-reads 40 (8.5 %).  At a callback block's start those
// five lanes are where the history belongs: k_fm_fifth_seams' five samples per level (the block seam rule, rtl_fm.c:416-432)
// are written over lane 4's x3..x7, over y3 of lane 3 and y0..y3 of lane 4, over z1 of lane 2 and z0, z1 of lanes 3 and 4.
// A wave never straddles a block.  n % 8 == 0.
// Measured (round 3, 8 GiB steps): with THREE passes this form is no faster than the LDS-tiled one (1.90 vs 1.92 ms alone: both move
// 9.7 GB at the ~5.1 TB/s this part gives a read stream with 11 % of writes mixed in; in the -F 9 chain the LDS kernel is 3-5 % ahead) --
// but it goes a level deeper for 6 more instructions per 16 samples, and that cuts the bytes: with FOUR passes -F ds=128 went
// 1.02 -> 1.24-1.33 TSample/s.  Five passes (1/32 out) bought nothing more: the kernel is then bound by its reads (5.5-5.9 TB/s).
#define FR_OUT 59                                         // outputs per wave
__device__ __forceinline__ uint32_t fr_shr(uint32_t v)
{
	// wave_shr:1 -- lane l takes lane l-1's value; bound_ctrl: lane 0 gets zero, so no `old` operand has to be set up per move
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}

// Written once for any depth: LV passes, 2^LV samples per lane (LV = 4, the default first group of a cascade of four or more passes: its
// output -- written once, read once -- is 1/16 of the capture; 3 and 5 are instantiated too).  One level: out[k] = W(X[2k-5 .. 2k]) over the
// lane's NIN values, X[-t] (t = 1..5) sitting `hops` = ceil(t / NIN) lanes to the left at register NIN * hops - t -- that many wave_shr moves
// away; at a block's start the same five places (in lanes 5 - hops) take the seam history instead.  WIDE levels (the fourth pass on: its
// inputs reach 1024) sum in 32 bits.  Seams: 5 LV dwords per block (k_fm_fifth_seams<.., LV>).  n % 2^LV == 0.
// MODE: fifth_any's (0 packed int16 sums, 1 32-bit sums of values below 2^14, 2 any int16: rx_power's raw buffers)
template <int NIN, int MODE>
__device__ __forceinline__ void fr_level(uint32_t (&in)[NIN], uint32_t (&out)[NIN / 2], unsigned lane, bool first, const uint32_t *__restrict__ sm)
{
	if (first) {
#pragma unroll
		for (int t = 1; t <= 5; t++) {
			const int hops = (t + NIN - 1) / NIN, idx = NIN * hops - t;
			if (lane == 5u - (unsigned)hops)
				in[idx] = sm[5 - t];
		}
	}
	uint32_t xm[6];                                           // xm[t] = X[-t]
#pragma unroll
	for (int t = 1; t <= 5; t++) {
		const int hops = (t + NIN - 1) / NIN, idx = NIN * hops - t;
		uint32_t v = fr_shr(in[idx]);
		if (hops > 1) v = fr_shr(v);
		if (hops > 2) v = fr_shr(v);
		xm[t] = v;
	}
#pragma unroll
	for (int k = 0; k < NIN / 2; k++) {
		uint32_t tap[6];
#pragma unroll
		for (int q = 0; q < 6; q++) {
			const int i = 2 * k - 5 + q;
			tap[q] = i >= 0 ? in[i >= 0 ? i : 0] : xm[i < 0 ? -i : 1];
		}
		out[k] = fifth_any<MODE>(tap[0], tap[1], tap[2], tap[3], tap[4], tap[5]);
	}
}

// LEFT more levels on NIN values per lane; DONE = passes already behind them (their seams at sm + 5 DONE; 32-bit sums from the fourth on)
// FIXED >= 0: that arithmetic mode at every level (rx_power: 2)
template <int NIN, int LEFT, int DONE, int FIXED = -1>
__device__ __forceinline__ void fr_cascade(uint32_t (&in)[NIN], uint32_t (&res)[NIN >> LEFT], unsigned lane, bool first, const uint32_t *__restrict__ sm)
{
	if constexpr (LEFT == 0) {
#pragma unroll
		for (int k = 0; k < NIN; k++)
			res[k] = in[k];
	} else {
		uint32_t o[NIN / 2];
		fr_level<NIN, FIXED >= 0 ? FIXED : (DONE >= 3 ? 1 : 0)>(in, o, lane, first, sm + 5 * DONE);
		fr_cascade<NIN / 2, LEFT - 1, DONE + 1, FIXED>(o, res, lane, first, sm);
	}
}

// the -A fast discriminator on two packed samples (multiply_conjugate + fast_atan2, rtl_fm.c:467-513); SMALL: |cr| + |cj| < 2^24
template <bool SMALL>
__device__ __forceinline__ int disc_fast(uint32_t a, uint32_t b)
{
	int cr, cj;
	mul_conj_pk(a, b, cr, cj);
	return fast_atan2_dev<SMALL>(cj, cr);
}

// generic_fir's sum over the nine samples w[0..8] BEFORE an output (rtl_fm.c:442-465); 24-bit multiplies, see k_fm_droop
__device__ __forceinline__ int mad24(int a, int b, int c)
{
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ uint32_t droop9(const uint32_t *w, int f1, int f2, int f3, int f4, int f5)
{
	const int si = mad24(lo16(w[0]) + lo16(w[8]), f1, mad24(lo16(w[1]) + lo16(w[7]), f2, mad24(lo16(w[2]) + lo16(w[6]), f3,
	               mad24(lo16(w[3]) + lo16(w[5]), f4, __mul24(lo16(w[4]), f5)))));
	const int sq = mad24(hi16(w[0]) + hi16(w[8]), f1, mad24(hi16(w[1]) + hi16(w[7]), f2, mad24(hi16(w[2]) + hi16(w[6]), f3,
	               mad24(hi16(w[3]) + hi16(w[5]), f4, __mul24(hi16(w[4]), f5)))));
	return pack_iq(si >> 15, sq >> 15);
}

// DD == 0: the cascade alone, one output per lane into `out`.
// DD != 0: the WHOLE -F chain of a cascade that fits the group -- LV passes, then (DD == 2) the droop FIR, then the -A fast discriminator --
// with nothing but pcm leaving the kernel (the `-M wbfm -F 9` chain wrote the 1/8-rate stream and read it back: 2 of its 13 GB per step).
// A lane then owns 4 << LV samples and four level-LV outputs w[4l .. 4l+3]; the FIR's nine samples before an output sit in the two lanes to
// the left and one register of the third (9 wave_shr moves), the discriminator's previous FIR output one more move away; the five halo
// lanes cover that (35 + 80 samples of 160).  At a block's start the previous block's last ten level-LV samples (k_fm_fifth_tails: the FIR and
// the discriminator run on across callback blocks, rtl_fm.c:442-465, 485-513) are written over the halo lanes' w before the FIR.  Two things
// are left to a small kernel behind this one (k_fm_dd_edges, stream B: they need the previous run's carries and the libm flag list): each
// block's FIRST demodulated sample -- polar_discriminant in double, rtl_fm.c:476-483, 667-682 -- and pre_r/pre_j; for them the kernel leaves
// each block's first and last FIR output in `edges`.
template <bool ROTATE, int LV, int DD, int TW = 1>
__global__ __launch_bounds__(256) void k_fm_fifth_regn(const uint32_t *__restrict__ iq, unsigned n, unsigned tiles_per_block, unsigned wgs_per_block,
                                                       unsigned total_wgs, const uint32_t *__restrict__ seams, uint32_t *__restrict__ out,
                                                       const uint32_t *__restrict__ tails, int f1, int f2, int f3, int f4, int f5,
                                                       int16_t *__restrict__ pcm, int pcm_chl2, uint32_t *__restrict__ edges)
{
	constexpr int NOUT = DD ? 4 : 1, R = NOUT << LV, NP = R / 4;   // outputs and samples per lane, 16-byte pieces per lane
	static_assert(LV >= 3 && LV <= 5 && NP <= 16, "3 to 5 passes");
	const unsigned lane = threadIdx.x & 63u;
	const unsigned per = gridDim.x >> 3;                     // XCD-contiguous order: workgroup b runs on XCD b % 8, every XCD takes one contiguous eighth
	const unsigned wgi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	if (wgi >= total_wgs)
		return;
	const unsigned blk32 = wgi / wgs_per_block;
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// a workgroup takes 4 TW consecutive tiles of its block, four at a time: wave wv walks tiles base + wv, base + 4 + wv, ...
	unsigned tile = (wgi - blk32 * wgs_per_block) * (4 * TW) + wv;
	if (tile >= tiles_per_block)
		return;
	const u64 blk = blk32;
	const uint32_t *braw = iq + blk * (u64)n;
	const unsigned K = n >> LV;
	const scale_k SK = scale_consts();
	const uint32_t *sm = seams + blk * (5 * LV);
	// lane l <-> block samples [s, s + R), s = 59 R tile + R (l - 5): the tile comes in through LDS-DMA (whole lines), wave-private.
	// Slot (h, lane) of the stage takes the piece its READER wants there: lane l reads its j-th piece from slot NP l + (j + rot(l)) % NP,
	// rot(l) = l NP / 16 -- sixteen neighbouring lanes' b128 reads then fall on sixteen different bank groups
	__shared__ u32x4 stage[4][64 * NP];
	const int s_max = (int)n - 4;
	const unsigned rot = (lane * NP) / 16u;
	auto fetch = [&](unsigned t) {
#pragma unroll
		for (int h = 0; h < NP; h++) {
			const unsigned slot = 64u * h + lane, sl = slot / NP, sj = (slot - (sl * NP) / 16u) % NP;   // slot % NP - rot(sl), mod NP
			int sp = (int)(t * (FR_OUT * R)) - 5 * R + (int)(4u * (sl * NP + sj));
			sp = sp < 0 ? 0 : (sp > s_max ? s_max : sp);
			__builtin_amdgcn_global_load_lds((const void *)(braw + sp), (__attribute__((address_space(3))) void *)&stage[wv][64 * h], 16, 0, 2);
		}
	};
	fetch(tile);
#pragma unroll 1
	for (int it = 0; it < TW; it++, tile += 4) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		uint32_t x[R];
		if constexpr (TW > 1) {
			// the stage is free as soon as its pieces sit in registers: the next tile's loads fly while this one is computed
			u32x4 raw[NP];
#pragma unroll
			for (int j = 0; j < NP; j++)
				raw[j] = stage[wv][NP * lane + ((j + rot) % NP)];
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (it + 1 < TW && tile + 4 < tiles_per_block)
				fetch(tile + 4);
#pragma unroll
			for (int j = 0; j < NP; j++)
				dec_contrib<false, ROTATE>(raw[j], x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3], SK);
		} else {
#pragma unroll
			for (int j = 0; j < NP; j++)
				dec_contrib<false, ROTATE>(stage[wv][NP * lane + ((j + rot) % NP)], x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3], SK);
		}
		const bool first = tile == 0;                            // wave-uniform
		uint32_t w[NOUT];
		fr_cascade<R, LV, 0>(x, w, lane, first, sm);
		const unsigned j = tile * FR_OUT + lane - 5u;            // the lane's place among the block's K / NOUT lanes
		if constexpr (DD == 0) {
			if (lane >= 5u && j < K)
				__builtin_nontemporal_store(w[0], out + blk * (u64)K + j);
		} else {
			if (first) {
				const uint32_t *tl = tails + blk * 10;               // the previous block's level-LV samples K-10 .. K-1
				if (lane == 4u) { w[0] = tl[6]; w[1] = tl[7]; w[2] = tl[8]; w[3] = tl[9]; }
				else if (lane == 3u) { w[0] = tl[2]; w[1] = tl[3]; w[2] = tl[4]; w[3] = tl[5]; }
				else if (lane == 2u) { w[2] = tl[0]; w[3] = tl[1]; }
			}
			uint32_t y[4];
			if constexpr (DD == 2) {
				uint32_t W[13];                                      // W[i] = level-LV sample 4l - 9 + i
#pragma unroll
				for (int k = 0; k < 4; k++) {
					W[9 + k] = w[k];
					W[5 + k] = fr_shr(w[k]);
					W[1 + k] = fr_shr(W[5 + k]);
				}
				W[0] = fr_shr(W[4]);
#pragma unroll
				for (int k = 0; k < 4; k++)
					y[k] = droop9(W + k, f1, f2, f3, f4, f5);
			} else {
#pragma unroll
				for (int k = 0; k < 4; k++)
					y[k] = w[k];
			}
			const uint32_t yp = fr_shr(y[3]);
			if (lane >= 5u && 4u * j < K) {
				// without the FIR the samples stay below 2^10 << (LV - 3) and |cr| + |cj| below 2^24 (LV <= 3): the short division
				constexpr bool SMALL = DD == 1 && LV == 3;
				const int r0 = j ? disc_fast<SMALL>(y[0], yp) : 0;    // a block's first sample: k_fm_dd_edges
				const int r1 = disc_fast<SMALL>(y[1], y[0]), r2 = disc_fast<SMALL>(y[2], y[1]), r3 = disc_fast<SMALL>(y[3], y[2]);
				const u64 m0 = blk * (u64)K + 4u * j;
				int16_t *dst = pcm + pcm_index(m0, pcm_chl2);        // four consecutive samples stay inside one 16-byte unit of the tiled layout
				*reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)(uint16_t)r0 | ((uint32_t)(uint16_t)r1 << 16), (uint32_t)(uint16_t)r2 | ((uint32_t)(uint16_t)r3 << 16));
				if (j == 0)
					edges[2 * blk] = y[0];
				if (4u * j + 4u == K)
					edges[2 * blk + 1] = y[3];
			}
		}
		if (TW > 1 && tile + 4 >= tiles_per_block)
			break;
	}
}

// ------------------------------------------------------------------ rx_power: the stateless cascade in registers (round 6)
//
// rx_power -F: `downsample_passes` stateless fifth_order passes per buffer (rtl_power.c:582-607 via 734-737), the droop FIR (626-654 via 739-742),
// then remove_dc (609-624).  Four passes through the LDS-tiled kernels crossed HBM nine times at the 1/16 rate (three passes out, the fourth in and
// out, the FIR in and out, the dc sums in, the transform's head in) -- 1.73 x the input bytes at N = 2^14.  Here: k_fm_fifth_regn's shape on RAW int16
// (no scale, no rotation, 32-bit tap sums at every level: fifth_int), four level-LV outputs per lane, the FIR over the neighbouring lanes' outputs
// (wave_shr moves, as the rx_fm whole-chain kernel does), one 16-byte store per lane, and the buffer's dc sums on the way out (a wave never straddles a
// buffer: one wave reduction, two int64 atomics).  The buffers are STATELESS and eased in: the first five outputs of every pass come from special
// formulas on the pass input's samples 0..8 (rtl_power.c:595-597 and the d == e quirk of the loop's first turns) -- they contaminate exactly the
// first five outputs of the NEXT pass and nothing else (output k >= 5 reads inputs >= 5).  This kernel computes the regular formula everywhere (the
// halo lanes of a buffer's first tile hold clamped garbage); k_pw_fifth_fix then recomputes a buffer's first 5 (with the FIR: 14) final samples from
// the raw buffer, literally, overwrites them and corrects the sums by what changed.
// The cascade on PAIRS of one component: register j of a level holds (X[2j-1], X[2j]) of I (or of Q) -- a window X[2k-5 .. 2k] is then three
// whole registers, pairs k-2, k-1, k, and its 32-bit tap sum three v_dot2_i32_i16 with the coefficient pairs (1,5), (10,10), (5,1): six per complex
// output where fifth_int spends twelve v_mad_i32_i16.  Cost: one v_perm_b32 per raw sample (I|Q words -> pairs), three wave_shr moves per level and
// component (the left lane's last two pairs and its last output), the same shift-and-pack as before -- now of two neighbouring outputs of one component.
typedef short pp_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pp_dot(uint32_t pair, uint32_t coef, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(pp_s16x2, pair), __builtin_bit_cast(pp_s16x2, coef), acc, false);
}
// NPAIR pairs in (samples -1 .. 2 NPAIR - 2 of the lane's window), NPAIR 32-bit tap sums out (outputs 0 .. NPAIR - 1, rtl_power.c:599-606 before the shift)
template <int NPAIR>
__device__ __forceinline__ void pp_level(const uint32_t (&p)[NPAIR], int (&sum)[NPAIR])
{
	static_assert(NPAIR >= 2, "the two pairs to the left sit in ONE neighbouring lane");
	const uint32_t m1 = fr_shr(p[NPAIR - 1]), m2 = fr_shr(p[NPAIR - 2]);          // pairs -1, -2
#pragma unroll
	for (int k = 0; k < NPAIR; k++) {
		const uint32_t a = k >= 2 ? p[k >= 2 ? k - 2 : 0] : (k == 1 ? m1 : m2), b = k >= 1 ? p[k >= 1 ? k - 1 : 0] : m1;
		// (the chain's first link in the three-source form with the inline constant 0: the accumulating two-source form wants a zeroed register first)
		int s0;
		asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(s0) : "v"(a), "s"(0x00050001u));
		sum[k] = pp_dot(p[k], 0x00010005u, pp_dot(b, 0x000a000au, s0));
	}
}
// the next level's pairs: ((sum[2m-1] >> 4) & 0xffff) | (sum[2m] >> 4) << 16 -- the int16 stores of rtl_power.c:599-606, two at a time
template <int NPAIR>
__device__ __forceinline__ void pp_pack(const int (&sum)[NPAIR], uint32_t (&q)[NPAIR / 2])
{
	const int left = (int)fr_shr((uint32_t)sum[NPAIR - 1]);                       // output -1
#pragma unroll
	for (int m = 0; m < NPAIR / 2; m++) {
		const int lo = m ? sum[m ? 2 * m - 1 : 0] : left;
		q[m] = (uint32_t)__builtin_amdgcn_ubfe((unsigned)lo, 4u, 16u) | (((unsigned)sum[2 * m] << 12) & 0xffff0000u);
	}
}
template <int NPAIR, int LEFT>
__device__ __forceinline__ void pp_cascade(const uint32_t (&pi)[NPAIR], const uint32_t (&pq)[NPAIR], uint32_t (&res)[NPAIR >> (LEFT - 1)])
{
	int si[NPAIR], sq[NPAIR];
	pp_level<NPAIR>(pi, si);
	pp_level<NPAIR>(pq, sq);
	if constexpr (LEFT == 1) {
#pragma unroll
		for (int k = 0; k < NPAIR; k++)
			res[k] = (uint32_t)__builtin_amdgcn_ubfe((unsigned)si[k], 4u, 16u) | (((unsigned)sq[k] << 12) & 0xffff0000u);
	} else {
		uint32_t qi[NPAIR / 2], qq[NPAIR / 2];
		pp_pack<NPAIR>(si, qi);
		pp_pack<NPAIR>(sq, qq);
		pp_cascade<NPAIR / 2, LEFT - 1>(qi, qq, res);
	}
}
// LV stateless passes on the lane's R raw I|Q words -> its R >> LV level-LV words (what fr_cascade<R, LV, 0, 2> computes)
template <int R, int LV>
__device__ __forceinline__ void pp_cascade_raw(const uint32_t (&x)[R], uint32_t (&w)[R >> LV])
{
	uint32_t pi[R / 2], pq[R / 2];
	const uint32_t xm1 = fr_shr(x[R - 1]);
#pragma unroll
	for (int j = 0; j < R / 2; j++) {
		const uint32_t a = j ? x[j ? 2 * j - 1 : 0] : xm1, b = x[2 * j];
		pi[j] = __builtin_amdgcn_perm(b, a, 0x05040100u);                          // (I of a, I of b)
		pq[j] = __builtin_amdgcn_perm(b, a, 0x07060302u);                          // (Q of a, Q of b)
	}
	pp_cascade<R / 2, LV>(pi, pq, w);
}

template <int LV, bool FIR, int TW>
__global__ __launch_bounds__(256) void k_pw_fifth_regn(const uint32_t *__restrict__ in, unsigned n, unsigned in_stride, unsigned tiles_per_block, unsigned wgs_per_block,
                                                       unsigned total_wgs, uint32_t *__restrict__ out, unsigned out_stride, int f1, int f2, int f3, int f4, int f5,
                                                       int2 *__restrict__ wave_part)
{
	constexpr int NOUT = 4, R = NOUT << LV, NP = R / 4;
	static_assert(LV >= 1 && NP <= 16, "at most 64 samples per lane");
	const unsigned lane = threadIdx.x & 63u;
	const unsigned per = gridDim.x >> 3;
	const unsigned wgi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	if (wgi >= total_wgs)
		return;
	const unsigned blk32 = wgi / wgs_per_block;
	const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// a workgroup takes 4 TW consecutive tiles of its buffer, four at a time: wave wv walks tiles base + wv, base + 4 + wv, ... -- with 16 KiB of
	// stage per wave only ten waves fit a CU, so a wave keeps the NEXT tile's loads in flight behind the tile it computes
	unsigned tile = (wgi - blk32 * wgs_per_block) * (4 * TW) + wv;
	if (tile >= tiles_per_block) {
		if (wave_part && lane == 0)
			wave_part[(size_t)wgi * 4 + wv] = make_int2(0, 0);
		return;
	}
	const u64 blk = blk32;
	const uint32_t *braw = in + blk * (u64)in_stride;
	const unsigned K = n >> LV;
	__shared__ u32x4 stage[4][64 * NP];
	const int s_max = (int)n - 4;
	const unsigned rot = (lane * NP) / 16u;
	auto fetch = [&](unsigned t) {
#pragma unroll
		for (int h = 0; h < NP; h++) {
			const unsigned slot = 64u * h + lane, sl = slot / NP, sj = (slot - (sl * NP) / 16u) % NP;
			int sp = (int)(t * (FR_OUT * R)) - 5 * R + (int)(4u * (sl * NP + sj));
			sp = sp < 0 ? 0 : (sp > s_max ? s_max : sp);
			__builtin_amdgcn_global_load_lds((const void *)(braw + sp), (__attribute__((address_space(3))) void *)&stage[wv][64 * h], 16, 0, 2);
		}
	};
	fetch(tile);
	int si = 0, sq = 0;                                      // the wave's share of the buffer's dc sums: 4 TW int16 per lane, far inside int32
#pragma unroll 1
	for (int it = 0; it < TW; it++, tile += 4) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		uint32_t x[R];
#pragma unroll
		for (int j = 0; j < NP; j++) {
			const u32x4 v = stage[wv][NP * lane + ((j + rot) % NP)];
			x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
		}
		if constexpr (TW > 1) {
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (it + 1 < TW && tile + 4 < tiles_per_block)
				fetch(tile + 4);
		}
		uint32_t w[NOUT];
		pp_cascade_raw<R, LV>(x, w);
		uint32_t y[4];
		if constexpr (FIR) {
			uint32_t W[13];                                      // W[i] = level-LV sample 4l - 9 + i
#pragma unroll
			for (int k = 0; k < 4; k++) {
				W[9 + k] = w[k];
				W[5 + k] = fr_shr(w[k]);
				W[1 + k] = fr_shr(W[5 + k]);
			}
			W[0] = fr_shr(W[4]);
#pragma unroll
			for (int k = 0; k < 4; k++)
				y[k] = droop9(W + k, f1, f2, f3, f4, f5);
		} else {
#pragma unroll
			for (int k = 0; k < 4; k++)
				y[k] = w[k];
		}
		const unsigned j = tile * FR_OUT + lane - 5u;
		if (lane >= 5u && 4u * j < K) {
			__builtin_nontemporal_store((u32x4){y[0], y[1], y[2], y[3]}, reinterpret_cast<u32x4 *>(out + blk * (u64)out_stride + 4u * j));
			si += lo16(y[0]) + lo16(y[1]) + lo16(y[2]) + lo16(y[3]);
			sq += hi16(y[0]) + hi16(y[1]) + hi16(y[2]) + hi16(y[3]);
		}
		if (TW > 1 && tile + 4 >= tiles_per_block)
			break;
	}
	if (wave_part) {
		// the wave's share as ONE plain store (k_pw_fifth_fix adds a buffer's shares): int64 atomics here kept the wave resident for their round trip
		// (the boxcar decimator's measurement, k_fm_decimate<.., DCS>)
#define DPP_ADD(V, CTRL, ROWS) V += __builtin_amdgcn_update_dpp(0, V, CTRL, ROWS, 0xf, true)
		DPP_ADD(si, 0x111, 0xf); DPP_ADD(sq, 0x111, 0xf); DPP_ADD(si, 0x112, 0xf); DPP_ADD(sq, 0x112, 0xf);
		DPP_ADD(si, 0x114, 0xf); DPP_ADD(sq, 0x114, 0xf); DPP_ADD(si, 0x118, 0xf); DPP_ADD(sq, 0x118, 0xf);
		DPP_ADD(si, 0x142, 0xa); DPP_ADD(sq, 0x142, 0xa); DPP_ADD(si, 0x143, 0xc); DPP_ADD(sq, 0x143, 0xc);
#undef DPP_ADD
		if (lane == 63)
			wave_part[(size_t)wgi * 4 + wv] = make_int2(si, sq);
	}
}

// the first NFIX final samples of every buffer, literally, over what k_pw_fifth_regn left there; the dc sums follow.  One wave per buffer: the levels' first
// samples in LDS, a lane per output (at most 105 per level), wave-level ordering between the levels.
__device__ __forceinline__ uint32_t pw_fifth_head1(const uint32_t *s, int k)
{
	int r[2];
#pragma unroll
	for (int h = 0; h < 2; h++) {
#define G(i) (h ? hi16(s[i]) : lo16(s[i]))
		if (k == 0) r[h] = ((G(0) + G(1)) * 10 + (G(2) + G(3)) * 5 + G(3) + G(5)) >> 4;
		else if (k == 1) r[h] = ((G(1) + G(2)) * 10 + (G(0) + G(3)) * 5 + G(4) + G(5)) >> 4;
		else if (k == 2) r[h] = (G(0) + (G(1) + G(4)) * 5 + (G(2) + G(3)) * 10 + G(5)) >> 4;
		else if (k == 3) r[h] = (G(2) + (G(3) + G(5)) * 5 + (G(4) + G(5)) * 10 + G(6)) >> 4;
		else if (k == 4) r[h] = (G(4) + (G(5) + G(7)) * 5 + (G(5) + G(6)) * 10 + G(8)) >> 4;
		else r[h] = (G(2 * k - 5) + (G(2 * k - 4) + G(2 * k - 1)) * 5 + (G(2 * k - 3) + G(2 * k - 2)) * 10 + G(2 * k)) >> 4;
#undef G
	}
	return pack_iq(r[0], r[1]);
}

template <int LV, bool FIR>
__global__ __launch_bounds__(256) void k_pw_fifth_fix(const uint32_t *__restrict__ in, unsigned n_bufs, unsigned in_stride, uint32_t *__restrict__ out, unsigned out_stride,
                                                      const int *__restrict__ fir, i64 *__restrict__ sums, const int2 *__restrict__ wave_part, unsigned parts_per_buf)
{
	constexpr int NFIX = FIR ? 14 : 5, CMAX = 216;               // samples needed at level p for NFIX at level LV: c(p - 1) = max(2 c(p) - 1, 9); LV = 4: 209 raw at most
	static_assert(LV == 4, "the level buffers are sized for four passes");
	__shared__ uint32_t lv[4][2][CMAX];
	const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const unsigned b = blockIdx.x * 4 + wv;
	if (b >= n_bufs)
		return;
	int cnt[LV + 1];
	cnt[LV] = NFIX;
#pragma unroll
	for (int p = LV; p > 0; p--)
		cnt[p - 1] = 2 * cnt[p] - 1 > 9 ? 2 * cnt[p] - 1 : 9;
	const uint32_t *src = in + (u64)b * in_stride;
	uint32_t *cur = lv[wv][0], *nxt = lv[wv][1];
	for (int i = (int)lane; i < cnt[0]; i += 64)
		cur[i] = src[i];
#pragma unroll
	for (int p = 1; p <= LV; p++) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		for (int k = (int)lane; k < cnt[p]; k += 64)
			nxt[k] = pw_fifth_head1(cur, k);
		uint32_t *t = cur; cur = nxt; nxt = t;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	// cur[0 .. NFIX): level-LV samples; with the FIR: samples 0..8 pass through, 9.. are filtered from the nine before (rtl_power.c:626-654)
	i64 di = 0, dq = 0;
	if (lane < NFIX) {
		const int d = (int)lane;
		uint32_t fin;
		if (!FIR || d < 9) {
			fin = cur[d];
		} else {
			const uint32_t *h = cur + d - 9;
			const int si = __mul24(lo16(h[0]) + lo16(h[8]), fir[1]) + __mul24(lo16(h[1]) + lo16(h[7]), fir[2]) + __mul24(lo16(h[2]) + lo16(h[6]), fir[3]) +
			               __mul24(lo16(h[3]) + lo16(h[5]), fir[4]) + __mul24(lo16(h[4]), fir[5]);
			const int sq = __mul24(hi16(h[0]) + hi16(h[8]), fir[1]) + __mul24(hi16(h[1]) + hi16(h[7]), fir[2]) + __mul24(hi16(h[2]) + hi16(h[6]), fir[3]) +
			               __mul24(hi16(h[3]) + hi16(h[5]), fir[4]) + __mul24(hi16(h[4]), fir[5]);
			fin = pack_iq(si >> 15, sq >> 15);
		}
		uint32_t *dst = out + (u64)b * out_stride + d;
		const uint32_t old = *dst;
		di = lo16(fin) - lo16(old);
		dq = hi16(fin) - hi16(old);
		*dst = fin;
	}
	if (sums) {
		// the buffer's sums: the shares k_pw_fifth_regn's waves left (computed with the regular formula everywhere) + what the first samples changed
		for (unsigned k = lane; k < parts_per_buf; k += 64) {
			const int2 p = wave_part[(size_t)b * parts_per_buf + k];
			di += p.x;
			dq += p.y;
		}
		for (int off = 32; off; off >>= 1) { di += __shfl_down(di, off); dq += __shfl_down(dq, off); }
		if (lane == 0) {
			sums[2 * (u64)b] = di;                               // the only writer: no atomics, nothing to zero first
			sums[2 * (u64)b + 1] = dq;
		}
	}
}

// the previous block's last ten level-LV samples for every block (block 0: the carried droop history hist[0..8] = s[-9..-1], or nothing
// without the FIR), and behind the last block the new droop history, its last nine (rtl_fm.c:453-463)
template <bool ROTATE, int LV>
__global__ void k_fm_fifth_tails(const uint32_t *__restrict__ iq, u64 n_blocks, unsigned n, const int16_t *__restrict__ droop_in,
                                 int16_t *__restrict__ droop_out, uint32_t *__restrict__ tails)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 b = gid / 16;
	const int q = (int)(gid % 16);
	if (q >= 10 || b > n_blocks)
		return;
	const int K = (int)(n >> LV);
	if (b == 0) {
		tails[q] = (droop_in && q) ? pack_iq(droop_in[q - 1], droop_in[9 + q - 1]) : 0u;
		return;
	}
	const uint32_t v = level_val<LV, ROTATE, false>(iq + (b - 1) * (u64)n, K - 10 + q);
	if (b < n_blocks)
		tails[b * 10 + q] = v;
	else if (droop_out && q) {
		droop_out[q - 1] = (int16_t)lo16(v);
		droop_out[9 + q - 1] = (int16_t)hi16(v);
	}
}

// what k_fm_fifth_regn<.., DD> leaves: each block's first demodulated sample (libm, with the 2^-33 window and the host fix-up records
// like k_fm_droop_disc) from the block's first FIR output and the previous block's last -- the carried pre_r/pre_j for block 0 -- and the
// run's pre_r/pre_j out
__global__ void k_fm_dd_edges(const uint32_t *__restrict__ edges, u64 n_blocks, u64 K, int16_t *__restrict__ pcm, int pcm_chl2, rxk_fm_dev *__restrict__ dev,
                              rxk_flag_rec *__restrict__ flag_list, int *__restrict__ flag_cnt, int flag_all)
{
	const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks)
		return;
	const uint32_t a = edges[2 * b];
	const int ar = lo16(a), aj = hi16(a);
	int br, bj;
	if (b) { br = lo16(edges[2 * b - 1]); bj = hi16(edges[2 * b - 1]); }
	else { br = dev->in_pre_r; bj = dev->in_pre_j; }
	const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
	const u64 m = b * K;
	// polar_discriminant, rtl_fm.c:476-483 (see k_fm_disc)
	const double v = atan2((double)cj, (double)cr) / 3.14159 * 16384.0;
	int out = (int)v;
	if (v != 0.0 && (flag_forced(flag_all, m) || fabs(v - rint(v)) < RXK_LIBM_WINDOW)) {
		const int idx = atomicAdd(flag_cnt, 1);
		if (idx < RXK_FLAG_CAP) {
			rxk_flag_rec rec;
			rec.m = m; rec.ar = ar; rec.aj = aj; rec.br = br; rec.bj = bj;
			flag_list[idx] = rec;
		}
		if (flag_all > 1)
			out += 77;
	}
	pcm[pcm_index(m, pcm_chl2)] = (int16_t)out;
	if (b == n_blocks - 1) {
		const uint32_t l = edges[2 * b + 1];
		dev->out_pre_r = lo16(l);
		dev->out_pre_j = hi16(l);
	}
}

// ------------------------------------------------------------------ F12 droop FIR

// rtl_fm.c:442-465: out[t] = (sum over the 9 samples BEFORE t) >> 15; hist carries across blocks,
// so over the concatenated post-cascade stream it is a plain FIR on s[t-9 .. t-1].
__global__ void k_fm_droop(const uint32_t *__restrict__ in, u64 M, const int *__restrict__ fir,
                           const int16_t *__restrict__ hist_in, int16_t *__restrict__ hist_out, uint32_t *__restrict__ out)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= M)
		return;
	int hi[9], hq[9];
#pragma unroll
	for (int j = 0; j < 9; j++) {
		const i64 idx = (i64)t - 9 + j;
		if (idx >= 0) {
			const uint32_t w = in[idx];
			hi[j] = lo16(w); hq[j] = hi16(w);
		} else {
			hi[j] = hist_in[9 + idx];              // hist[0..8] = s[-9..-1]
			hq[j] = hist_in[9 + 9 + idx];
		}
	}
	// sums of two int16 times cic_9_tables coefficients (|f| < 2^17): both fit 24 bits, the full-rate 24-bit multiply keeps the low
	// 32 bits of the product -- the reference's wrapping int arithmetic
	const int f1 = fir[1], f2 = fir[2], f3 = fir[3], f4 = fir[4], f5 = fir[5];
	const int si = __mul24(hi[0] + hi[8], f1) + __mul24(hi[1] + hi[7], f2) + __mul24(hi[2] + hi[6], f3) + __mul24(hi[3] + hi[5], f4) + __mul24(hi[4], f5);
	const int sq = __mul24(hq[0] + hq[8], f1) + __mul24(hq[1] + hq[7], f2) + __mul24(hq[2] + hq[6], f3) + __mul24(hq[3] + hq[5], f4) + __mul24(hq[4], f5);
	out[t] = pack_iq(si >> 15, sq >> 15);
	if (t == M - 1) {
		// new history = the last 9 INPUT samples s[M-9 .. M-1]
		for (int j = 0; j < 9; j++) {
			const i64 idx = (i64)M - 9 + j;
			if (idx >= 0) {
				const uint32_t w = in[idx];
				hist_out[j] = (int16_t)lo16(w);
				hist_out[9 + j] = (int16_t)hi16(w);
			} else {
				hist_out[j] = hist_in[9 + idx + 0];
				hist_out[9 + j] = hist_in[9 + 9 + idx];
			}
		}
	}
}


// F12 + F5/F6 in one pass (the `-M wbfm -F 9` chain: three fifth_order passes leave an eighth of the capture rate, and the
// one-thread-per-output k_fm_droop + dense k_fm_disc pair then costs more than the cascade).  A thread takes four
// consecutive outputs t0..t0+3 of the droop FIR -- out[t] = sum over the nine samples BEFORE t, rtl_fm.c:442-465 -- from four
// aligned 16-byte loads of the post-cascade stream (s[t0-12 .. t0+3]; the neighbours' loads hit the same lines), computes
// out[t0-1] once more for the first product, and demodulates: fast_atan2 for ordinary samples, the libm form (with the
// 2^-33 window and the host fix-up records) for each callback block's first sample.  lp_out != NULL also stores the FIR output
// (the drop-in hands lowpassed[] back); pcm goes out linear or tiled.
template <bool STORE_LP>
__global__ __launch_bounds__(256) void k_fm_droop_disc(
	const uint32_t *__restrict__ in, u64 M, const int *__restrict__ fir, const int16_t *__restrict__ hist_in, int16_t *__restrict__ hist_out,
	uint32_t *__restrict__ lp_out, u64 uniform_k, int16_t *__restrict__ pcm, int pcm_chl2, rxk_fm_dev *__restrict__ dev,
	rxk_flag_rec *__restrict__ flag_list, int *__restrict__ flag_cnt, int flag_all)
{
	const u64 t0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
	if (t0 >= M)
		return;
	// s[t0-12 .. t0+3]: stream samples, the carried history (hist[0..8] = s[-9..-1]) before the run, zero before that
	uint32_t sv[16];
	if (t0 >= 12 && t0 + 4 <= M) {
		const uint4 *q = reinterpret_cast<const uint4 *>(in + t0 - 12);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint4 w = q[k];
			sv[4 * k] = w.x; sv[4 * k + 1] = w.y; sv[4 * k + 2] = w.z; sv[4 * k + 3] = w.w;
		}
	} else {
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const i64 idx = (i64)t0 - 12 + k;
			sv[k] = idx >= (i64)M ? 0u : idx >= 0 ? in[idx] : idx >= -9 ? pack_iq(hist_in[9 + idx], hist_in[18 + idx]) : 0u;
		}
	}
	const int f1 = fir[1], f2 = fir[2], f3 = fir[3], f4 = fir[4], f5 = fir[5];
	uint32_t o[5];                                            // out[t0-1 .. t0+3]; out[t] uses s[t-9 .. t-1] = sv[t-t0+3 .. t-t0+11]
#pragma unroll
	for (int r = 0; r < 5; r++) {
		const int b = r + 2;
		// 24-bit multiplies: sums of two int16 and coefficients below 2^17 (see k_fm_droop)
		const int si = __mul24(lo16(sv[b]) + lo16(sv[b + 8]), f1) + __mul24(lo16(sv[b + 1]) + lo16(sv[b + 7]), f2) + __mul24(lo16(sv[b + 2]) + lo16(sv[b + 6]), f3) +
		               __mul24(lo16(sv[b + 3]) + lo16(sv[b + 5]), f4) + __mul24(lo16(sv[b + 4]), f5);
		const int sq = __mul24(hi16(sv[b]) + hi16(sv[b + 8]), f1) + __mul24(hi16(sv[b + 1]) + hi16(sv[b + 7]), f2) + __mul24(hi16(sv[b + 2]) + hi16(sv[b + 6]), f3) +
		               __mul24(hi16(sv[b + 3]) + hi16(sv[b + 5]), f4) + __mul24(hi16(sv[b + 4]), f5);
		o[r] = pack_iq(si >> 15, sq >> 15);
	}
	const int n = (int)((M - t0) < 4 ? (M - t0) : 4);
	if (STORE_LP) {
		if (n == 4)
			*reinterpret_cast<uint4 *>(lp_out + t0) = make_uint4(o[1], o[2], o[3], o[4]);
		else
			for (int r = 0; r < n; r++) lp_out[t0 + r] = o[1 + r];
	}
	int16_t res[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const u64 m = t0 + r;
		if (r >= n) { res[r] = 0; continue; }
		const uint32_t a = o[1 + r];
		int br, bj;
		if (m) { br = lo16(o[r]); bj = hi16(o[r]); }
		else { br = dev->in_pre_r; bj = dev->in_pre_j; }
		const int ar = lo16(a), aj = hi16(a);
		const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
		const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
		const bool first = (uniform_k & (uniform_k - 1)) ? (m % uniform_k) == 0 : (m & (uniform_k - 1)) == 0;
		int out;
		if (__builtin_expect(first, 0)) {
			// polar_discriminant, rtl_fm.c:476-483 (see k_fm_disc)
			const double v = atan2((double)cj, (double)cr) / 3.14159 * 16384.0;
			out = (int)v;
			if (v != 0.0 && (flag_forced(flag_all, m) || fabs(v - rint(v)) < RXK_LIBM_WINDOW)) {
				const int idx = atomicAdd(flag_cnt, 1);
				if (idx < RXK_FLAG_CAP) {
					rxk_flag_rec rec;
					rec.m = m; rec.ar = ar; rec.aj = aj; rec.br = br; rec.bj = bj;
					flag_list[idx] = rec;
				}
				if (flag_all > 1)
					out += 77;
			}
		} else {
			out = fast_atan2_dev(cj, cr);
		}
		res[r] = (int16_t)out;
		if (m == M - 1) {
			dev->out_pre_r = ar;
			dev->out_pre_j = aj;
		}
	}
	int16_t *dst = pcm + pcm_index(t0, pcm_chl2);             // four consecutive samples stay inside one 16-byte unit of the tiled layout
	if (n == 4)
		*reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16),
		                                             (uint32_t)(uint16_t)res[2] | ((uint32_t)(uint16_t)res[3] << 16));
	else
		for (int r = 0; r < n; r++) dst[r] = res[r];
	if (t0 + 4 >= M) {
		// new history = the last 9 INPUT samples s[M-9 .. M-1]
		for (int j = 0; j < 9; j++) {
			const i64 idx = (i64)M - 9 + j;
			const uint32_t w = idx >= 0 ? in[idx] : pack_iq(hist_in[9 + idx], hist_in[18 + idx]);
			hist_out[j] = (int16_t)lo16(w);
			hist_out[9 + j] = (int16_t)hi16(w);
		}
	}
}

// ------------------------------------------------------------------ callback pre-stage alone

// rtlsdr_callback's scale + rotate (rtl_fm.c:845-857) as an elementwise pass, for the
// drop-in callback that has to hand a host lowpassed[] back to the reference's threads.
__global__ void k_fm_prestage(const uint32_t *__restrict__ in, unsigned n, int rotate, uint32_t *__restrict__ out)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	int ri, rq;
	load_rot<false>(in, i, rotate ? i : 0u, ri, rq);
	out[i] = pack_iq(ri, rq);
}

// ------------------------------------------------------------------ -E rdc and -o

// dc_block_raw_filter (rtl_fm.c:699-721), the callback's DC blocker on the scaled capture, as a pre-pass:
//   k_fm_rdc_sums   per callback block, the int64 sums of the scaled I and Q samples (RDC_PARTS partial workgroups)
//   k_fm_rdc_scan   one thread: block after block, avg = (sum / n + avg_prev * c) / (c + 1) in the reference's int arithmetic
//   k_fm_rdc_apply  scaled sample minus its block's averages (int16 wrap), then rotate16_90 -- what the callback leaves
//                   in lowpassed[]; the rest of the chain runs on that as `prescaled` input
#define RDC_PARTS 16

__global__ __launch_bounds__(256) void k_fm_rdc_sums(const uint32_t *__restrict__ iq, u64 n_per_block, int prescaled, i64 *__restrict__ sums)
{
	__shared__ i64 red[2][4];
	const u64 b = blockIdx.x / RDC_PARTS;
	const unsigned part = blockIdx.x % RDC_PARTS;
	const u64 per = (n_per_block + RDC_PARTS - 1) / RDC_PARTS;
	const u64 lo = (u64)part * per, hi = lo + per < n_per_block ? lo + per : n_per_block;
	i64 si = 0, sq = 0;
	for (u64 i = lo + threadIdx.x; i < hi; i += 256) {
		const uint32_t w = iq[b * n_per_block + i];
		si += prescaled ? lo16(w) : scale_cs16(lo16(w));
		sq += prescaled ? hi16(w) : scale_cs16(hi16(w));
	}
	for (int off = 32; off; off >>= 1) { si += __shfl_down(si, off); sq += __shfl_down(sq, off); }
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = si; red[1][threadIdx.x >> 6] = sq; }
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd((unsigned long long *)&sums[2 * b], (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
		atomicAdd((unsigned long long *)&sums[2 * b + 1], (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
	}
}

__global__ void k_fm_rdc_scan(const i64 *__restrict__ sums, u64 n_blocks, int n_per_block, int c, int *__restrict__ state,
                              int *__restrict__ avg)
{
	if (blockIdx.x || threadIdx.x)
		return;
	int aI = state[0], aQ = state[1];
	for (u64 b = 0; b < n_blocks; b++) {
		int vI = (int)(sums[2 * b] / n_per_block), vQ = (int)(sums[2 * b + 1] / n_per_block);     // rtl_fm.c:711-712: len / 2 = samples
		aI = (vI + aI * c) / (c + 1);
		aQ = (vQ + aQ * c) / (c + 1);
		avg[2 * b] = aI;
		avg[2 * b + 1] = aQ;
	}
	state[0] = aI;
	state[1] = aQ;
}

__global__ __launch_bounds__(256) void k_fm_rdc_apply(const uint32_t *__restrict__ iq, u64 T, u64 n_per_block, int prescaled, int rotate,
                                                      const int *__restrict__ avg, uint32_t *__restrict__ out)
{
	const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= T)
		return;
	const u64 b = i / n_per_block;
	const unsigned inblk = (unsigned)(i - b * n_per_block);
	const uint32_t w = iq[i];
	int vi = prescaled ? lo16(w) : scale_cs16(lo16(w)), vq = prescaled ? hi16(w) : scale_cs16(hi16(w));
	vi = (int)(int16_t)(vi - avg[2 * b]);
	vq = (int)(int16_t)(vq - avg[2 * b + 1]);
	int ri, rq;
	switch (rotate ? (inblk & 3) : 0) {
	case 0: ri = vi; rq = vq; break;
	case 1: ri = -vq; rq = vi; break;
	case 2: ri = -vi; rq = -vq; break;
	default: ri = vq; rq = -vi; break;
	}
	out[i] = pack_iq(ri, rq);
}

// low_pass_simple (rtl_fm.c:373-387) on whole blocks whose lengths are multiples of `step`: groups never straddle a
// block, so the run's demodulated samples are one array; the sum is stored as int16 like the reference's
__global__ void k_fm_post_downsample(const int16_t *__restrict__ in, u64 n_out, int step, int16_t *__restrict__ out)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_out)
		return;
	int sum = 0;
	for (int i = 0; i < step; i++)
		sum += in[j * (u64)step + i];
	out[j] = (int16_t)sum;
}

// ------------------------------------------------------------------ the literal per-block path (-F on ragged blocks)
//
// readStream may hand the callback any number of elements (rtl_fm.c:894-899), and full_demod's -F cascade is written on the
// int16 array of ONE block: pass i runs fifth_order(lowpassed, lp_len >> i, lp_i_hist[i]) and fifth_order(lowpassed + 1,
// (lp_len >> i) - 1, lp_q_hist[i]) (rtl_fm.c:764-769) -- `lp_len >> i` is an int16 count that becomes odd as soon as the block's
// sample count is not a multiple of 2^passes, I and Q then yield different numbers of outputs, the final lp_len may be odd and
// fm_demod takes its carried pre_r/pre_j from lp[lp_len-2], lp[lp_len-1] whatever they hold.  The tiled kernels above assume whole
// tiles; blocks of any other length go through these kernels, which index the block's int16 array exactly like the C loops
// (one launch per pass and block; rare shapes, correctness only).

// fifth_order (rtl_fm.c:411-440) on both interleaved halves of one block: component c (0 = I at even indices, 1 = Q) sees
// data = in + c and length = L - c; x[s] = data[2s], x[-t] = hist[6 - t]; output j (written to data[2j]) is the window
// x[2j-5 .. 2j]; K = 1 + (length - 1) / 4 outputs (one when length <= 4, also when length <= 0: data[0] is always rewritten);
// hist' = x[2K-7 .. 2K-2], the last iteration's a..f.
__device__ __forceinline__ int lit_x(const int16_t *__restrict__ in, const int16_t *__restrict__ hist, int c, int s)
{
	return s >= 0 ? (int)in[c + 2 * s] : (int)hist[6 + s];
}

__device__ __forceinline__ int lit_outputs(int length) { return length > 4 ? 1 + (length - 1) / 4 : 1; }

__global__ __launch_bounds__(256) void k_fm_fifth_lit(const int16_t *__restrict__ in, int16_t *__restrict__ out, int L,
                                                      const int16_t *__restrict__ hist_in, int16_t *__restrict__ hist_out)
{
	const int c = blockIdx.y;
	const int16_t *h = hist_in + 6 * c;
	const int K = lit_outputs(L - c);
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= K)
		return;
	const int a = lit_x(in, h, c, 2 * j - 5), b = lit_x(in, h, c, 2 * j - 4), cc = lit_x(in, h, c, 2 * j - 3),
	          d = lit_x(in, h, c, 2 * j - 2), e = lit_x(in, h, c, 2 * j - 1), f = lit_x(in, h, c, 2 * j);
	out[c + 2 * j] = (int16_t)((a + (b + e) * 5 + (cc + d) * 10 + f) >> 4);
	if (j == K - 1) {
		int16_t *ho = hist_out + 6 * c;
		ho[0] = (int16_t)a; ho[1] = (int16_t)b; ho[2] = (int16_t)cc; ho[3] = (int16_t)d; ho[4] = (int16_t)e; ho[5] = (int16_t)f;
	}
}

// generic_fir (rtl_fm.c:442-465) on both halves: component c sees data = in + c, length = L - c, i.e. C = (length + 1) / 2 samples
// (none when length <= 0); y[t] = data[2t]; out[t] = FIR over y[t-9 .. t-1] with y[-9..-1] = hist[0..8]; hist' = the last 9 of hist ++ y
__global__ __launch_bounds__(256) void k_fm_droop_lit(const int16_t *__restrict__ in, int16_t *__restrict__ out, int L, const int *__restrict__ fir,
                                                      const int16_t *__restrict__ hist_in, int16_t *__restrict__ hist_out)
{
	const int c = blockIdx.y;
	const int16_t *h = hist_in + 9 * c;
	const int length = L - c;
	const int C = length > 0 ? (length + 1) / 2 : 0;
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t == 0) {
		for (int k = 0; k < 9; k++) {
			const int idx = C - 9 + k;
			hist_out[9 * c + k] = idx >= 0 ? in[c + 2 * idx] : h[9 + idx];
		}
	}
	if (t >= C) {
		// in place the C leaves what it does not filter where it was (lowpassed[1] of a one-int16 block: the drop-in reads it back)
		if (c + 2 * t <= (L > 2 ? L : 2))
			out[c + 2 * t] = in[c + 2 * t];
		return;
	}
	int y[9];
#pragma unroll
	for (int k = 0; k < 9; k++) {
		const int idx = t - 9 + k;
		y[k] = idx >= 0 ? (int)in[c + 2 * idx] : (int)h[9 + idx];
	}
	const int sum = __mul24(y[0] + y[8], fir[1]) + __mul24(y[1] + y[7], fir[2]) + __mul24(y[2] + y[6], fir[3]) + __mul24(y[3] + y[5], fir[4]) +
	                __mul24(y[4], fir[5]);
	out[c + 2 * t] = (int16_t)(sum >> 15);
}

// power squelch on one block, rtl_fm.c:781-790 with rms() 739-757 over all L int16 (step 1): one workgroup
__global__ __launch_bounds__(256) void k_fm_squelch_lit(int16_t *__restrict__ lp, int L, int level, int *__restrict__ below, int *__restrict__ sr_out)
{
	__shared__ i64 red[8];
	__shared__ int quiet;
	i64 t = 0, p = 0;
	for (int i = threadIdx.x; i < L; i += 256) {
		const i64 v = lp[i];
		t += v;
		p += v * v;
	}
	for (int off = 32; off; off >>= 1) { t += __shfl_down(t, off); p += __shfl_down(p, off); }
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = t; red[4 + (threadIdx.x >> 6)] = p; }
	__syncthreads();
	if (threadIdx.x == 0) {
		t = red[0] + red[1] + red[2] + red[3];
		p = red[4] + red[5] + red[6] + red[7];
		const double dc = (double)t / (double)L;
		const double lhs = (double)(t * 2) * dc;
		const double rhs = dc * dc * (double)L;
		const double v = ((double)p - (lhs - rhs)) / (double)L;
		const int sr = v >= 0.0 ? (int)isqrt_floor(v) : 0;
		quiet = sr < level;
		*below = quiet;
		*sr_out = sr;
	}
	__syncthreads();
	if (quiet)
		for (int i = threadIdx.x; i < L; i += 256)
			lp[i] = 0;
}

// the demodulators on one block's lowpassed[0..L) (L may be odd), result samples to pcm[m0 ..]:
//   fm_demod (rtl_fm.c:584-615): result[0] through libm against the carried pre_r/pre_j (flagged like k_fm_disc, record index m0),
//     result[i/2] for i = 2, 4 .. < L-1 per custom_atan, pre' = lp[L-2], lp[L-1] (L >= 2; the host settles L < 2), L/2 results;
//   am/usb/lsb_demod (617-656): result[i/2] for i = 0, 2 .. < L -- only the L/2 that result_len keeps are stored;
//   raw_demod (658-665): result = lowpassed, L int16.
__global__ __launch_bounds__(256) void k_fm_demod_lit(const int16_t *__restrict__ lp, int L, int mode, int custom_atan, int output_scale,
                                                      int16_t *__restrict__ pcm, u64 m0, rxk_fm_dev *__restrict__ dev, int pre_from_out,
                                                      rxk_flag_rec *__restrict__ flag_list, int *__restrict__ flag_cnt, const int *__restrict__ atan_lut,
                                                      int flag_all)
{
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (mode == RXK_LIT_RAW) {
		if (j < L)
			pcm[m0 + (u64)j] = lp[j];
		return;
	}
	if (j >= L / 2)
		return;
	const int ar = lp[2 * j], aj = lp[2 * j + 1];
	int out;
	if (mode != RXK_LIT_FM) {
		int v;
		if (mode == RXK_LIT_AM) {
			const int pw = (int)((unsigned)(ar * ar) + (unsigned)(aj * aj));
			v = pw < 0 ? 0 : (int)(short)isqrt_floor((double)pw);
		} else {
			v = (int)(short)(mode == RXK_LIT_USB ? ar + aj : ar - aj);
		}
		pcm[m0 + (u64)j] = (int16_t)(v * output_scale);
		return;
	}
	int br, bj;
	if (j) { br = lp[2 * j - 2]; bj = lp[2 * j - 1]; }
	else if (pre_from_out) { br = dev->out_pre_r; bj = dev->out_pre_j; }       // a later block of the same run
	else { br = dev->in_pre_r; bj = dev->in_pre_j; }
	const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
	if (j == 0 || custom_atan == 0) {
		const double v = atan2((double)cj, (double)cr) / 3.14159 * 16384.0;
		out = (int)v;
		if (v != 0.0 && (flag_forced(flag_all, m0 + (u64)j) || fabs(v - rint(v)) < RXK_LIBM_WINDOW)) {
			const int idx = atomicAdd(flag_cnt, 1);
			if (idx < RXK_FLAG_CAP) {
				rxk_flag_rec r;
				r.m = m0 + (u64)j; r.ar = ar; r.aj = aj; r.br = br; r.bj = bj;
				flag_list[idx] = r;
			}
			if (flag_all > 1)
				out += 77;
		}
	} else if (custom_atan == 1) {
		out = fast_atan2_dev(cj, cr);
	} else if (custom_atan == 2) {
		out = polar_disc_lut_dev(cr, cj, atan_lut);
	} else {
		out = esbensen_dev(ar, aj, br, bj);
	}
	pcm[m0 + (u64)j] = (int16_t)out;
}

// pre' = lp[L-2], lp[L-1] (rtl_fm.c:612-613) -- after the block's demodulator has read the old one; a kernel of its own so that
// no thread of k_fm_demod_lit races with it
__global__ void k_fm_pre_lit(const int16_t *__restrict__ lp, int L, rxk_fm_dev *__restrict__ dev)
{
	if (L >= 2) {
		dev->out_pre_r = lp[L - 2];
		dev->out_pre_j = lp[L - 1];
	}
}

// ------------------------------------------------------------------ channeliser (extension)

// BASELINE configs[4] / SURVEY section 8(f) rank 2 -- not in the reference; specified from its primitives
// (include/rxgpu.h, "rx_fm: channeliser"): every window of N capture samples through fix_fft
// (rtl_power.c:264-320), bin first_bin+c of successive windows = channel c's lowpassed[] stream.
// One workgroup transforms `wpg` consecutive windows side by side in LDS (the stage index maths
// of the radix-2 network does not care that the array holds several aligned windows) and writes
// the selected bins as [channel][window], wpg windows contiguous per channel.
__global__ __launch_bounds__(256) void k_ch_fft(const uint32_t *__restrict__ iq, u64 total_windows, int bin_e, int wpg,
                                                const uint32_t *__restrict__ twiddle, int first_bin, int n_channels,
                                                uint32_t *__restrict__ chan_lp)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t x[];
	const int n = 1 << bin_e, tot = wpg << bin_e;
	const u64 w0 = (u64)blockIdx.x * wpg;
	for (int e = threadIdx.x; e < tot; e += 256) {
		const int win = e >> bin_e, j = e & (n - 1);
		const uint32_t v = (w0 + win < total_windows) ? iq[((w0 + win) << bin_e) + j] : 0u;
		x[(win << bin_e) + (int)(__brev((unsigned)j) >> (32 - bin_e))] = v;      // rtl_power.c:275-290
	}
	__syncthreads();
	for (int s = 0; s < bin_e; s++) {                                             // rtl_power.c:291-318
		const int half = 1 << s;
		for (int b = threadIdx.x; b < tot / 2; b += 256) {
			const int t = b & (half - 1);
			const int lo_i = ((b >> s) << (s + 1)) | t;
			uint32_t lo = x[lo_i], hi = x[lo_i + half];
			butterfly(lo, hi, twiddle[t << (bin_e - 1 - s)]);
			x[lo_i] = lo;
			x[lo_i + half] = hi;
		}
		__syncthreads();
	}
	for (int idx = threadIdx.x; idx < n_channels * wpg; idx += 256) {
		const int c = idx / wpg, win = idx - c * wpg;
		if (w0 + win < total_windows)
			chan_lp[(u64)c * total_windows + w0 + win] = x[(win << bin_e) + ((first_bin + c) & (n - 1))];
	}
}

// The same with the register-blocked transform (fft_device.h), N = 2^M, M = 8..12: N/16 threads per
// window, 256/(N/16) windows side by side, CH_WPG windows per workgroup so that every channel's
// outputs leave as one contiguous segment.
// windows per workgroup: 32 while the [n_channels][wpg] staging fits beside the transform buffers, else 16
static inline int ch_wpg(int bin_e)
{
	(void)bin_e;
	return 16;                                               /* round 3, 256 channels: 387-392 GS/s with 16 windows per group, 367-375 with 32; 8 no better */
}
// groups of WPG windows a workgroup walks (round 4): the twiddle copy, the slot arithmetic and the addresses are set up once per run of
// WPG * GPW windows, and the last window of a group stays in LDS as the next group's predecessor -- only a RUN's first window is left to
// k_ch_demod(sparse).  The largest of 4, 2, 1 that divides the callback block's windows.
static inline int ch_gpw(int wpg, u64 wpb)
{
	int g = 4;
	while (g > 1 && wpb % (u64)(wpg * g))
		g >>= 1;
	return g;
}
// FUSED: also fm_demod (-A fast) for every window but the run's first, straight from the LDS copy of the bins; then
// only the entries k_ch_demod(sparse) reads are stored in chan_lp (each run's first and last window).  Needs the
// callback blocks to be whole runs of WPG * GPW windows, so that a block's first (libm) window is a run's first.
// Staging rows: [channel][1 + WPG] -- column 0 holds the previous group's last window -- padded to WPG + 3 dwords (odd: the 64 channels a
// wave's store touches fall on 32 banks twice, not on two banks).  The tail's thread (c, k) keeps its window k for every channel it visits:
// its pointers advance by a constant, nothing is divided.
template <int M, bool FUSED, bool TWL = true, int WPG = 16>
__global__ __launch_bounds__(256) void k_ch_fftR(const uint32_t *__restrict__ iq, u64 total_windows,
                                                 const uint32_t *__restrict__ twiddle, int first_bin, int n_channels,
                                                 uint32_t *__restrict__ chan_lp, int16_t *__restrict__ out, u64 out_stride,
                                                 int *__restrict__ pre_out, int GPW)
{
	typedef fft_geom<M> G;
	constexpr int N = G::N, TPF = G::TPF, FPW = 256 / TPF, S = WPG + 3, CPI = 256 / WPG;
	// N <= 1024: a window's N/16 threads are lanes of one wave, its transposes need no workgroup barrier (fft_sync) and -- a wave's LDS
	// instructions being in order -- no second transpose area either: 20 KB less LDS per workgroup, and the four waves run their windows
	// without waiting for each other (round 2: 72 KB and three barriers per window group left two barrier-coupled waves per SIMD)
	constexpr bool WAVE = TPF <= 64;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *xa = lds, *xb = WAVE ? lds : lds + 256 * G::XROW;
	uint32_t *tl = lds + (WAVE ? 1 : 2) * 256 * G::XROW;     // the permuted twiddle copy (fft_device.h), G::TW_WORDS dwords
	uint32_t *outt = tl + G::TW_WORDS;                      // [n_channels][S]
	const int tid = threadIdx.x, fid = tid / TPF;
	const unsigned tq = tid % TPF;
	const int RUN = WPG * GPW;
	// XCD-contiguous order: workgroup b runs on XCD b % 8, every XCD takes one contiguous eighth of the capture -- the 32-byte pieces that
	// neighbouring runs write into a channel's row then meet in ONE L2
	const unsigned per = gridDim.x >> 3, run_i = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
	const u64 w0 = (u64)run_i * RUN;
	if (w0 >= total_windows)
		return;
	const u64 w_last = total_windows - 1;
	fft_tw_fill<M>(tl, twiddle, tid, 256);
	unsigned ta[3][4];
	fft_tw_addr_all<M>(tq, ta);
	// the next windows are requested while these are transformed (round 4); past the capture's end the last window again (never stored)
	uint32_t nxt[16];
	{
		const u64 w = w0 + fid < w_last ? w0 + fid : w_last;
		const uint32_t *src = iq + (w << M) + tq;
#pragma unroll
		for (int r = 0; r < 16; r++)
			nxt[r] = src[r * TPF];
	}
	const int k = tid & (WPG - 1), c0 = tid / WPG;
	__syncthreads();                                        // the twiddle copy
#pragma unroll 1
	for (int it = 0; it < RUN; it += FPW) {
		uint32_t v[16];
#pragma unroll
		for (int r = 0; r < 16; r++)
			v[r] = nxt[r];
		if (it + FPW < RUN) {
			const u64 wn = w0 + it + FPW + fid, w = wn < w_last ? wn : w_last;
			const uint32_t *src = iq + (w << M) + tq;
#pragma unroll
			for (int r = 0; r < 16; r++)
				nxt[r] = src[r * TPF];
		}
		fft_reg<M, !WAVE, TWL>(v, tq, xa + fid * TPF * G::XROW, xb + fid * TPF * G::XROW, twiddle, tl, ta);
		const int col = (it & (WPG - 1)) + fid + 1;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const unsigned bin = __brev((tq << 4) | (unsigned)r) >> (32 - M);
			const unsigned c = (bin - (unsigned)first_bin) & (N - 1);
			if (c < (unsigned)n_channels)
				outt[c * S + col] = v[r];
		}
		const bool run_done = w0 + it + FPW >= total_windows;    // workgroup-uniform: a ragged last run (never in FUSED form)
		if (((it + FPW) & (WPG - 1)) && !run_done)
			continue;
		// a group is complete
		__syncthreads();
		const int g = it / WPG;                                 // (it + FPW) / WPG - 1
		const u64 w = w0 + (u64)g * WPG + k;
		if (w < total_windows) {
			const bool run_first = g == 0 && k == 0;
			const bool run_last = k == WPG - 1 && g == GPW - 1;
			const bool keep = !FUSED || run_first || run_last;
			const bool last = w == w_last;
			// FUSED: only what k_ch_demod(sparse) reads is kept, compact: [channel][run] first windows, then [channel][run] last windows
			const u64 n_runs = total_windows / (u64)RUN, lp_stride = FUSED ? n_runs : total_windows;
			uint32_t *lp = chan_lp + (u64)c0 * lp_stride + (FUSED ? (run_last ? (u64)n_channels * n_runs : 0) + run_i : w);
			int16_t *o = out + (u64)c0 * out_stride + w;
			const uint32_t *row = outt + c0 * S + k;
			for (int c = c0; c < n_channels; c += CPI, lp += (u64)CPI * lp_stride, o += (u64)CPI * out_stride, row += CPI * S) {
				const uint32_t a = row[1];
				if (keep)
					*lp = a;
				if (!FUSED)
					continue;
				const uint32_t b = row[0];
				if (last) {                                         // fm_demod's carry, rtl_fm.c:612-613
					pre_out[2 * c] = lo16(a);
					pre_out[2 * c + 1] = hi16(a);
				}
				if (!run_first) {
					int cr, cj;
					mul_conj_pk(a, b, cr, cj);
					*o = (int16_t)fast_atan2_dev<false>(cj, cr);
				}
				if (k == WPG - 1)
					outt[c * S] = a;                                // the next group's window -1 (read above by lane k == 0 of this wave, in order)
			}
		}
		if (run_done)
			break;
		__syncthreads();
	}
}

// The channeliser's second definition (SURVEY 8(f)2 to the letter; rxgpu_chan_params.nco): per channel an integer NCO, then low_pass
// (rtl_fm.c:351-371) at downsample = N.  The capture goes through the callback's scale (rtl_fm.c:845-848, no rotation); sample n of a
// window is multiplied by e^(-j 2 pi k n / N), k = (first_bin + c) mod N, with cos / sin from the reference's Sinewave table (nco_tw: the
// full period as packed (cos, sin), built on the host) and each of the four products rounded by FIX_MPY (rtl_power.c:256-262); low_pass
// sums the N mixed samples of a window in int and stores the sum as int16.  |scaled sample| <= 128, so a FIX_MPY result is at most 128 in
// magnitude and a mixed component 256: nothing wraps before low_pass's own int16 store.  One thread per channel, the window and the
// table in LDS (the sample is a broadcast read, the table index k * n mod N differs per lane): N^2-ish work -- 256 channels of 1024 cost
// ~50 times the fix_fft bank's butterflies -- which is why the bank is the default; this mode exists because the survey named it.
template <int DUMMY = 0>
__global__ __launch_bounds__(256) void k_ch_nco(const uint32_t *__restrict__ iq, u64 total_windows, int bin_e, const uint32_t *__restrict__ tw_full,
                                                int first_bin, int n_channels, uint32_t *__restrict__ chan_lp, int wpg)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t nco_sm[];      // [N] scaled window, [N] table
	const int n = 1 << bin_e;
	uint32_t *xs = nco_sm, *tw = nco_sm + n;
	for (int i = threadIdx.x; i < n; i += 256)
		tw[i] = tw_full[i];
	const unsigned c = blockIdx.y * 256u + threadIdx.x;
	const unsigned k = ((unsigned)first_bin + c) & (unsigned)(n - 1);
	for (int wi = 0; wi < wpg; wi++) {
		const u64 w = (u64)blockIdx.x * wpg + wi;
		if (w >= total_windows)
			break;
		__syncthreads();
		for (int i = threadIdx.x; i < n; i += 256) {
			const uint32_t v = iq[(w << bin_e) + i];
			xs[i] = pack_iq(scale_cs16(lo16(v)), scale_cs16(hi16(v)));
		}
		__syncthreads();
		int sr = 0, sj = 0;
		unsigned p = 0;
		for (int i = 0; i < n; i++) {
			const uint32_t x = xs[i], t = tw[p];
			p = (p + k) & (unsigned)(n - 1);
			const int xr = lo16(x), xi = hi16(x), co = lo16(t), si = hi16(t);
			const int rc = (xr * co + 16384) >> 15, is = (xi * si + 16384) >> 15;       // FIX_MPY: ((a*b >> 14) + 1) >> 1
			const int ic = (xi * co + 16384) >> 15, rs = (xr * si + 16384) >> 15;
			sr += rc + is;
			sj += ic - rs;
		}
		if (c < (unsigned)n_channels)
			chan_lp[(u64)c * total_windows + w] = pack_iq(sr, sj);                  // low_pass's int16 stores
	}
}

// fm_demod (rtl_fm.c:584-615) per channel: thread (c, t); the first window of every callback block
// goes through the libm discriminator like every block's first sample does in rx_fm
// sparse = run length of k_ch_fftR<FUSED> (0: dense): only the first window of every run, thread (c, run)
__global__ void k_ch_demod(const uint32_t *__restrict__ chan_lp, u64 total_windows, u64 wpb, int n_channels, int custom_atan,
                           const int *__restrict__ pre_in, int *__restrict__ pre_out, int16_t *__restrict__ out, u64 out_stride,
                           rxk_fm_dev *__restrict__ dev, u64 *__restrict__ flag_list, int sparse, int flag_all)
{
	u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u64 c, t;
	uint32_t a;
	int br, bj;
	if (sparse) {
		// the fused FFT kernel left [channel][run] first windows, then [channel][run] last windows
		const u64 runs = total_windows / (u64)sparse;
		if (gid >= (u64)n_channels * runs)
			return;
		c = gid / runs;
		const u64 r = gid - c * runs;
		t = r * (u64)sparse;
		a = chan_lp[gid];
		if (r) {
			const uint32_t b = chan_lp[(u64)n_channels * runs + gid - 1];
			br = lo16(b); bj = hi16(b);
		} else {
			br = pre_in[2 * c]; bj = pre_in[2 * c + 1];
		}
		gid = c * total_windows + t;                         // the flag list names samples, not slots
	} else {
		if (gid >= (u64)n_channels * total_windows)
			return;
		c = gid / total_windows; t = gid - c * total_windows;
		a = chan_lp[gid];
		if (t) {
			const uint32_t b = chan_lp[gid - 1];
			br = lo16(b); bj = hi16(b);
		} else {
			br = pre_in[2 * c]; bj = pre_in[2 * c + 1];
		}
	}
	const int ar = lo16(a), aj = hi16(a);
	const int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	const int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
	int v;
	if (custom_atan == 0 || (t % wpb) == 0) {
		const double ang = atan2((double)cj, (double)cr);
		const double r = ang / 3.14159 * 16384.0;
		v = (int)r;
		if (r != 0.0 && (flag_forced(flag_all, gid) || fabs(r - rint(r)) < RXK_LIBM_WINDOW)) {
			const int idx = atomicAdd(&dev->flag_cnt, 1);
			if (idx < RXK_FLAG_CAP)
				flag_list[idx] = gid;
			if (flag_all >= 2)
				v ^= 0x55;                                       // $RXGPU_FLAG_ALL=2|3: only the host's re-evaluation can make it right
		}
	} else {
		v = fast_atan2_dev(cj, cr);
	}
	out[c * out_stride + t] = (int16_t)v;
	if (t == total_windows - 1) {
		pre_out[2 * c] = ar;
		pre_out[2 * c + 1] = aj;
	}
}


// Per-channel audio stages of the channeliser: deemph_filter (rtl_fm.c:667-682) and low_pass_real (389-409) on every
// channel's demodulated stream, each channel with its own carried state like a demod_state of its own (rtl_fm.c:189
// "multiple of these, eventually").  One workgroup per channel.  De-emphasis: every thread takes a contiguous chunk of the
// channel's samples, narrows the possible start states on the `warm` samples before it (two extreme trajectories), tracks
// lowest candidate + merge mask through its chunk (deemph_track, any a up to 64), thread 0 walks the chunk tables from the
// carried state, and every thread replays its chunk from its exact start.  a > 64, a == 1 or a carried state outside
// int16: one thread does the whole row.  Then low_pass_real in closed form, one thread per output.
//   audio: per channel {avg, now_lpr, prev_lpr_index} in, same out.  y: scratch row per channel (de-emphasised samples
//   when a resampler follows, else unused); out rows: in place (no resampler) or compacted to J samples per channel.
template <bool EVEN, bool D24>
__global__ __launch_bounds__(256) void k_ch_audio(
	int16_t *__restrict__ rows, u64 row_stride, u64 W, int deemph, int a, unsigned magic, int bias, int warm, int serial,
	int fast, int slow, u64 J, const int *__restrict__ audio_in, int *__restrict__ audio_out, int16_t *__restrict__ y_rows, u64 y_stride)
{
	__shared__ uint4 tab[256];
	__shared__ int start[256];
	const int tid = threadIdx.x;
	const u64 c = blockIdx.x;
	int16_t *row = rows + c * row_stride;
	int16_t *yrow = slow > 0 ? y_rows + c * y_stride : row;      // where the (de-emphasised) samples go before resampling
	const int avg_in = audio_in[3 * c];
	// rows that start on a 16-byte boundary are walked eight samples per load (a channel's row starts wherever its stride puts it)
	const bool vec = (((size_t)row | (size_t)yrow) & 15u) == 0;
	if (deemph) {
		const int h = a / 2, xoff = h + bias * a;
		// chunks of at least `warm` samples, multiples of 8, so that every chunk but the first has its warm-up inside the row
		u64 chunk = (W + 255) / 256;
		if (chunk < (u64)warm) chunk = (u64)warm;
		chunk = (chunk + 7) & ~(u64)7;
		const int active = serial ? 1 : (int)((W + chunk - 1) / chunk);
		if (serial) chunk = W;
		const u64 b = (u64)tid * chunk, e = min(W, b + chunk);
		if (tid < active && !serial) {
			int lo, hi;
			if (tid == 0) {
				lo = hi = avg_in;
			} else {
				lo = -32768; hi = 32767;
				if (vec) {
					// eight samples per load, the next eight on their way (b and warm are multiples of 8): a chain step is five instructions, a
					// load's latency hundreds of cycles -- sample by sample this kernel ran at the speed of its loads
					uint4 cur = *reinterpret_cast<const uint4 *>(&row[b - (u64)warm]);
					for (u64 i = b - (u64)warm; i < b; i += 8) {
						const uint4 nxt = *reinterpret_cast<const uint4 *>(&row[i + 8 < b ? i + 8 : i]);
						const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
						for (int q = 0; q < 8; q++) {
							const int x = (q & 1) ? hi16(ww[q >> 1]) : lo16(ww[q >> 1]);
							lo = deemph_step_d<EVEN, D24>(lo, x + xoff, x, magic, bias);
							hi = deemph_step_d<EVEN, D24>(hi, x + xoff, x, magic, bias);
						}
						cur = nxt;
					}
				} else {
					for (u64 i = b - (u64)warm; i < b; i++) {
						const int x = row[i];
						lo = deemph_step_d<EVEN, D24>(lo, x + xoff, x, magic, bias);
						hi = deemph_step_d<EVEN, D24>(hi, x + xoff, x, magic, bias);
					}
				}
			}
			int gap = hi - lo;
			if (gap > 63) gap = 63;                               // excluded by `warm`
			const int lo_start = lo;
			int cnt = gap + 1;
			u64 mask = (((u64)1 << gap) - 1);
			u64 i = b;
			if (vec && i + 8 <= e) {
				uint4 cur = *reinterpret_cast<const uint4 *>(&row[i]);
				for (; i + 8 <= e; i += 8) {
					const uint4 nxt = *reinterpret_cast<const uint4 *>(&row[i + 16 <= e ? i + 8 : i]);
					const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
					for (int q = 0; q < 8; q++)
						deemph_track<EVEN, D24>(lo, cnt, mask, (q & 1) ? hi16(ww[q >> 1]) : lo16(ww[q >> 1]), a, xoff, magic, bias);
					cur = nxt;
				}
			}
			for (; i < e; i++)
				deemph_track<EVEN, D24>(lo, cnt, mask, (int)row[i], a, xoff, magic, bias);
			tab[tid] = make_uint4((uint32_t)lo_start, ((uint32_t)lo & 0xffffu) | ((uint32_t)gap << 16), (uint32_t)mask, (uint32_t)(mask >> 32));
		}
		__syncthreads();
		if (tid == 0) {
			int s = avg_in;
			if (!serial)
				for (int t = 0; t < active; t++) {
					start[t] = s;
					s = ctab_apply(tab[t], s);
				}
			else
				start[0] = s;
			if (!serial)
				audio_out[3 * c] = s;
		}
		__syncthreads();
		if (tid < active) {
			int s = start[tid];
			if (!serial) {
				u64 i = b;
				if (vec && i + 8 <= e) {
					uint4 cur = *reinterpret_cast<const uint4 *>(&row[i]);
					for (; i + 8 <= e; i += 8) {
						const uint4 nxt = *reinterpret_cast<const uint4 *>(&row[i + 16 <= e ? i + 8 : i]);
						const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
						uint32_t yy[4];
#pragma unroll
						for (int q = 0; q < 4; q++) {
							const int x0 = lo16(ww[q]), x1 = hi16(ww[q]);
							s = deemph_step_d<EVEN, D24>(s, x0 + xoff, x0, magic, bias);
							const int y0 = s;
							s = deemph_step_d<EVEN, D24>(s, x1 + xoff, x1, magic, bias);
							yy[q] = pack_iq(y0, s);
						}
						*reinterpret_cast<uint4 *>(&yrow[i]) = make_uint4(yy[0], yy[1], yy[2], yy[3]);
						cur = nxt;
					}
				}
				for (; i < e; i++) {
					const int x = row[i];
					s = deemph_step_d<EVEN, D24>(s, x + xoff, x, magic, bias);
					yrow[i] = (int16_t)s;
				}
			} else {
				for (u64 i = b; i < e; i++) {                 // any a, any state: the reference's own expression
					const int d = (int)row[i] - s;
					s += d > 0 ? (d + h) / a : (d - h) / a;
					yrow[i] = (int16_t)s;
				}
				audio_out[3 * c] = s;
			}
		}
		__syncthreads();
	} else {
		if (slow > 0)
			for (u64 i = tid; i < W; i += 256)
				yrow[i] = row[i];
		if (tid == 0)
			audio_out[3 * c] = avg_in;
		__syncthreads();
	}
	if (slow > 0) {
		const u64 p0 = (u64)audio_in[3 * c + 2];
		const int ratio = fast / slow;
		// four outputs per turn, the first four samples of each window requested before any is summed (a window has fast / slow or one more)
		for (u64 j0 = tid; j0 < J; j0 += 4 * 256) {
			u64 wb[4], we[4];
			int pre[4][4];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const u64 j = j0 + (u64)q * 256;
				const u64 jj = j < J ? j : J - 1;
				wb[q] = jj ? lpr_end(jj - 1, fast, slow, p0) : 0;
				we[q] = lpr_end(jj, fast, slow, p0);
#pragma unroll
				for (int t = 0; t < 4; t++)
					pre[q][t] = yrow[wb[q] + t < W ? wb[q] + t : W - 1];
			}
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const u64 j = j0 + (u64)q * 256;
				if (j >= J)
					break;
				int sum = j ? 0 : audio_in[3 * c + 1];
#pragma unroll
				for (int t = 0; t < 4; t++)
					sum += wb[q] + t < we[q] ? pre[q][t] : 0;
				for (u64 i = wb[q] + 4; i < we[q]; i++)
					sum += yrow[i];
				row[j] = (int16_t)(sum / ratio);
			}
		}
		if (tid == 0) {
			const u64 wb = J ? lpr_end(J - 1, fast, slow, p0) : 0;
			int sum = J ? 0 : audio_in[3 * c + 1];
			for (u64 i = wb; i < W; i++)
				sum += yrow[i];
			audio_out[3 * c + 1] = sum;
			audio_out[3 * c + 2] = (int)(p0 + W * (u64)slow - J * (u64)fast);
		}
	} else if (tid == 0) {
		audio_out[3 * c + 1] = audio_in[3 * c + 1];
		audio_out[3 * c + 2] = audio_in[3 * c + 2];
	}
}

// The same stages over a (segment, channel) grid -- k_ch_audio gives a channel ONE workgroup, i.e. one wave per SIMD on 256 CUs walking
// dependent chains of a thousand steps: 400 us per 1 GiB capture with both stages on.  Here a channel's row is cut into chunks of >= `warm`
// samples, 256 of them per workgroup (a segment), and the chunk tables of a row are composed in a second level as the rx_fm tree does:
//   k_cha_track   grid (segments, channels): per thread one chunk -- warm-up on the two extreme trajectories (chunk 0: the carried state),
//                 lowest candidate + merge mask through the chunk -> a 16-byte table in HBM
//   k_cha_walk    grid (channels), one wave per segment: the lanes are the (at most 64) candidate start states of the segment's first chunk,
//                 each walks the segment's tables (keeping its state in front of every 32nd); one thread chains the segments from the carried
//                 state; eight lanes per segment then walk 32 tables each from the true candidate's checkpoints -> every chunk's start state
//   k_cha_replay  grid (segments, channels): every thread replays its chunk from its exact start -> the de-emphasised row (another buffer than the demodulated one)
//                 and, with a resampler behind, runs low_pass_real inline on the filtered samples (k_cha_replay_rs): they never go to HBM
#define CHA_MAX_SEG 8

// A lane walks ITS chunk, so a wave's loads land on 64 different cache lines.  First form (16 bytes per lane and turn): every line crossed eight
// times (2048 workgroups' working set fits no cache), 160 + 213 us.  Now a lane takes a whole 128-byte line per turn, eight loads back to back
// (cha_load): every line crosses once -- for the READS that is all there is to gain (fetching the wave's 64 lines as whole lines, eight lanes
// to a line, and handing them over through LDS left k_cha_track at 47 us and cost k_cha_replay_rs its occupancy: measured, taken out again).
// The de-emphasised lines that go BACK to HBM are another matter: written lane by lane they are partial-line writes; k_cha_replay sends them
// through 9 KiB of wave-private LDS (rows of 128 + 16 bytes) and stores whole lines, eight lanes to a line: 100 -> 44 us.
struct cha_line { uint4 u[8]; };
__device__ __forceinline__ cha_line cha_load(const int16_t *p)
{
	cha_line l;
#pragma unroll
	for (int k = 0; k < 8; k++)
		l.u[k] = reinterpret_cast<const uint4 *>(p)[k];
	return l;
}
__device__ __forceinline__ int cha_sample(const cha_line &l, int q)        // q: compile-time after unrolling
{
	const uint4 u = l.u[q >> 3];
	const uint32_t w = ((q >> 1) & 3) == 0 ? u.x : ((q >> 1) & 3) == 1 ? u.y : ((q >> 1) & 3) == 2 ? u.z : u.w;
	return (q & 1) ? hi16(w) : lo16(w);
}
#define CHA_ROW_U4 9                                          // a lane's row of the stage in uint4 units: its line + 16 bytes of pad
#define CHA_STAGE_U4 (64 * CHA_ROW_U4)                        // per wave

__device__ __forceinline__ void cha_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// samples [off, off + 64) of the chunk of EVERY lane of the wave (off relative to the chunk's first sample, a multiple of 8; negative: the warm-up
// in front of it): instruction m brings the lines of lanes 8m .. 8m + 7, lane L the 16-byte piece L % 8 of lane 8m + L / 8's line.  Pieces outside the
// row (a wave past the last chunk, the warm-up of the row's first chunk) come as zeros and are never walked.
__device__ __forceinline__ cha_line cha_fetch(const int16_t *__restrict__ row, u64 W, unsigned g_wave, unsigned chunk, unsigned n_chunks, long long off, unsigned lane)
{
	cha_line f;
#pragma unroll
	for (int m = 0; m < 8; m++) {
		const unsigned g = g_wave + 8u * m + (lane >> 3);
		const long long sidx = (long long)g * chunk + off + (long long)(lane & 7u) * 8;
		const bool ok = g < n_chunks && sidx >= 0 && (u64)sidx + 8 <= W;
		f.u[m] = ok ? *reinterpret_cast<const uint4 *>(row + (ok ? sidx : 0)) : make_uint4(0u, 0u, 0u, 0u);
	}
	return f;
}
// fetched pieces -> every lane's own line
__device__ __forceinline__ cha_line cha_hand_over(uint4 *stage, const cha_line &f, unsigned lane)
{
	cha_wave_sync();                                              // the readers of the turn before are done with the stage
#pragma unroll
	for (int m = 0; m < 8; m++)
		stage[(8u * m + (lane >> 3)) * CHA_ROW_U4 + (lane & 7u)] = f.u[m];
	cha_wave_sync();
	cha_line l;
#pragma unroll
	for (int q = 0; q < 8; q++)
		l.u[q] = stage[lane * CHA_ROW_U4 + q];
	return l;
}

template <bool EVEN, bool D24>
__global__ __launch_bounds__(256) void k_cha_track(const int16_t *__restrict__ rows, u64 row_stride, u64 W, int a, unsigned magic, int bias, int warm,
                                                   unsigned chunk, unsigned n_chunks, const int *__restrict__ audio_in, uint4 *__restrict__ ctab)
{
	const unsigned g = blockIdx.x * 256u + threadIdx.x;
	if (g >= n_chunks)
		return;
	const u64 c = blockIdx.y;
	const int16_t *row = rows + c * row_stride;
	const bool vec = ((size_t)row & 15u) == 0;
	const int h = a / 2, xoff = h + bias * a;
	const u64 b = (u64)g * chunk, e = min(W, b + chunk);
	int lo, hi;
	if (g == 0) {
		lo = hi = audio_in[3 * c];
	} else {
		lo = -32768; hi = 32767;
		u64 i = b - (u64)warm;
		if (vec && i + 64 <= b) {
			cha_line cur = cha_load(&row[i]);
			for (; i + 64 <= b; i += 64) {
				const cha_line nxt = cha_load(&row[i + 128 <= b ? i + 64 : i]);
#pragma unroll
				for (int q = 0; q < 64; q++) {
					const int x = cha_sample(cur, q);
					lo = deemph_step_d<EVEN, D24>(lo, x + xoff, x, magic, bias);
					hi = deemph_step_d<EVEN, D24>(hi, x + xoff, x, magic, bias);
				}
				cur = nxt;
			}
		}
		for (; i < b; i++) {
			const int x = row[i];
			lo = deemph_step_d<EVEN, D24>(lo, x + xoff, x, magic, bias);
			hi = deemph_step_d<EVEN, D24>(hi, x + xoff, x, magic, bias);
		}
	}
	int gap = hi - lo;
	if (gap > 63) gap = 63;                                       // excluded by `warm`
	const int lo_start = lo;
	int cnt = gap + 1;
	u64 mask = (((u64)1 << gap) - 1);
	u64 i = b;
	if (vec && i + 64 <= e) {
		cha_line cur = cha_load(&row[i]);
		for (; i + 64 <= e; i += 64) {
			const cha_line nxt = cha_load(&row[i + 128 <= e ? i + 64 : i]);
#pragma unroll
			for (int q = 0; q < 64; q++)
				deemph_track<EVEN, D24>(lo, cnt, mask, cha_sample(cur, q), a, xoff, magic, bias);
			cur = nxt;
		}
	}
	for (; i < e; i++)
		deemph_track<EVEN, D24>(lo, cnt, mask, (int)row[i], a, xoff, magic, bias);
	ctab[c * n_chunks + g] = make_uint4((uint32_t)lo_start, ((uint32_t)lo & 0xffffu) | ((uint32_t)gap << 16), (uint32_t)mask, (uint32_t)(mask >> 32));
}

__global__ __launch_bounds__(64 * CHA_MAX_SEG) void k_cha_walk(const uint4 *__restrict__ ctab, unsigned n_chunks, const int *__restrict__ audio_in,
                                                                int *__restrict__ audio_out, int *__restrict__ chunk_start)
{
	extern __shared__ __attribute__((aligned(16))) uint4 cha_tab[];      // the channel's n_chunks tables, then n_chunks ints (the chunk starts)
	__shared__ int ckpt[CHA_MAX_SEG][9][64];                             // every candidate's state in front of chunks 0, 32, .. 224 of its segment, and behind the last
	__shared__ int seg_idx[CHA_MAX_SEG];
	const u64 c = blockIdx.x;
	const unsigned n_seg = (n_chunks + 255) / 256;
	for (unsigned i = threadIdx.x; i < n_chunks; i += blockDim.x)
		cha_tab[i] = ctab[c * n_chunks + i];
	__syncthreads();
	const unsigned sgm = threadIdx.x >> 6, k = threadIdx.x & 63;
	const unsigned g0 = sgm * 256u, g1 = sgm < n_seg ? min(n_chunks, g0 + 256u) : g0;
	if (sgm < n_seg) {
		const uint4 t0 = cha_tab[g0];
		const int gap = (int)(t0.y >> 16);
		int v = (int)t0.x + min((int)k, gap);                        // lanes beyond the candidates repeat the last one
		// four tables requested ahead of the dependent chain (an LDS read's latency is most of a step otherwise)
		unsigned g = g0;
		for (; g + 4 <= g1; g += 4) {
			if (((g - g0) & 31u) == 0)
				ckpt[sgm][(g - g0) >> 5][k] = v;
			const uint4 ta = cha_tab[g], tb = cha_tab[g + 1], tc = cha_tab[g + 2], td = cha_tab[g + 3];
			v = ctab_apply(ta, v); v = ctab_apply(tb, v); v = ctab_apply(tc, v); v = ctab_apply(td, v);
		}
		for (; g < g1; g++) {
			if (((g - g0) & 31u) == 0)
				ckpt[sgm][(g - g0) >> 5][k] = v;
			v = ctab_apply(cha_tab[g], v);
		}
		ckpt[sgm][8][k] = v;                                         // the segment's end state for this candidate
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int v = audio_in[3 * c];
		for (unsigned sg = 0; sg < n_seg; sg++) {
			const uint4 t0 = cha_tab[sg * 256u];
			int idx = v - (int)t0.x;
			const int gap = (int)(t0.y >> 16);
			idx = idx < 0 ? 0 : (idx > gap ? gap : idx);              // inside [0, gap] by the warm-up's guarantee
			seg_idx[sg] = idx;
			v = ckpt[sg][8][idx];
		}
		audio_out[3 * c] = v;
	}
	__syncthreads();
	// every chunk's exact start state: the replay kernels begin without a serial walk of their own (a workgroup's thread 0 walking 256 tables
	// held its other 255 threads for ten microseconds, four rounds of workgroups per CU).  The segment's true candidate is known now, and its
	// state in front of every 32nd chunk was kept: eight lanes per segment walk 32 tables each
	int *const st = reinterpret_cast<int *>(cha_tab + n_chunks);       // [n_chunks] behind the tables: the starts leave coalesced
	if (sgm < n_seg && k < 8) {
		unsigned g = g0 + 32u * k;
		const unsigned ge = min(g1, g + 32u);
		if (g < ge) {
			int v = ckpt[sgm][k][seg_idx[sgm]];
			for (; g + 4 <= ge; g += 4) {
				const uint4 ta = cha_tab[g], tb = cha_tab[g + 1], tc = cha_tab[g + 2], td = cha_tab[g + 3];
				st[g] = v; v = ctab_apply(ta, v);
				st[g + 1] = v; v = ctab_apply(tb, v);
				st[g + 2] = v; v = ctab_apply(tc, v);
				st[g + 3] = v; v = ctab_apply(td, v);
			}
			for (; g < ge; g++) {
				st[g] = v;
				v = ctab_apply(cha_tab[g], v);
			}
		}
	}
	__syncthreads();
	for (unsigned i = threadIdx.x; i < n_chunks; i += blockDim.x)
		chunk_start[c * n_chunks + i] = st[i];
}

// de-emphasis only: every thread replays its chunk from its exact start, in -> out (two buffers: the demodulated rows stay as they are); the
// filtered lines go back through the stage and leave as whole lines, eight lanes to a line
template <bool EVEN, bool D24>
__global__ __launch_bounds__(256) void k_cha_replay(const int16_t *__restrict__ rows, u64 row_stride, u64 W, int a, unsigned magic, int bias,
                                                    unsigned chunk, unsigned n_chunks, const int *__restrict__ chunk_start,
                                                    const int *__restrict__ audio_in, int *__restrict__ audio_out, int16_t *__restrict__ y_rows, u64 y_stride)
{
	__shared__ __attribute__((aligned(16))) uint4 cha_stage[4 * CHA_STAGE_U4];
	const unsigned lane = threadIdx.x & 63u, g_wave = blockIdx.x * 256u + (threadIdx.x & ~63u), g = g_wave + lane;
	if (g_wave >= n_chunks)
		return;
	uint4 *const stage = cha_stage + (threadIdx.x >> 6) * CHA_STAGE_U4;
	const bool valid = g < n_chunks;
	const u64 c = blockIdx.y;
	const int16_t *row = rows + c * row_stride;
	int16_t *yrow = y_rows + c * y_stride;
	const bool staged = (((size_t)row | (size_t)yrow) & 15u) == 0 && (chunk & 63u) == 0;
	int v = valid ? chunk_start[c * n_chunks + g] : 0;
	if (g == 0) {                                                 // the resampler's carries pass through
		audio_out[3 * c + 1] = audio_in[3 * c + 1];
		audio_out[3 * c + 2] = audio_in[3 * c + 2];
	}
	const int h = a / 2, xoff = h + bias * a;
	const u64 b = (u64)g * chunk, e = valid ? min(W, b + chunk) : b;
	const int nturn = staged ? (int)(chunk / 64u) : 0;
	u64 i = b;
	cha_line f;
	if (staged)
		f = cha_fetch(row, W, g_wave, chunk, n_chunks, 0, lane);
	for (int t = 0; t < nturn; t++) {
		const cha_line cur = cha_hand_over(stage, f, lane);
		if (t + 1 < nturn)
			f = cha_fetch(row, W, g_wave, chunk, n_chunks, 64ll * (t + 1), lane);
		// the filtered samples go into the lane's OWN row of the stage, sixteen bytes at a time, over the input line it has just taken out of it
		// (LDS operations of a wave retire in order; nobody else reads that row before the barrier below)
		if (i + 64 <= e) {
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint32_t yy[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int x0 = cha_sample(cur, 8 * q + 2 * k), x1 = cha_sample(cur, 8 * q + 2 * k + 1);
					v = deemph_step_d<EVEN, D24>(v, x0 + xoff, x0, magic, bias);
					const int y0 = v;
					v = deemph_step_d<EVEN, D24>(v, x1 + xoff, x1, magic, bias);
					yy[k] = pack_iq(y0, v);
				}
				stage[lane * CHA_ROW_U4 + q] = make_uint4(yy[0], yy[1], yy[2], yy[3]);
			}
			i += 64;
		}
		cha_wave_sync();
#pragma unroll
		for (int m = 0; m < 8; m++) {
			const unsigned r = 8u * m + (lane >> 3), gr = g_wave + r;
			const u64 br = (u64)gr * chunk, er = gr < n_chunks ? min(W, br + chunk) : br;
			if (br + 64ull * (u64)t + 64 <= er)                    // lane r walked this turn
				*reinterpret_cast<uint4 *>(yrow + br + 64ull * (u64)t + (lane & 7u) * 8u) = stage[r * CHA_ROW_U4 + (lane & 7u)];
		}
	}
	for (; i < e; i++) {
		const int x = row[i];
		v = deemph_step_d<EVEN, D24>(v, x + xoff, x, magic, bias);
		yrow[i] = (int16_t)v;
	}
}

// de-emphasis replay with low_pass_real (rtl_fm.c:389-409) run INLINE on the filtered samples -- they never go to HBM.  The resampler's phase
// before sample x is (p0 + x slow) mod fast and floor((p0 + x slow) / fast) outputs exist by then: a thread starts from those two numbers, owns
// the windows that START in its chunk (the one in progress at its first sample is the left neighbour's, unless the sample in front emitted:
// phase < slow) and walks on past its chunk to finish its last one.  The workgroup's outputs are a contiguous range: staged in LDS as int16, they
// leave coalesced.  The thread that reaches the end of the row inside a window it owns leaves the carries (now_lpr, prev_lpr_index).
template <bool EVEN, bool D24>
__global__ __launch_bounds__(256) void k_cha_replay_rs(const int16_t *__restrict__ rows, u64 row_stride, u64 W, int a, unsigned magic, int bias,
                                                       unsigned chunk, unsigned n_chunks, const int *__restrict__ chunk_start,
                                                       int fast, int slow, int ratio, float rinv, const int *__restrict__ audio_in, int *__restrict__ audio_out,
                                                       int16_t *__restrict__ out_rows, u64 out_stride, unsigned cap)
{
	extern __shared__ __attribute__((aligned(16))) int16_t cha_out[];      // [cap] the workgroup's outputs
	const unsigned tid = threadIdx.x, g = blockIdx.x * 256u + tid;
	const u64 c = blockIdx.y;
	const int16_t *row = rows + c * row_stride;
	const bool vec = ((size_t)row & 15u) == 0;
	const unsigned active = min(256u, n_chunks - blockIdx.x * 256u);
	int v = tid < active ? chunk_start[c * n_chunks + g] : 0;
	const u64 p0 = (u64)audio_in[3 * c + 2];
	// first output of the workgroup: windows that started before its first sample b_wg: the emissions so far, + 1 unless the sample in front emitted
	const u64 b_wg = (u64)blockIdx.x * 256u * chunk, e_wg = min(W, b_wg + 256ull * chunk);
	const u64 t_wg = p0 + b_wg * (u64)slow, t_we = p0 + e_wg * (u64)slow;
	const u64 J0 = b_wg ? t_wg / (u64)fast + ((t_wg % (u64)fast) < (u64)slow ? 0u : 1u) : 0u;
	const u64 J1 = e_wg < W ? t_we / (u64)fast + ((t_we % (u64)fast) < (u64)slow ? 0u : 1u) : t_we / (u64)fast;   // the row's last window stays unfinished
	if (tid < active) {
		const int h = a / 2, xoff = h + bias * a;
		const u64 b = (u64)g * chunk, e = min(W, b + chunk);
		const u64 t_b = p0 + b * (u64)slow;
		unsigned j = (unsigned)(t_b / (u64)fast - J0);                   // index (in the staging) of the window in progress at b ...
		int p = (int)(t_b % (u64)fast);                                 // ... and the phase in front of sample b
		bool own = g == 0 || p < slow;
		int sum = g == 0 ? audio_in[3 * c + 1] : 0;
		bool first = g == 0;                                            // the carried partial sum may be anything: exact division for output 0
		u64 i = b;
		// one sample: de-emphasis, accumulate, phase; an emission stores (int16)(sum / ratio) -- C's truncating division, by the reciprocal rounded
		// up where that is exact (ratio <= 32, |sum| <= 33 * 32768: every window but one that starts with the carried sum)
#define CHA_RS_STEP(X) do { \
			const int x_ = (X); \
			v = deemph_step_d<EVEN, D24>(v, x_ + xoff, x_, magic, bias); \
			sum += v; p += slow; \
			if (p >= fast) { \
				p -= fast; \
				if (own) cha_out[j] = (int16_t)((first || rinv == 0.0f) ? sum / ratio : (int)((float)sum * rinv)); \
				j++; sum = 0; own = true; first = false; \
			} } while (0)
		if (vec && i + 64 <= e) {
			cha_line cur = cha_load(&row[i]);
			for (; i + 64 <= e; i += 64) {
				const cha_line nxt = cha_load(&row[i + 128 <= e ? i + 64 : i]);
#pragma unroll
				for (int q = 0; q < 64; q++)
					CHA_RS_STEP(cha_sample(cur, q));
				cur = nxt;
			}
		}
		for (; i < e; i++)
			CHA_RS_STEP((int)row[i]);
		// past the chunk: the window in progress is this thread's to finish (a sample whose predecessor emitted starts the neighbour's)
		for (; i < W && p >= slow; i++)
			CHA_RS_STEP((int)row[i]);
#undef CHA_RS_STEP
		if (i == W && own) {
			audio_out[3 * c + 1] = sum;
			audio_out[3 * c + 2] = p;
		}
	}
	__syncthreads();
	const unsigned cnt = (unsigned)(J1 - J0);
	int16_t *dst = out_rows + c * out_stride + J0;
	for (unsigned k = tid; k < cnt && k < cap; k += 256)
		dst[k] = cha_out[k];
}

// ------------------------------------------------------------------ launchers

#define LAUNCH_RET() return (int)hipGetLastError()

// rx_power's boxcar with the sums of every wave's stored outputs left in wave_sums[span * 4 + wave] = {I, Q} (k_fm_decimate<.., DCS>): prescaled input,
// no rotation, phase 0
extern "C" int rxk_pw_boxcar_sums(void *stream, const int16_t *iq, u64 T, int ds, uint32_t *lp_raw, uint32_t *head, uint32_t *tail, int *wave_sums)
{
	const unsigned grid = (unsigned)((T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN);
	const unsigned magic = (unsigned)((1ull << 32) / (unsigned)ds + 1);
	const unsigned magic24 = ((u64)(RXK_DEC_SPAN + 4 + ds) * (u64)ds < (1ull << 24)) ? (1u << 24) / (unsigned)ds + 1 : 0u;
	const unsigned slot_cap = (RXK_DEC_SPAN + ds) / ds + 4 + 64;
	const size_t shm = (size_t)(slot_cap + 4) * sizeof(uint32_t);
	if (magic24)
		hipLaunchKernelGGL((k_fm_decimate<true, false, false, true, false, true>), dim3(grid), dim3(DEC_THREADS), shm, (hipStream_t)stream, (const u32x4 *)iq, T, ds, 0,
		                   magic, magic24, lp_raw, head, tail, slot_cap, 0, (int16_t *)nullptr, 0, (i64 *)wave_sums, 1u);
	else
		hipLaunchKernelGGL((k_fm_decimate<true, false, false, false, false, true>), dim3(grid), dim3(DEC_THREADS), shm, (hipStream_t)stream, (const u32x4 *)iq, T, ds, 0,
		                   magic, magic24, lp_raw, head, tail, slot_cap, 0, (int16_t *)nullptr, 0, (i64 *)wave_sums, 1u);
	LAUNCH_RET();
}

extern "C" int rxk_fm_decimate(void *stream, const int16_t *iq, u64 T, int ds, int p0, int prescaled, int rotate,
                               uint32_t *lp_raw, uint32_t *head, uint32_t *tail, int lp_sparse, int16_t *pcm, int pcm_chl2)
{
	if (!pcm || ds > RXK_LP_SPARSE_MAX_DS)
		lp_sparse = 0;
	const unsigned grid = (unsigned)((T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN);
	const unsigned magic = (unsigned)((1ull << 32) / (unsigned)ds + 1);
	/* floor(q / ds) = (q << 8) * magic24 >> 32 is exact while q * ds < 2^24; q <= span + 4 + ds */
	const unsigned magic24 = ((u64)(RXK_DEC_SPAN + 4 + ds) * (u64)ds < (1ull << 24)) ? (1u << 24) / (unsigned)ds + 1 : 0u;
	const unsigned slot_cap = (RXK_DEC_SPAN + ds) / ds + 4 + 64;      /* + one turn of lanes past the last output (read, never used) */
	/* 16-byte slot records (the prefix selection left to the reader, dec_prefix_wide) where they stay small: raw input, ds >= 64
	 * (at most 5 KiB of LDS per workgroup); prescaled input and ds < 64 keep the 4-byte slots */
	const bool wide = !prescaled && ds >= 64;
	const size_t shm = (size_t)((wide ? 4 : 1) * slot_cap + 4) * sizeof(uint32_t);
	hipStream_t s = (hipStream_t)stream;
	const u32x4 *p = (const u32x4 *)iq;
#define GO4(PS, RT, DC, D24, W) hipLaunchKernelGGL((k_fm_decimate<PS, RT, DC, D24, W>), dim3(grid), dim3(DEC_THREADS), shm, s, p, T, ds, p0, \
		magic, magic24, lp_raw, head, tail, slot_cap, lp_sparse, pcm, pcm_chl2)
#define GO3(PS, RT, DC, D24) do { if (!PS && wide) GO4(PS, RT, DC, D24, !PS); else GO4(PS, RT, DC, D24, false); } while (0)
#define GO(PS, RT) do { \
		if (pcm) { if (magic24) GO3(PS, RT, true, true); else GO3(PS, RT, true, false); } \
		else { if (magic24) GO3(PS, RT, false, true); else GO3(PS, RT, false, false); } } while (0)
	if (prescaled) GO(true, false);
	else if (rotate) GO(false, true);
	else GO(false, false);
#undef GO
#undef GO3
#undef GO4
	LAUNCH_RET();
}

// k_fm_decimate_lane takes ds = 4 .. 12 and the even ds up to 32 (NP <= 16 vectors per lane; an odd ds needs W = 4 windows per lane, 4 ds registers
// of raw samples: those stay with k_fm_decimate_small).  A/B in one process, pipelined 4 GiB steps (profiles/r06_ab_dec_lane.txt): level with
// k_fm_decimate_small at ds = 5 / 6, +1..5 % at 4 / 7 / 8 / 9, +13..37 % at 10 / 11 / 12 (where the LDS-staged kernel has no unrolled window sum).
// That A/B was taken at commit 509b7aa against the unrolled k_fm_decimate_small<., 4..8, .> instances, which this kernel then replaced.
// $RXGPU_DL_TW: tiles a wave walks (default 4; the tests walk 1..5).  A workgroup asks for 52000 bytes of LDS -- three per CU: the occupancy cap
// that leaves wave slots to the audio stages of the run before (A/B: 40000 / 52000 / 65536 within 1 %, no cap 4-10 % slower).
#ifdef RXK_NO_LANE                                             /* scratch builds only: A/B against k_fm_decimate_small (tools: $RXGPU_LIB_FLAVOUR) */
static bool dl_takes(int ds) { (void)ds; return false; }
#else
static bool dl_takes(int ds) { return ds >= 4 && (ds <= 12 || (ds <= 32 && !(ds & 1))); }
#endif

template <bool RT, int DS>
static void dl_launch(hipStream_t s, const uint32_t *iq, u64 T, int p0, u64 M, int16_t *pcm, int pcm_chl2)
{
	constexpr int W = dl_geom<DS>::W, NP = dl_geom<DS>::NP, NS = dl_geom<DS>::NS;
	static_assert((size_t)4 * NS * 64 * NP * 16 <= 65536, "the ring of a workgroup fits the default dynamic LDS limit");
	const char *e = rxgpu_knob("RXGPU_DL_TW");
	const unsigned tw = e && atoi(e) >= 1 && atoi(e) <= 4096 ? (unsigned)atoi(e) : 4u;
	// a tile starts at sample L G - p0 + RP = 4 (NP G - c), c = (p0 - RP) / 4: on a 128-byte line iff NP G == c (mod 8).  Odd NP: one lane
	// number g_a mod 8 does it for every tile of every wave; even NP: the g_a that leaves the fewest low bits (A/B: 3.5 % of the kernel)
	const int c = (p0 - (p0 & 3)) / 4;
	int g_a = 0, best = -1;
	for (int g = 0; g < 8; g++) {
		const int d = ((NP * g - c) % 8 + 8) % 8, z = d == 0 ? 3 : __builtin_ctz((unsigned)d);
		if (z > best) { best = z; g_a = g; }
	}
	const u64 lanes = (M + W - 1) / W, per_wave = 64ull * tw - DL_HALO;
	const unsigned n_waves = lanes > (u64)g_a ? (unsigned)((lanes - (u64)g_a + per_wave - 1) / per_wave) : 1u;
	const unsigned grid = ((n_waves + 3) / 4 + 7u) & ~7u;
	size_t lds = (size_t)4 * NS * 64 * NP * 16;
	if (lds < 52000)
		lds = 52000;
#define DLK(RPV) hipLaunchKernelGGL((k_fm_decimate_lane<RT, DS, RPV>), dim3(grid), dim3(256), lds, s, iq, T, p0, M, pcm, pcm_chl2, tw, n_waves, g_a)
	switch (p0 & 3) {
	case 0: DLK(0); break;
	case 1: DLK(1); break;
	case 2: DLK(2); break;
	default: DLK(3); break;
	}
#undef DLK
}

extern "C" int rxk_fm_decimate_small(void *stream, const int16_t *iq, u64 T, int ds, int p0, int rotate, unsigned long long M, int16_t *pcm,
                                     int pcm_chl2)
{
	{
		if (dl_takes(ds) && M) {
			hipStream_t s = (hipStream_t)stream;
			const uint32_t *p = (const uint32_t *)iq;
#define DL(D) case D: if (rotate) dl_launch<true, D>(s, p, T, p0, M, pcm, pcm_chl2); else dl_launch<false, D>(s, p, T, p0, M, pcm, pcm_chl2); break
			switch (ds) { DL(4); DL(5); DL(6); DL(7); DL(8); DL(9); DL(10); DL(11); DL(12); DL(14); DL(16); DL(18); DL(20); DL(22); DL(24); DL(26); DL(28); DL(30); DL(32); }
#undef DL
			LAUNCH_RET();
		}
	}
	/* what is left for this kernel: the odd ds from 13 to 31 (k_fm_decimate_lane would hold 4 ds registers of raw samples per lane).
	 * The staged span, padded to a fifth of the CU's LDS: five workgroups per CU (20 waves) run the kernel as fast as eight do, and the audio
	 * stages of the previous run -- long, latency-bound waves on the other stream -- always find slots beside them */
	const unsigned span = dsm_span(ds);
	const unsigned grid = ((unsigned)((T + span - 1) / span) + 7u) & ~7u;
	size_t lds = (size_t)(span + 2 * DSM_HALO + 8) * 4;
	if (lds < 32000)
		lds = 32000;
	hipStream_t s = (hipStream_t)stream;
	const u32x4 *p = (const u32x4 *)iq;
	const bool four = (span + 2 * DSM_HALO) / 4 <= 1024;
#define GO(RT) do { if (four) hipLaunchKernelGGL((k_fm_decimate_small<RT, 4>), dim3(grid), dim3(256), lds, s, p, T, ds, p0, M, pcm, pcm_chl2, span, span / (unsigned)ds); \
		else hipLaunchKernelGGL((k_fm_decimate_small<RT, 5>), dim3(grid), dim3(256), lds, s, p, T, ds, p0, M, pcm, pcm_chl2, span, span / (unsigned)ds); } while (0)
	if (rotate) GO(true); else GO(false);
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_decimate_generic(void *stream, const int16_t *iq, u64 T, int ds, int p0, u64 n_per_block,
                                       int prescaled, int rotate, const rxk_fm_dev *dev, uint32_t *lp, u64 M)
{
	if (!M)
		return 0;
	const unsigned grid = (unsigned)((M + 255) / 256);
	hipStream_t s = (hipStream_t)stream;
	if (prescaled)
		hipLaunchKernelGGL((k_fm_decimate_generic<true>), dim3(grid), dim3(256), 0, s, (const uint32_t *)iq, T, ds, p0, n_per_block, 0, dev, lp, M);
	else
		hipLaunchKernelGGL((k_fm_decimate_generic<false>), dim3(grid), dim3(256), 0, s, (const uint32_t *)iq, T, ds, p0, n_per_block, rotate, dev, lp, M);
	LAUNCH_RET();
}

extern "C" int rxk_fm_disc(void *stream, const int16_t *iq, u64 T, int ds, int p0, u64 n_per_block, int prescaled,
                           int rotate, int seams, const uint32_t *lp_raw, const uint32_t *head, const uint32_t *tail,
                           uint32_t *lp, u64 M, int first_mode, u64 uniform_k, int custom_atan, int do_tail,
                           int16_t *pcm, rxk_fm_dev *dev, rxk_flag_rec *flag_list, int *flag_cnt, int sparse, u64 n_blocks, const int *atan_lut,
                           int lp_sparse, int flag_all, int pcm_chl2)
{
	if (!sparse || !seams)
		lp_sparse = 0;
	/* seams == 2 (after rxk_fm_decimate_small): no span seams, just the run's first two outputs */
	const u64 n_wg = seams == 2 ? 1 : (T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN;
	const unsigned out_blocks = sparse ? (unsigned)((2 * n_wg + n_blocks + 1 + 255) / 256) : (unsigned)((M + 255) / 256);
	const unsigned grid = out_blocks + (do_tail ? 1 : 0);
	if (!grid)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	if (prescaled)
		hipLaunchKernelGGL((k_fm_disc<true>), dim3(grid), dim3(256), 0, s, (const uint32_t *)iq, T, ds, p0, n_per_block, 0, seams,
		                   lp_raw, head, tail, lp, M, first_mode, uniform_k, custom_atan, do_tail, pcm, dev, flag_list, flag_cnt, out_blocks, sparse, n_wg, n_blocks, atan_lut, lp_sparse, flag_all, pcm_chl2);
	else
		hipLaunchKernelGGL((k_fm_disc<false>), dim3(grid), dim3(256), 0, s, (const uint32_t *)iq, T, ds, p0, n_per_block, rotate, seams,
		                   lp_raw, head, tail, lp, M, first_mode, uniform_k, custom_atan, do_tail, pcm, dev, flag_list, flag_cnt, out_blocks, sparse, n_wg, n_blocks, atan_lut, lp_sparse, flag_all, pcm_chl2);
	LAUNCH_RET();
}

static unsigned magic_for(int a) { return a > 1 ? (unsigned)((1ull << 32) / (unsigned)a + 1) : 0u; }

static int bias_for(int a) { return 65536 / a + 2; }

// the 24-bit forms: the unsigned division takes (2^26/a + 1) < 2^24, i.e. a >= 5; the three-instruction step multiplies by the same
// constant with a SIGNED 24-bit multiply-high (v_mul_hi_i32_i24), so it needs 2^26/a + 1 < 2^23: a >= 9.  One predicate for both.
static bool deemph_d24(int a) { return a >= 9 && a < 256; }
static unsigned deemph_magic(int a) { return deemph_d24(a) ? (1u << 26) / (unsigned)a + 1 : magic_for(a); }
// the kernels that divide only through deemph_div / deemph_mul (k_ch_audio, k_cha_*, k_fm_row_audio) take the two-multiply 24-bit form from a = 5:
// floor(t / a) == ((t << 6) * (2^26 / a + 1)) >> 32 for every t < 2^18 whenever the constant fits 24 bits UNSIGNED (a >= 5; checked for all t and
// a = 5 .. 272).  deemph_d24 starts at 9 because the tiled rx_fm kernels' three-instruction step multiplies SIGNED 24-bit operands (de_step).
static bool deemph_d24u(int a) { return a >= 5 && a < 256; }
static unsigned deemph_magic_u(int a) { return deemph_d24u(a) ? (1u << 26) / (unsigned)a + 1 : magic_for(a); }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

template <typename K>
static void deemph_lds_attr(K kernel, size_t lds)
{
	if (lds > 48 * 1024)
		(void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

extern "C" int rxk_fm_deemph_scan(void *stream, const int16_t *pcm, u64 M, int a, int group, int chunk, int warm,
                                  int lo0, int hi0, int *pre, int *p_tab, int *p_lo, int *p_gap, rxk_fm_dev *dev)
{
	if (!M)
		return 0;
	const u64 n_chunks = (M + chunk - 1) / chunk;
	const unsigned grid = (unsigned)((n_chunks + 63) / 64);
	const size_t lds = 65 * (size_t)(chunk / 8 + 1) * 16 + (size_t)(64 * group + 128) * 4;
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph_magic(a);
	const int bias = bias_for(a), l2 = ilog2(chunk);
#define GO(GS, EV, D) do { deemph_lds_attr(k_fm_deemph_scan<GS, EV, D>, lds); \
		hipLaunchKernelGGL((k_fm_deemph_scan<GS, EV, D>), dim3(grid), dim3(64), lds, s, pcm, M, a, mg, bias, l2, warm, \
		                   lo0, hi0, pre, p_tab, p_lo, p_gap, dev); } while (0)
	if (group == 16) {
		if (deemph_d24(a)) { if (a & 1) GO(16, false, true); else GO(16, true, true); }
		else { if (a & 1) GO(16, false, false); else GO(16, true, false); }
	} else {
		if (a & 1) GO(64, false, true); else GO(64, true, true);
	}
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_up(void *stream, u64 n_child, int group, const int *tab, const int *lo, const int *gap,
                                int *p_tab, int *p_lo, int *p_gap)
{
	const u64 parents = (n_child + DEEMPH_FAN - 1) / DEEMPH_FAN;
	const u64 threads = parents * group;
	hipLaunchKernelGGL(k_fm_deemph_up, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   n_child, group, tab, lo, gap, p_tab, p_lo, p_gap);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_top(void *stream, int n, int group, const int *tab, const int *lo, const int *gap,
                                 int *start, rxk_fm_dev *dev)
{
	int R = 1;
	while (R * R < n) R++;
	const int nseg = (n + R - 1) / R;
	const size_t shm = ((size_t)n * group + 2 * (size_t)n + (size_t)nseg * group + nseg) * sizeof(int);
	static size_t allowed = 64 * 1024;            /* raise the dynamic-LDS cap once, not per launch */
	if (shm > allowed) {
		(void)hipFuncSetAttribute((const void *)k_fm_deemph_top, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
		allowed = 144 * 1024;
	}
	hipLaunchKernelGGL(k_fm_deemph_top, dim3(1), dim3(256), shm, (hipStream_t)stream, n, group, tab, lo, gap, start, dev);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_down(void *stream, u64 n_child, int group, const int *tab, const int *lo,
                                  const int *p_start, int *start)
{
	const u64 parents = (n_child + DEEMPH_FAN - 1) / DEEMPH_FAN;
	hipLaunchKernelGGL(k_fm_deemph_down, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   n_child, group, tab, lo, p_start, start);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_apply(void *stream, const int16_t *pcm, u64 M, int a, int group, int chunk, const int *pre,
                                   const int *p_lo, const int *p_start, int16_t *y)
{
	if (!M)
		return 0;
	const u64 n_chunks = (M + chunk - 1) / chunk;
	const unsigned grid = (unsigned)((n_chunks + 63) / 64);
	const size_t lds = 65 * (size_t)(chunk / 8 + 1) * 16;
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph_magic(a);
	const int bias = bias_for(a), l2 = ilog2(chunk);
#define GO(EV, D) do { deemph_lds_attr(k_fm_deemph_apply<EV, D>, lds); \
		hipLaunchKernelGGL((k_fm_deemph_apply<EV, D>), dim3(grid), dim3(64), lds, s, pcm, M, a, mg, bias, l2, group, pre, p_lo, p_start, y); } while (0)
	if (deemph_d24(a)) { if (a & 1) GO(false, true); else GO(true, true); }
	else { if (a & 1) GO(false, false); else GO(true, false); }
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_tiled_ok(int a, int group, int chunk, int fast, int slow)
{
	if (!(a & 1) || !deemph_d24(a) || (group != 16 && group != 64) || (chunk != 128 && chunk != 256))
		return 0;
	if (slow <= 0 || fast < 2 * slow || fast / slow > 32)
		return 0;
	return chunk == 128 ? 7 : 8;
}

extern "C" int rxk_fm_deemph_scan_t(void *stream, const int16_t *pcm_t, u64 M, int a, int group, int chl2, int warm, int lo0, int gap_w,
                                    void *ctab, rxk_fm_dev *dev)
{
	if (!M)
		return 0;
	const u64 n_chunks = (M + (1u << chl2) - 1) >> chl2;
	const unsigned grid = (unsigned)((n_chunks + 255) / 256);
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph_magic(a);
	const char *pick = rxgpu_knob("RXGPU_SCAN_T");                       /* "1": always scan_t, "0": scan_r wherever it applies (tests) */
	/* (256-sample chunks -- a = 19 at 240 kHz, BASELINE configs[0] -- always take scan_t: the register form reads every chunk once where scan_t
	 * reads the tails twice, 0.43 instead of 0.74 GB per 4 GiB of capture at ds = 5, but at 145 VGPRs per wave it took 1.2 ms instead of 0.66 and
	 * the pipelined step did not move: round 4, A/B in one process; the instantiation went in round 6) */
	if (chl2 == 7 && (pick ? pick[0] == '0' : M >= (1ull << 25))) {
		/* 128-sample chunks of a LONG run (the small-decimation chains, where the audio stages' traffic counts): the chunk in
		 * registers, the warm-up from the neighbouring lane; 63 chunks per wave.  Short runs keep scan_t: behind the big-ds
		 * decimator (56 VGPRs x 8 waves per SIMD) a wave of this kernel (81 VGPRs) waits for TWO of its waves to leave, and the
		 * audio chain of the headline run doubles (A/B at ds=118: 3 % on the pipelined step) */
		const unsigned rgrid = (unsigned)(((n_chunks + 62) / 63 + 3) / 4);
		if (group == 16)
			hipLaunchKernelGGL((k_fm_deemph_scan_r<16>), dim3(rgrid), dim3(256), 0, s, pcm_t, M, a, mg, warm, lo0, gap_w, (uint4 *)ctab, dev);
		else
			hipLaunchKernelGGL((k_fm_deemph_scan_r<64>), dim3(rgrid), dim3(256), 0, s, pcm_t, M, a, mg, warm, lo0, gap_w, (uint4 *)ctab, dev);
		LAUNCH_RET();
	}
#define GO(GS, CL) hipLaunchKernelGGL((k_fm_deemph_scan_t<GS, CL>), dim3(grid), dim3(256), 0, s, pcm_t, M, a, mg, warm, lo0, gap_w, (uint4 *)ctab, dev)
	if (group == 16) { if (chl2 == 7) GO(16, 7); else GO(16, 8); }
	else { if (chl2 == 7) GO(64, 7); else GO(64, 8); }
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_up0(void *stream, u64 n_chunks, int group, const void *ctab, int *p_tab, int *p_lo, int *p_gap)
{
	const u64 parents = (n_chunks + DEEMPH_FAN - 1) / DEEMPH_FAN;
	const u64 threads = parents * group;
	hipLaunchKernelGGL(k_fm_deemph_up0, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   n_chunks, group, (const uint4 *)ctab, p_tab, p_lo, p_gap);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_down0(void *stream, u64 n_chunks, const void *ctab, const int *p_start, int *start)
{
	const u64 parents = (n_chunks + DEEMPH_FAN - 1) / DEEMPH_FAN;
	hipLaunchKernelGGL(k_fm_deemph_down0, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
	                   n_chunks, (const uint4 *)ctab, p_start, start);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_apply_rs_t(void *stream, const int16_t *pcm_t, u64 M, int a, int chl2, const int *start, int fast, int slow,
                                        int16_t *out, rxk_fm_dev *dev)
{
	if (!M)
		return 0;
	const u64 n_chunks = (M + (1u << chl2) - 1) >> chl2;
	const int ratio = fast / slow;
	// the reciprocal rounded UP: (int)((float)sum * rinv) is C's truncating sum / ratio for |sum| <= 34 * 32768, ratio <= 32
	const float rinv = __builtin_nextafterf((float)(1.0 / (double)ratio), __builtin_inff());
	// outputs one wave can produce: 64 chunks' worth of input, +1 window finished for a neighbour, +1 rounding
	const int wcap = ((int)((((u64)64 << chl2) * (u64)slow) / (u64)fast) + 6) & ~1;       /* + the spare element in front, even */
	// small workgroups: about 12 KiB of staging each (2 waves at the wbfm ratios), so that one finds room beside the
	// decimator's workgroups of the next run, which fill most of a CU's LDS
	int wpb = 4;
	while (wpb > 1 && (size_t)wpb * wcap * sizeof(int) > 12800)
		wpb >>= 1;
	const size_t lds = (size_t)wpb * wcap * sizeof(int);
	if (lds > 65536)
		return -1;
	const unsigned grid = (unsigned)((n_chunks + 64 * wpb - 1) / (64 * wpb));
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph_magic(a);
	if (chl2 == 7)
		hipLaunchKernelGGL((k_fm_deemph_apply_rs_t<7>), dim3(grid), dim3(64 * wpb), lds, s, pcm_t, M, a, mg, start, fast, slow, rinv, wcap, out, dev);
	else
		hipLaunchKernelGGL((k_fm_deemph_apply_rs_t<8>), dim3(grid), dim3(64 * wpb), lds, s, pcm_t, M, a, mg, start, fast, slow, rinv, wcap, out, dev);
	LAUNCH_RET();
}

extern "C" int rxk_fm_deemph_serial(void *stream, const int16_t *pcm, u64 M, int a, int16_t *y, rxk_fm_dev *dev)
{
	hipLaunchKernelGGL(k_fm_deemph_serial, dim3(1), dim3(64), 0, (hipStream_t)stream, pcm, M, a, y, dev);
	LAUNCH_RET();
}

extern "C" int rxk_fm_resample(void *stream, const int16_t *y, u64 n, int fast, int slow, u64 J, int16_t *out, rxk_fm_dev *dev)
{
	hipLaunchKernelGGL(k_fm_resample, dim3((unsigned)((J + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, n, fast, slow, J, out, dev);
	LAUNCH_RET();
}

// A few bytes from one place to another as ONE WAVE.  hipMemcpyAsync of 4..240 bytes becomes a blit kernel whose workgroup
// waits for room while an HBM-bound kernel fills the chip -- rocprofv3 showed the 4-byte copy of the flag count sitting for a
// millisecond in front of the audio stages, and the next run's decimator waiting for those.  dst may be pinned host memory.
// a device buffer into its page-locked host mirror (16-byte aligned both sides): 16-byte units, the odd bytes at the end one by one
__global__ __launch_bounds__(256) void k_copy_mirror(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, unsigned n)
{
	const unsigned units = n >> 4;
	for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u)
		reinterpret_cast<uint4 *>(dst)[u] = reinterpret_cast<const uint4 *>(src)[u];
	if (blockIdx.x == 0 && threadIdx.x < (n & 15u))
		dst[(units << 4) + threadIdx.x] = src[(units << 4) + threadIdx.x];
}

__global__ void k_copy_small(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, unsigned n)
{
	for (unsigned i = threadIdx.x; i < n; i += 64)
		dst[i] = src[i];
}

extern "C" int rxk_copy_mirror(void *stream, void *dst, const void *src, unsigned bytes)
{
	if (!bytes)
		return 0;
	const unsigned grid = (bytes / 16 + 255) / 256;
	hipLaunchKernelGGL(k_copy_mirror, dim3(grid ? (grid > 64 ? 64 : grid) : 1), dim3(256), 0, (hipStream_t)stream, (uint8_t *)dst, (const uint8_t *)src, bytes);
	LAUNCH_RET();
}

extern "C" int rxk_copy_small(void *stream, void *dst, const void *src, unsigned bytes)
{
	if (!bytes)
		return 0;
	hipLaunchKernelGGL(k_copy_small, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint8_t *)dst, (const uint8_t *)src, bytes);
	LAUNCH_RET();
}

extern "C" int rxk_fm_carry_advance(void *stream, rxk_fm_dev *dev, int advance, int *snap)
{
	hipLaunchKernelGGL(k_fm_carry_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, advance, snap);
	LAUNCH_RET();
}

extern "C" int rxk_fm_audio_carry(void *stream, rxk_fm_dev *dev, const int *snap)
{
	hipLaunchKernelGGL(k_fm_audio_carry, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, snap);
	LAUNCH_RET();
}

extern "C" int rxk_fm_passthrough_carry(void *stream, rxk_fm_dev *dev, int deemph_off, int resample_off)
{
	hipLaunchKernelGGL(k_fm_passthrough_carry, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, deemph_off, resample_off);
	LAUNCH_RET();
}

extern "C" int rxk_fm_fifth_pass(void *stream, const void *in, int in_is_raw, int prescaled, int rotate, u64 n_blocks,
                                 unsigned n_in, unsigned in_stride, uint32_t *out, unsigned out_stride,
                                 const int16_t *hist_in, int16_t *hist_out)
{
	const unsigned n_out = (n_in + 1) / 2;
	if (!n_blocks || !n_out)
		return 0;
	const dim3 grid((n_out + 255) / 256, (unsigned)(n_blocks < 65535 ? n_blocks : 65535));
	hipStream_t s = (hipStream_t)stream;
	if (!in_is_raw)
		hipLaunchKernelGGL((k_fm_fifth_pass<false, true, false>), grid, dim3(256), 0, s, in, n_blocks, n_in, in_stride, out, out_stride, hist_in, hist_out);
	else if (prescaled)
		hipLaunchKernelGGL((k_fm_fifth_pass<true, true, false>), grid, dim3(256), 0, s, in, n_blocks, n_in, in_stride, out, out_stride, hist_in, hist_out);
	else if (rotate)
		hipLaunchKernelGGL((k_fm_fifth_pass<true, false, true>), grid, dim3(256), 0, s, in, n_blocks, n_in, in_stride, out, out_stride, hist_in, hist_out);
	else
		hipLaunchKernelGGL((k_fm_fifth_pass<true, false, false>), grid, dim3(256), 0, s, in, n_blocks, n_in, in_stride, out, out_stride, hist_in, hist_out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_droop(void *stream, const uint32_t *in, u64 M, const int *fir, const int16_t *hist_in,
                            int16_t *hist_out, uint32_t *out)
{
	if (!M)
		return 0;
	hipLaunchKernelGGL(k_fm_droop, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, M, fir, hist_in, hist_out, out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_rdc(void *stream, const int16_t *iq, u64 n_blocks, u64 n_per_block, int prescaled, int rotate,
                          int rdc_block_const, int *state, long long *sums, int *avg, int16_t *out)
{
	if (!n_blocks || !n_per_block)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const u64 T = n_blocks * n_per_block;
	hipError_t e = hipMemsetAsync(sums, 0, (size_t)n_blocks * 16, s);
	if (e != hipSuccess)
		return (int)e;
	hipLaunchKernelGGL(k_fm_rdc_sums, dim3((unsigned)(n_blocks * RDC_PARTS)), dim3(256), 0, s, (const uint32_t *)iq, n_per_block, prescaled, (i64 *)sums);
	hipLaunchKernelGGL(k_fm_rdc_scan, dim3(1), dim3(64), 0, s, (const i64 *)sums, n_blocks, (int)n_per_block, rdc_block_const, state, avg);
	hipLaunchKernelGGL(k_fm_rdc_apply, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, (const uint32_t *)iq, T, n_per_block, prescaled, rotate,
	                   avg, (uint32_t *)out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_post_downsample(void *stream, const int16_t *in, u64 n_out, int step, int16_t *out)
{
	if (!n_out)
		return 0;
	hipLaunchKernelGGL(k_fm_post_downsample, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, n_out, step, out);
	LAUNCH_RET();
}

// The same with input and second output in HOST memory the device can address (page-locked: rxgpu_pin / rxgpu_dropin_pin): the callback's block
// crosses PCIe inside this launch -- 16-byte reads of the raw block, 16-byte writes of the scaled one to HBM (for full_demod) and back to the
// caller's buffer -- instead of in two DMA transfers around a kernel (three stream operations, each with its own latency: 47 us for 1 MiB).
__global__ __launch_bounds__(256) void k_fm_prestage_zc(const uint32_t *__restrict__ in, unsigned n, int rotate, uint32_t *__restrict__ out_dev,
                                                        uint32_t *__restrict__ out_host)
{
	// two samples per thread, 8-byte pieces: buf16[] sits 8 bytes off a 16-byte boundary in struct dongle_state (rtl_fm.c:128-147)
	const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned i0 = 2 * v;
	if (i0 >= n)
		return;
	if (i0 + 2 <= n) {
		const uint2 w = *reinterpret_cast<const uint2 *>(in + i0);
		const uint32_t ww[2] = {w.x, w.y};
		uint32_t r[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const int i = scale_cs16(lo16(ww[k])), q = scale_cs16(hi16(ww[k]));
			int ri = i, rq = q;
			if (rotate) {
				switch ((i0 + k) & 3u) {
				case 1: ri = -q; rq = i; break;
				case 2: ri = -i; rq = -q; break;
				case 3: ri = q; rq = -i; break;
				default: break;
				}
			}
			r[k] = pack_iq(ri, rq);
		}
		const uint2 o = make_uint2(r[0], r[1]);
		*reinterpret_cast<uint2 *>(out_dev + i0) = o;
		*reinterpret_cast<uint2 *>(out_host + i0) = o;
		return;
	}
	int ri, rq;
	load_rot<false>(in, i0, rotate ? i0 : 0u, ri, rq);
	const uint32_t o = pack_iq(ri, rq);
	out_dev[i0] = o;
	out_host[i0] = o;
}

extern "C" int rxk_fm_prestage_zc(void *stream, const int16_t *in_host, unsigned n_complex, int rotate, int16_t *out_dev, int16_t *out_host)
{
	if (!n_complex)
		return 0;
	const unsigned vecs = (n_complex + 1) / 2;
	hipLaunchKernelGGL(k_fm_prestage_zc, dim3((vecs + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t *)in_host, n_complex, rotate, (uint32_t *)out_dev, (uint32_t *)out_host);
	LAUNCH_RET();
}

extern "C" int rxk_fm_prestage(void *stream, const int16_t *in, unsigned n_complex, int rotate, int16_t *out)
{
	if (!n_complex)
		return 0;
	hipLaunchKernelGGL(k_fm_prestage, dim3((n_complex + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t *)in, n_complex, rotate, (uint32_t *)out);
	LAUNCH_RET();
}

// the seam histories of a fused group alone (k_fm_fifth_seams), for callers that run it ahead of the group on another stream;
// rxk_fm_fifth_fused with hist_in == NULL then skips it
extern "C" int rxk_fm_fifth_seams(void *stream, const void *in, int stage2, int rotate, u64 n_blocks, unsigned n, int fuse,
                                  const int16_t *hist_in, int16_t *hist_out, uint32_t *seams)
{
	const u64 seam_threads = (n_blocks + 1) * 16;
	const unsigned sgrid = (unsigned)((seam_threads + 255) / 256);
	const uint32_t *p = (const uint32_t *)in;
	hipStream_t s = (hipStream_t)stream;
	if (fuse >= 4 && !stage2) {                                  /* the four- / five-pass register kernels' layout: 5 * fuse dwords per block */
		const unsigned sgrid4 = (unsigned)(((n_blocks + 1) * 32 + 255) / 256);
		if (fuse == 4) {
			if (rotate) hipLaunchKernelGGL((k_fm_fifth_seams<true, false, 4>), dim3(sgrid4), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
			else hipLaunchKernelGGL((k_fm_fifth_seams<false, false, 4>), dim3(sgrid4), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
		} else {
			if (rotate) hipLaunchKernelGGL((k_fm_fifth_seams<true, false, 5>), dim3(sgrid4), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
			else hipLaunchKernelGGL((k_fm_fifth_seams<false, false, 5>), dim3(sgrid4), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
		}
		LAUNCH_RET();
	}
	if (stage2) hipLaunchKernelGGL((k_fm_fifth_seams<false, true>), dim3(sgrid), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
	else if (rotate) hipLaunchKernelGGL((k_fm_fifth_seams<true, false>), dim3(sgrid), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
	else hipLaunchKernelGGL((k_fm_fifth_seams<false, false>), dim3(sgrid), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out);
	LAUNCH_RET();
}

// `fuse` (1..3) fifth_order passes in one LDS-tiled launch.  stage2 == 0: raw cs16 in (scale + rotate, packed int16
// arithmetic); stage2 != 0: packed level samples in (int arithmetic), hist_* already offset to the first pass done
extern "C" int rxk_fm_fifth_fused(void *stream, const void *in, int stage2, int rotate, u64 n_blocks, unsigned n, int fuse,
                                  const int16_t *hist_in, int16_t *hist_out, uint32_t *seams, uint32_t *out)
{
	hipStream_t s = (hipStream_t)stream;
	const unsigned tiles = n / FF_RAW;
	const unsigned tpw = tiles % 16 == 0 ? 16 : tiles % 8 == 0 ? 8 : tiles % 4 == 0 ? 4 : tiles % 2 == 0 ? 2 : 1;   // tiles one workgroup walks
	const u64 seam_threads = (n_blocks + 1) * 16;
	const unsigned sgrid = (unsigned)((seam_threads + 255) / 256);
	const unsigned grid = (unsigned)(n_blocks * (tiles / tpw));
	const uint32_t *p = (const uint32_t *)in;
#define SEAMS(RT, S2) hipLaunchKernelGGL((k_fm_fifth_seams<RT, S2>), dim3(sgrid), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out)
	/* the raw stage's workgroups take a fifth of the CU's LDS each (15 KiB of tiles + this pad): with eight per CU the kernel is no
	 * faster, and the later passes / discriminator / audio stages of the previous run on the other stream wait for wave slots
	 * (A/B in one process, -F ds=128 pipelined: 0.92 -> 0.98 TSample/s) */
	const size_t pad = stage2 ? 0 : 17000;
#define FUSED(F, RT, S2) hipLaunchKernelGGL((k_fm_fifth_fused<F, RT, S2, false>), dim3(grid), dim3(256), pad, s, p, n, tiles, tpw, seams, out, n, n >> F)
#define GO(RT, S2) do { if (hist_in) SEAMS(RT, S2); if (fuse == 1) FUSED(1, RT, S2); else if (fuse == 2) FUSED(2, RT, S2); else FUSED(3, RT, S2); } while (0)
	/* four passes on the raw capture: the register kernel, 16 samples per lane (five -- 1/32 out -- bought nothing more: round 3).  A group of three
	 * passes that is not the whole chain (k_fm_fifth_regn<., 3, DD>, rxk_fm_fifth_dd) stays with the LDS-tiled kernel below: 3-5 % ahead there (A/B) */
	if (!stage2 && fuse == 4) {
		const unsigned tiles_r = ((n >> fuse) + FR_OUT - 1) / FR_OUT;
		/* one tile per wave: walking two or four with the next one's loads in flight (what the whole-chain kernel below does) made this
		 * one, which has half the arithmetic per byte, 5-10 % slower (A/B, -F ds=128: 1730 / 1823 / 1908 us per step) */
		const unsigned wgs_per_block = (tiles_r + 3) / 4;
		const u64 total = n_blocks * (u64)wgs_per_block;
		if (total > 0xfffffff0ull)
			return (int)hipErrorInvalidValue;
		const unsigned rgrid = (unsigned)((total + 7) / 8 * 8);
		const unsigned sgridn = (unsigned)(((n_blocks + 1) * 32 + 255) / 256);
#define SEAMN(RT, LVV) hipLaunchKernelGGL((k_fm_fifth_seams<RT, false, LVV>), dim3(sgridn), dim3(256), 0, s, p, n_blocks, n, fuse, hist_in, seams, hist_out)
#define REGK(RT, LVV, T) hipLaunchKernelGGL((k_fm_fifth_regn<RT, LVV, 0, T>), dim3(rgrid), dim3(256), 0, s, p, n, tiles_r, wgs_per_block, (unsigned)total, seams, out, \
		                                    (const uint32_t *)nullptr, 0, 0, 0, 0, 0, (int16_t *)nullptr, 0, (uint32_t *)nullptr)
#define REGN(RT, LVV) REGK(RT, LVV, 1)
#define GON(LVV) do { if (rotate) { if (hist_in) SEAMN(true, LVV); REGN(true, LVV); } else { if (hist_in) SEAMN(false, LVV); REGN(false, LVV); } } while (0)
		GON(4);
#undef GON
#undef REGN
#undef REGK
#undef SEAMN
		LAUNCH_RET();
	}
	if (stage2) GO(false, true);
	else if (rotate) GO(true, false);
	else GO(false, false);
#undef GO
#undef FUSED
#undef SEAMS
	LAUNCH_RET();
}

// the whole -F chain of a three-pass cascade in one launch (k_fm_fifth_regn<.., 3, DD>): seams from rxk_fm_fifth_seams (fuse = 3), tails from
// rxk_fm_fifth_tails; fir == NULL: no droop FIR.  n % RXK_FIFTH_TILE == 0
extern "C" int rxk_fm_fifth_dd(void *stream, const void *in, int rotate, u64 n_blocks, unsigned n, int fuse, const uint32_t *seams, const uint32_t *tails,
                               const int *fir, int16_t *pcm, int pcm_chl2, uint32_t *edges)
{
	if (fuse != 3)
		return (int)hipErrorInvalidValue;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t *p = (const uint32_t *)in;
	const unsigned tiles_r = ((n >> fuse) / 4 + FR_OUT - 1) / FR_OUT;
	/* two tiles per wave, the second one's loads in flight behind the first one's arithmetic (A/B round 3: 1 / 2 / 4 tiles, two is ahead); short
	 * blocks: one */
	const unsigned twn = tiles_r >= 8 ? 2u : 1u;
	const unsigned wgs_per_block = (tiles_r + 4 * twn - 1) / (4 * twn);
	const u64 total = n_blocks * (u64)wgs_per_block;
	if (total > 0xfffffff0ull)
		return (int)hipErrorInvalidValue;
	const unsigned rgrid = (unsigned)((total + 7) / 8 * 8);
#define DDK(RT, D, T) hipLaunchKernelGGL((k_fm_fifth_regn<RT, 3, D, T>), dim3(rgrid), dim3(256), 0, s, p, n, tiles_r, wgs_per_block, (unsigned)total, seams, (uint32_t *)nullptr, \
		                                 tails, fir ? fir[1] : 0, fir ? fir[2] : 0, fir ? fir[3] : 0, fir ? fir[4] : 0, fir ? fir[5] : 0, pcm, pcm_chl2, edges)
#define DDT(RT, D) do { if (twn == 2) DDK(RT, D, 2); else DDK(RT, D, 1); } while (0)
	if (fir) { if (rotate) DDT(true, 2); else DDT(false, 2); }
	else { if (rotate) DDT(true, 1); else DDT(false, 1); }
#undef DDT
#undef DDK
	LAUNCH_RET();
}

extern "C" int rxk_fm_fifth_tails(void *stream, const void *in, int rotate, u64 n_blocks, unsigned n, int fuse, const int16_t *droop_in, int16_t *droop_out,
                                  uint32_t *tails)
{
	if (fuse != 3)
		return (int)hipErrorInvalidValue;
	const unsigned grid = (unsigned)(((n_blocks + 1) * 16 + 255) / 256);
	if (rotate) hipLaunchKernelGGL((k_fm_fifth_tails<true, 3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)in, n_blocks, n, droop_in, droop_out, tails);
	else hipLaunchKernelGGL((k_fm_fifth_tails<false, 3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)in, n_blocks, n, droop_in, droop_out, tails);
	LAUNCH_RET();
}

extern "C" int rxk_fm_dd_edges(void *stream, const uint32_t *edges, u64 n_blocks, u64 K, int16_t *pcm, int pcm_chl2, rxk_fm_dev *dev, rxk_flag_rec *flag_list,
                               int *flag_cnt, int flag_all)
{
	hipLaunchKernelGGL(k_fm_dd_edges, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, edges, n_blocks, K, pcm, pcm_chl2, dev, flag_list,
	                   flag_cnt, flag_all);
	LAUNCH_RET();
}

extern "C" int rxk_fm_fifth_lit(void *stream, const int16_t *in, int16_t *out, int L, const int16_t *hist_in, int16_t *hist_out)
{
	const int K = L > 4 ? 1 + (L - 1) / 4 : 1;                     // the I half has at least as many outputs as the Q half
	hipLaunchKernelGGL(k_fm_fifth_lit, dim3((unsigned)((K + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, in, out, L, hist_in, hist_out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_droop_lit(void *stream, const int16_t *in, int16_t *out, int L, const int *fir, const int16_t *hist_in, int16_t *hist_out)
{
	const int n = (L > 2 ? L : 2) / 2 + 2;                          // the filtered samples of the longer half and the few entries behind them
	hipLaunchKernelGGL(k_fm_droop_lit, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, in, out, L, fir, hist_in, hist_out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_squelch_lit(void *stream, int16_t *lp, int L, int level, int *below, int *sr_out)
{
	hipLaunchKernelGGL(k_fm_squelch_lit, dim3(1), dim3(256), 0, (hipStream_t)stream, lp, L, level, below, sr_out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_demod_lit(void *stream, const int16_t *lp, int L, int mode, int custom_atan, int output_scale, int16_t *pcm,
                                unsigned long long m0, rxk_fm_dev *dev, int pre_from_out, rxk_flag_rec *flag_list, int *flag_cnt,
                                const int *atan_lut, int flag_all)
{
	const int n = mode == RXK_LIT_RAW ? L : L / 2;
	if (n > 0) {
		hipLaunchKernelGGL(k_fm_demod_lit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lp, L, mode, custom_atan, output_scale,
		                   pcm, (u64)m0, dev, pre_from_out, flag_list, flag_cnt, atan_lut, flag_all);
		if (hipGetLastError() != hipSuccess)
			return (int)hipErrorLaunchFailure;
	}
	if (mode == RXK_LIT_FM && L >= 2)
		hipLaunchKernelGGL(k_fm_pre_lit, dim3(1), dim3(1), 0, (hipStream_t)stream, lp, L, dev);
	LAUNCH_RET();
}

extern "C" int rxk_fm_squelch(void *stream, uint32_t *lp, rxk_fm_blocks blk, int level, int *below, int *sr_out)
{
	hipLaunchKernelGGL(k_fm_squelch, dim3((unsigned)blk.n_blocks), dim3(256), 0, (hipStream_t)stream, lp, blk, level, below, sr_out);
	LAUNCH_RET();
}

extern "C" int rxk_fm_simple_demod(void *stream, const uint32_t *lp, u64 M, int mode, int output_scale, int16_t *pcm)
{
	if (!M)
		return 0;
	hipLaunchKernelGGL(k_fm_simple_demod, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lp, M, mode, output_scale, pcm);
	LAUNCH_RET();
}

extern "C" int rxk_fm_dc_block(void *stream, int16_t *y, u64 M, rxk_fm_blocks blk, int adc_block_const, long long *sums,
                               int *avgs, rxk_fm_dev *dev)
{
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(k_fm_dc_sums, dim3((unsigned)blk.n_blocks), dim3(256), 0, s, y, blk, (i64 *)sums);
	hipLaunchKernelGGL(k_fm_dc_scan, dim3(1), dim3(64), 0, s, (const i64 *)sums, blk, adc_block_const, avgs, dev);
	hipLaunchKernelGGL(k_fm_dc_apply, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, y, M, blk, avgs);
	LAUNCH_RET();
}

extern "C" int rxk_ch_fused_ok(int bin_e, u64 wpb, int custom_atan, int n_channels)
{
	const int wpg = ch_wpg(bin_e);
	/* the run of windows a workgroup walks: what k_ch_demod(sparse) then skips over */
	return (bin_e >= 8 && bin_e <= 12 && custom_atan == 1 && wpb % wpg == 0) ? wpg * ch_gpw(wpg, wpb) : 0;
}

extern "C" int rxk_ch_fft(void *stream, const int16_t *iq, u64 total_windows, int bin_e, const uint32_t *twiddle,
                          int first_bin, int n_channels, uint32_t *chan_lp, int fused, int16_t *out, u64 out_stride, int *pre_out)
{
	if (!total_windows)
		return 0;
	if (bin_e >= 8 && bin_e <= 12) {
		const int CH_WPG = ch_wpg(bin_e);
		const int GPW = fused ? fused / CH_WPG : ch_gpw(CH_WPG, 0);
		/* N <= 1024: one transpose area (k_ch_fftR); rows of RXK_FFT_XROW dwords (fft_exchange's layout), the permuted twiddle copy, the staging */
		const size_t shm = (size_t)((bin_e <= 10 ? 1 : 2) * 256 * RXK_FFT_XROW + 8 * ((1 << (bin_e - 4)) + 8) + n_channels * (CH_WPG + 3)) * 4;
		const u64 run = (u64)CH_WPG * GPW;
		const unsigned grid = (unsigned)(((total_windows + run - 1) / run + 7) / 8 * 8);   /* XCD-contiguous order inside the kernel */
		hipStream_t s = (hipStream_t)stream;
		const uint32_t *p = (const uint32_t *)iq;
#define GOF__(MM, FU, TW, WP) do { if (shm > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_ch_fftR<MM, FU, TW, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
		hipLaunchKernelGGL((k_ch_fftR<MM, FU, TW, WP>), dim3(grid), dim3(256), shm, s, p, total_windows, twiddle + (1 << (MM - 1)), first_bin, n_channels, chan_lp, \
		                   out, out_stride, pre_out, GPW); } while (0)
#define GOF(MM, FU) GOF__(MM, FU, true, 16)
#define GOC(MM) do { if (fused) GOF(MM, true); else GOF(MM, false); } while (0)
		switch (bin_e) {
		case 8: GOC(8); break; case 9: GOC(9); break; case 10: GOC(10); break; case 11: GOC(11); break; default: GOC(12); break;
		}
#undef GOC
#undef GOF
#undef GOF__
		LAUNCH_RET();
	}
	int wpg = bin_e >= 13 ? 1 : (8192 >> bin_e);
	const size_t shm = ((size_t)wpg << bin_e) * 4;
	static size_t allowed = 64 * 1024;
	if (shm > allowed) {
		(void)hipFuncSetAttribute((const void *)k_ch_fft, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
		allowed = 144 * 1024;
	}
	const unsigned grid = (unsigned)((total_windows + wpg - 1) / wpg);
	hipLaunchKernelGGL(k_ch_fft, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const uint32_t *)iq, total_windows, bin_e, wpg,
	                   twiddle, first_bin, n_channels, chan_lp);
	LAUNCH_RET();
}

extern "C" int rxk_ch_nco(void *stream, const int16_t *iq, u64 total_windows, int bin_e, const uint32_t *tw_full, int first_bin, int n_channels,
                          uint32_t *chan_lp)
{
	if (!total_windows)
		return 0;
	const int wpg = 4;
	const size_t shm = (size_t)2 * ((size_t)1 << bin_e) * 4;
	if (shm > 64 * 1024)
		(void)hipFuncSetAttribute((const void *)k_ch_nco<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
	hipLaunchKernelGGL(k_ch_nco<0>, dim3((unsigned)((total_windows + wpg - 1) / wpg), (unsigned)((n_channels + 255) / 256)), dim3(256), shm, (hipStream_t)stream,
	                   (const uint32_t *)iq, total_windows, bin_e, tw_full, first_bin, n_channels, chan_lp, wpg);
	LAUNCH_RET();
}

extern "C" int rxk_ch_demod(void *stream, const uint32_t *chan_lp, u64 total_windows, u64 wpb, int n_channels, int custom_atan,
                            const int *pre_in, int *pre_out, int16_t *out, u64 out_stride, rxk_fm_dev *dev, u64 *flag_list, int sparse)
{
	const u64 total = sparse ? (u64)n_channels * ((total_windows + sparse - 1) / sparse) : (u64)n_channels * total_windows;
	if (!total)
		return 0;
	hipLaunchKernelGGL(k_ch_demod, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chan_lp, total_windows, wpb,
	                   n_channels, custom_atan, pre_in, pre_out, out, out_stride, dev, flag_list, sparse,
	                   rxgpu_knob("RXGPU_FLAG_ALL") ? atoi(rxgpu_knob("RXGPU_FLAG_ALL")) : 0);
	LAUNCH_RET();
}

// deemph_filter (rtl_fm.c:667-682) + low_pass_real (389-409) on ONE short row (the drop-in's single blocks: a thousand samples behind ds = 118),
// k_ch_audio's scheme with the row in LDS and the chunk length cut loose from the warm-up: k_ch_audio walks global memory sample by sample
// (46 us for 1 110 samples: every step of every chain waits for its load) and gives each thread a chunk of at least `warm` samples (ten threads
// busy).  Here thread 0 takes the first `warm` samples from the carried state, every other thread a chunk of C samples behind them, warmed up
// on the `warm` samples before its own (two trajectories from the ends of the int16 range, as there), then the chunk tables are walked by one
// thread and every chunk is replayed from its exact start.  a > 64, a == 1 or a carried state outside int16 (serial): one thread, the
// reference's own expression.  W <= row_cap samples (the launcher's LDS size).  row: in (demodulated) / out (result); audio: {avg, now_lpr,
// prev_lpr_index} in, the same three out at audio + 3.
template <bool EVEN, bool D24>
__global__ __launch_bounds__(256) void k_fm_row_audio(const int16_t *row_g, int16_t *row_out, unsigned W, int deemph, int a, unsigned magic, int bias,
                                                      int warm, int serial, int fast, int slow, unsigned J, const int *audio, int *audio_out,
                                                      int16_t *__restrict__ row_h, int *__restrict__ audio_h, const uint32_t *__restrict__ hdr,
                                                      uint32_t *__restrict__ hdr_h, unsigned hdr_words)
{
	extern __shared__ __attribute__((aligned(16))) int16_t ra_row[];       // [W] the samples, then (in place) the de-emphasised ones
	__shared__ uint4 tab[256];
	__shared__ int start[256];
	const int tid = threadIdx.x;
	for (unsigned i = tid; i < W; i += 256)
		ra_row[i] = row_g[i];
	// row_h, audio_h, hdr_h: the page-locked host mirror of the result row, the three audio carries and k_fm_block_dd's header -- what the
	// caller reads after the stream's synchronisation, no copy operation in between
	for (unsigned i = tid; i < hdr_words; i += 256)
		hdr_h[i] = hdr[i];
	const int avg_in = audio[0];
	__syncthreads();
	if (deemph) {
		const int h = a / 2, xoff = h + bias * a;
		// chunk 0 = [0, w0), chunk t >= 1 = [w0 + (t - 1) C, w0 + t C): at most 255 of them
		const unsigned w0 = serial ? W : min((unsigned)warm, W);
		// (the walk over the chunk tables is one thread's, ~250 cycles per table, a step of a chain ~50: chunks of 48 balance the two at W ~ 1000)
		unsigned Cn = (W - w0 + 254) / 255;
		Cn = Cn < 48 ? 48 : ((Cn + 7) & ~7u);                 // a multiple of 8: the chunk starts stay 8-byte aligned for the uint2 reads below
		const int active = 1 + (int)((W - w0 + Cn - 1) / Cn);
		const unsigned b = tid ? w0 + (unsigned)(tid - 1) * Cn : 0u, e = tid ? min(W, b + Cn) : w0;
		if (tid < active && !serial) {
			int lo, hi;
			if (tid == 0) {
				lo = hi = avg_in;
			} else {
				lo = -32768; hi = 32767;
				// four samples per LDS read (b and warm are multiples of 8): a chain step is five instructions, a read's latency a hundred cycles
				for (unsigned i = b - (unsigned)warm; i < b; i += 4) {
					const uint2 v = *reinterpret_cast<const uint2 *>(&ra_row[i]);
					const int xs[4] = {lo16(v.x), hi16(v.x), lo16(v.y), hi16(v.y)};
#pragma unroll
					for (int q = 0; q < 4; q++) {
						lo = deemph_step_d<EVEN, D24>(lo, xs[q] + xoff, xs[q], magic, bias);
						hi = deemph_step_d<EVEN, D24>(hi, xs[q] + xoff, xs[q], magic, bias);
					}
				}
			}
			int gap = hi - lo;
			if (gap > 63) gap = 63;                               // excluded by `warm`
			const int lo_start = lo;
			int cnt = gap + 1;
			u64 mask = (((u64)1 << gap) - 1);
			unsigned i = b;
			for (; i + 4 <= e; i += 4) {
				const uint2 v = *reinterpret_cast<const uint2 *>(&ra_row[i]);
				const int xs[4] = {lo16(v.x), hi16(v.x), lo16(v.y), hi16(v.y)};
#pragma unroll
				for (int q = 0; q < 4; q++)
					deemph_track<EVEN, D24>(lo, cnt, mask, xs[q], a, xoff, magic, bias);
			}
			for (; i < e; i++)
				deemph_track<EVEN, D24>(lo, cnt, mask, (int)ra_row[i], a, xoff, magic, bias);
			tab[tid] = make_uint4((uint32_t)lo_start, ((uint32_t)lo & 0xffffu) | ((uint32_t)gap << 16), (uint32_t)mask, (uint32_t)(mask >> 32));
		}
		__syncthreads();
		if (tid == 0) {
			int s = avg_in;
			if (!serial) {
#pragma unroll 4
				for (int t = 0; t < active; t++) {
					start[t] = s;
					s = ctab_apply(tab[t], s);
				}
				audio_out[0] = s; audio_h[0] = s;
			} else {
				for (unsigned i = 0; i < W; i++) {                // any a, any state: the reference's own expression
					const int d = (int)ra_row[i] - s;
					s += d > 0 ? (d + h) / a : (d - h) / a;
					ra_row[i] = (int16_t)s;
				}
				audio_out[0] = s; audio_h[0] = s;
			}
		}
		__syncthreads();
		if (tid < active && !serial) {
			int s = start[tid];
			unsigned i = b;
			for (; i + 4 <= e; i += 4) {
				const uint2 v = *reinterpret_cast<const uint2 *>(&ra_row[i]);
				const int xs[4] = {lo16(v.x), hi16(v.x), lo16(v.y), hi16(v.y)};
				int ys[4];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					s = deemph_step_d<EVEN, D24>(s, xs[q] + xoff, xs[q], magic, bias);
					ys[q] = s;
				}
				*reinterpret_cast<uint2 *>(&ra_row[i]) = make_uint2(pack_iq(ys[0], ys[1]), pack_iq(ys[2], ys[3]));
			}
			for (; i < e; i++) {
				const int x = ra_row[i];
				s = deemph_step_d<EVEN, D24>(s, x + xoff, x, magic, bias);
				ra_row[i] = (int16_t)s;
			}
		}
		__syncthreads();
	} else if (tid == 0) {
		audio_out[0] = avg_in; audio_h[0] = avg_in;
	}
	if (slow > 0) {
		const u64 p0 = (u64)audio[2];
		const int ratio = fast / slow;
		for (unsigned j = tid; j < J; j += 256) {
			const u64 wb = j ? lpr_end(j - 1, fast, slow, p0) : 0, we = lpr_end(j, fast, slow, p0);
			int sum = j ? 0 : audio[1];
			for (u64 i = wb; i < we; i++)
				sum += ra_row[i];
			if (row_out)
				row_out[j] = (int16_t)(sum / ratio);
			row_h[j] = (int16_t)(sum / ratio);
		}
		if (tid == 0) {
			const u64 wb = J ? lpr_end(J - 1, fast, slow, p0) : 0;
			int sum = J ? 0 : audio[1];
			for (u64 i = wb; i < W; i++)
				sum += ra_row[i];
			const int pl = (int)(p0 + (u64)W * (u64)slow - (u64)J * (u64)fast);
			audio_out[1] = sum; audio_h[1] = sum;
			audio_out[2] = pl; audio_h[2] = pl;
		}
	} else {
		for (unsigned i = tid; i < W; i += 256) {
			if (row_out && (deemph || row_out != row_g))
				row_out[i] = ra_row[i];
			row_h[i] = ra_row[i];
		}
		if (tid == 0) { const int nl = audio[1], pl = audio[2]; audio_out[1] = nl; audio_out[2] = pl; audio_h[1] = nl; audio_h[2] = pl; }
	}
}

extern "C" int rxk_fm_row_audio(void *stream, const int16_t *row, int16_t *row_out, unsigned W, int deemph, int a, int warm, int serial, int fast, int slow,
                                unsigned J, const int *audio, int *audio_out, int16_t *row_h, int *audio_h, const void *hdr, void *hdr_h, unsigned hdr_words)
{
	if (!W)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph ? deemph_magic_u(a) : 0u;
	const int bias = deemph ? bias_for(a) : 0;
	const size_t lds = ((size_t)W * 2 + 15) & ~(size_t)15;
#define GO(EV, D) hipLaunchKernelGGL((k_fm_row_audio<EV, D>), dim3(1), dim3(256), lds, s, row, row_out, W, deemph, a, mg, bias, warm, serial, fast, slow, J, audio, audio_out, \
		row_h, audio_h, (const uint32_t *)hdr, (uint32_t *)hdr_h, hdr_words)
	if (deemph && deemph_d24u(a)) { if (a & 1) GO(false, true); else GO(true, true); }
	else { if (!deemph || (a & 1)) GO(false, false); else GO(true, false); }
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_block_dd(void *stream, const int16_t *blk, unsigned n, int ds, int p0, int now_r, int now_j, int pre_r, int pre_j,
                               int custom_atan, uint32_t *lp, uint32_t *lp_host, int16_t *pcm, int16_t *keep, rxk_blk_out *out, int *audio_in, int avg,
                               int now_lpr, int prev_lpr_index)
{
	const unsigned long long M = ((unsigned long long)p0 + n) / (unsigned long long)ds;
	unsigned grid = (unsigned)((M + 1 + 3) / 4);
	if (grid > 2048)
		grid = 2048;
	hipLaunchKernelGGL(k_fm_block_dd, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)blk, n, ds, p0, now_r, now_j, pre_r, pre_j,
	                   custom_atan, rxgpu_knob("RXGPU_FLAG_ALL") ? atoi(rxgpu_knob("RXGPU_FLAG_ALL")) : 0, lp, lp_host, pcm, keep, out, audio_in, avg, now_lpr, prev_lpr_index);
	LAUNCH_RET();
}

// chunk length and count of the segmented form for a row of W samples; 0 chunks = the row is too short (or too long for CHA_MAX_SEG segments of
// 256 chunks at a sane chunk length): k_ch_audio serves it
extern "C" unsigned rxk_ch_audio_chunks(u64 W, int warm, unsigned *chunk_out)
{
	u64 chunk = (u64)((warm + 7) & ~7);
	const u64 need = (W + 256u * CHA_MAX_SEG - 1) / (256u * CHA_MAX_SEG);
	if (chunk < need)
		chunk = (need + 7) & ~(u64)7;
	if (chunk < 8)
		chunk = 8;
	if (chunk >= 48)
		chunk = (chunk + 63) & ~(u64)63;                              // whole 128-byte lines per lane and turn (k_cha_track)
	const u64 n = (W + chunk - 1) / chunk;
	if (chunk_out)
		*chunk_out = (unsigned)chunk;
	return (n < 64 || chunk > 4096) ? 0u : (unsigned)n;
}

// LDS int16 slots a workgroup of k_cha_replay_rs stages its outputs in (256 chunks' worth + the window finished for a neighbour + rounding);
// 0: the fused form does not fit (the caller keeps k_ch_audio)
static unsigned cha_rs_cap(unsigned chunk, int fast, int slow)
{
	const u64 cap = (256ull * chunk * (u64)slow) / (u64)fast + 4;
	return cap * 2 <= 40960 ? (unsigned)cap : 0u;
}

// can the (segment, channel) form serve rows of W samples (deemph on, a in 2..64, carried states inside int16)?  It reads the demodulated rows
// from one buffer and writes the audio to another.
extern "C" int rxk_ch_audio_seg_ok(u64 W, int warm, int fast, int slow)
{
	unsigned chunk = 0;
	if (!rxk_ch_audio_chunks(W, warm, &chunk) || W >= 0x7fffffffull)
		return 0;
	return slow > 0 ? (cha_rs_cap(chunk, fast, slow) != 0 && fast / slow >= 1 && fast / slow <= 32) : 1;      /* windows shorter than any chunk */
}

// ctab: n_channels * n_chunks tables; seg_start: n_channels * n_chunks ints (every chunk's start state).  in_rows != out_rows.
extern "C" int rxk_ch_audio_seg(void *stream, const int16_t *in_rows, u64 in_stride, int16_t *out_rows, u64 out_stride, u64 W, int n_channels, int a,
                                int warm, int fast, int slow, const int *audio_in, int *audio_out, void *ctab_v, int *seg_start)
{
	uint4 *ctab = (uint4 *)ctab_v;
	unsigned chunk = 0;
	const unsigned n_chunks = rxk_ch_audio_chunks(W, warm, &chunk);
	if (!n_chunks || !rxk_ch_audio_seg_ok(W, warm, fast, slow) || (const int16_t *)out_rows == in_rows)
		return (int)hipErrorInvalidValue;
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph_magic_u(a);
	const int bias = bias_for(a);
	const unsigned n_seg = (n_chunks + 255) / 256;
	const dim3 grid(n_seg, (unsigned)n_channels);
	const int ratio = slow > 0 ? fast / slow : 1;
	// the reciprocal rounded UP: (int)((float)sum * rinv) is C's truncating sum / ratio for |sum| <= 34 * 32768, ratio <= 32 (rxk_fm_deemph_apply_rs_t)
	const float rinv = (slow > 0 && ratio <= 32) ? __builtin_nextafterf((float)(1.0 / (double)ratio), __builtin_inff()) : 0.0f;
	const unsigned cap = slow > 0 ? cha_rs_cap(chunk, fast, slow) : 0u;
#define GO(EV, D) do { \
		hipLaunchKernelGGL((k_cha_track<EV, D>), grid, dim3(256), 0, s, in_rows, in_stride, W, a, mg, bias, warm, chunk, n_chunks, audio_in, ctab); \
		hipLaunchKernelGGL(k_cha_walk, dim3((unsigned)n_channels), dim3(64 * n_seg), (size_t)n_chunks * 20, s, ctab, n_chunks, audio_in, audio_out, seg_start); \
		if (slow > 0) \
			hipLaunchKernelGGL((k_cha_replay_rs<EV, D>), grid, dim3(256), (size_t)cap * 2, s, in_rows, in_stride, W, a, mg, bias, chunk, n_chunks, seg_start, \
			                   fast, slow, ratio, rinv, audio_in, audio_out, out_rows, out_stride, cap); \
		else \
			hipLaunchKernelGGL((k_cha_replay<EV, D>), grid, dim3(256), 0, s, in_rows, in_stride, W, a, mg, bias, chunk, n_chunks, seg_start, \
			                   audio_in, audio_out, out_rows, out_stride); } while (0)
	if (deemph_d24u(a)) { if (a & 1) GO(false, true); else GO(true, true); }
	else { if (a & 1) GO(false, false); else GO(true, false); }
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_ch_audio(void *stream, int16_t *rows, u64 row_stride, u64 W, int n_channels, int deemph, int a, int warm, int serial,
                            int fast, int slow, u64 J, const int *audio_in, int *audio_out, int16_t *y_rows, u64 y_stride)
{
	if (!W || !n_channels)
		return 0;
	hipStream_t s = (hipStream_t)stream;
	const unsigned mg = deemph ? deemph_magic_u(a) : 0u;
	const int bias = deemph ? bias_for(a) : 0;
#define GO(EV, D) hipLaunchKernelGGL((k_ch_audio<EV, D>), dim3((unsigned)n_channels), dim3(256), 0, s, rows, row_stride, W, deemph, a, mg, bias, warm, serial, \
		fast, slow, J, audio_in, audio_out, y_rows, y_stride)
	if (deemph && deemph_d24u(a)) { if (a & 1) GO(false, true); else GO(true, true); }
	else { if (!deemph || (a & 1)) GO(false, false); else GO(true, false); }
#undef GO
	LAUNCH_RET();
}

extern "C" int rxk_fm_droop_disc(void *stream, const uint32_t *in, u64 M, const int *fir, const int16_t *hist_in, int16_t *hist_out,
                                 uint32_t *lp_out, u64 uniform_k, int16_t *pcm, int pcm_chl2, rxk_fm_dev *dev, rxk_flag_rec *flag_list,
                                 int *flag_cnt, int flag_all)
{
	if (!M)
		return 0;
	const unsigned grid = (unsigned)(((M + 3) / 4 + 255) / 256);
	hipStream_t s = (hipStream_t)stream;
	if (lp_out)
		hipLaunchKernelGGL((k_fm_droop_disc<true>), dim3(grid), dim3(256), 0, s, in, M, fir, hist_in, hist_out, lp_out, uniform_k, pcm, pcm_chl2, dev,
		                   flag_list, flag_cnt, flag_all);
	else
		hipLaunchKernelGGL((k_fm_droop_disc<false>), dim3(grid), dim3(256), 0, s, in, M, fir, hist_in, hist_out, lp_out, uniform_k, pcm, pcm_chl2, dev,
		                   flag_list, flag_cnt, flag_all);
	LAUNCH_RET();
}

// rx_power's downsample_iq (rtl_power.c:656-662): `fuse` (1..3) stateless fifth_order passes over n_bufs independent buffers of n
// complex samples (n % RXK_FIFTH_TILE == 0) in one LDS-tiled launch; strides in complex samples
extern "C" int rxk_pw_fifth_fused(void *stream, const int16_t *in, unsigned long long n_bufs, unsigned n, unsigned in_stride, int fuse,
                                  int16_t *out, unsigned out_stride)
{
	hipStream_t s = (hipStream_t)stream;
	const unsigned tiles = n / FF_RAW;
	const unsigned tpw = tiles % 16 == 0 ? 16 : tiles % 8 == 0 ? 8 : tiles % 4 == 0 ? 4 : tiles % 2 == 0 ? 2 : 1;
	const unsigned grid = (unsigned)(n_bufs * (tiles / tpw));
	const uint32_t *p = (const uint32_t *)in;
	uint32_t *o = (uint32_t *)out;
#define FUSED(F) hipLaunchKernelGGL((k_fm_fifth_fused<F, false, true, true>), dim3(grid), dim3(256), 0, s, p, n, tiles, tpw, nullptr, o, in_stride, out_stride)
	if (fuse == 1) FUSED(1); else if (fuse == 2) FUSED(2); else FUSED(3);
#undef FUSED
	LAUNCH_RET();
}

// four stateless passes (+ the droop FIR: fir = cic_9_tables[4] on the device, or NULL) in registers, the buffers' dc sums accumulated into sums[2 * buffer],
// [2 * buffer + 1] (zeroed by the caller; NULL: none).  n % 64 == 0, n >= 256; strides in complex samples, multiples of 4; out 16-byte aligned
// wave_part (sums != NULL): rxk_pw_fifth_regn4_parts(n_bufs, n) int pairs of scratch, one per wave
extern "C" unsigned long long rxk_pw_fifth_regn4_parts(unsigned long long n_bufs, unsigned n)
{
	const unsigned tiles_r = ((n >> 4) / 4 + FR_OUT - 1) / FR_OUT;
	const unsigned twn = tiles_r >= 4 * 2 ? 2 : 1;
	return n_bufs * (unsigned long long)((tiles_r + 4 * twn - 1) / (4 * twn)) * 4ull;
}

extern "C" int rxk_pw_fifth_regn4(void *stream, const int16_t *in, unsigned long long n_bufs, unsigned n, unsigned in_stride, const int *fir_dev, const int *fir_host,
                                  int16_t *out, unsigned out_stride, long long *sums, int *wave_part)
{
	hipStream_t s = (hipStream_t)stream;
	constexpr int LV = 4, TW = 2;
	if (sums && !wave_part)
		return (int)hipErrorInvalidValue;
	const unsigned tiles_r = ((n >> LV) / 4 + FR_OUT - 1) / FR_OUT;
	const unsigned twn = tiles_r >= 4 * TW ? TW : 1;
	const unsigned wgs_per_block = (tiles_r + 4 * twn - 1) / (4 * twn);
	const u64 total = n_bufs * (u64)wgs_per_block;
	if (total > 0xfffffff0ull || n_bufs > 0x7fffffffull)
		return (int)hipErrorInvalidValue;
	const unsigned rgrid = (unsigned)((total + 7) / 8 * 8);
	const uint32_t *p = (const uint32_t *)in;
	uint32_t *o = (uint32_t *)out;
	const unsigned fgrid = (unsigned)((n_bufs + 3) / 4);
	const int f1 = fir_host ? fir_host[1] : 0, f2 = fir_host ? fir_host[2] : 0, f3 = fir_host ? fir_host[3] : 0, f4 = fir_host ? fir_host[4] : 0, f5 = fir_host ? fir_host[5] : 0;
#define PWR(FI, T) hipLaunchKernelGGL((k_pw_fifth_regn<LV, FI, T>), dim3(rgrid), dim3(256), 0, s, p, n, in_stride, tiles_r, wgs_per_block, (unsigned)total, o, out_stride, \
		                              f1, f2, f3, f4, f5, sums ? (int2 *)wave_part : (int2 *)nullptr)
	if (fir_host) {
		if (twn == TW) PWR(true, TW); else PWR(true, 1);
		hipLaunchKernelGGL((k_pw_fifth_fix<LV, true>), dim3(fgrid), dim3(256), 0, s, p, (unsigned)n_bufs, in_stride, o, out_stride, fir_dev, (i64 *)sums, (const int2 *)wave_part, wgs_per_block * 4u);
	} else {
		if (twn == TW) PWR(false, TW); else PWR(false, 1);
		hipLaunchKernelGGL((k_pw_fifth_fix<LV, false>), dim3(fgrid), dim3(256), 0, s, p, (unsigned)n_bufs, in_stride, o, out_stride, fir_dev, (i64 *)sums, (const int2 *)wave_part, wgs_per_block * 4u);
	}
#undef PWR
	LAUNCH_RET();
}
