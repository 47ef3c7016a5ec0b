/* rx_oracle.c -- CPU restatement of the rx_tools sample-stream DSP path.
 *
 * TEST INFRASTRUCTURE ONLY (see rx_oracle.h): the checker, never the product.
 * Plain C, explicit state, no globals; written from the behaviour of the reference
 * (citations are file:line under /root/reference/src) and pinned against the
 * reference's own code compiled unmodified (oracle/_ref) -- tests/test_oracle_vs_ref.py
 * and the tests/golden fixtures.
 *
 * Build with -fwrapv -ffp-contract=off (oracle/Makefile): signed overflow must wrap
 * like the reference's does in practice, and the fp64 expressions must not be fused.
 */
#include "rx_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int16_t wrap16(int v) { return (int16_t)(uint16_t)(unsigned)v; }

/* ========================================================================== rx_fm */

/* rtl_fm.c:846 -- `s->buf16[i] = ( (int16_t)buf[i] / 32767.0 * 128.0 + 0.4);`
 * evaluated in double, converted to int16 by C truncation toward zero. */
int16_t rxo_scale_sample(int16_t x)
{
	double v = (double)x / 32767.0;
	v = v * 128.0;
	v = v + 0.4;
	return (int16_t)v;
}

/* rtl_fm.c:309-327 -- multiply complex sample n of the block by j^n:
 * n%4==0 (I,Q) ; 1 (-Q,I) ; 2 (-I,-Q) ; 3 (Q,-I).  The reference walks groups of
 * 8 int16; a trailing group of 2/4/6 int16 is handled by the same per-pair rule
 * because each pair only touches its own two slots. */
void rxo_rotate_90(int16_t *buf, uint32_t len)
{
	for (uint32_t p = 0; p + 1 < len; p += 2) {
		int16_t i = buf[p], q = buf[p + 1];
		switch ((p >> 1) & 3) {
		case 0: break;
		case 1: buf[p] = wrap16(-q); buf[p + 1] = i; break;
		case 2: buf[p] = wrap16(-i); buf[p + 1] = wrap16(-q); break;
		case 3: buf[p] = q; buf[p + 1] = wrap16(-i); break;
		}
	}
}

/* rtl_fm.c:351-371 -- boxcar sum over `downsample` complex samples with the running
 * sums and the fill count carried across calls; the int sums are stored into the
 * int16 buffer (wrap), no scaling. */
int rxo_low_pass(int16_t *lp, int lp_len, int downsample, int *now_r, int *now_j, int *prev_index)
{
	int out = 0;
	for (int in = 0; in < lp_len; in += 2) {
		*now_r += lp[in];
		*now_j += lp[in + 1];
		*prev_index += 1;
		if (*prev_index < downsample)
			continue;
		lp[out] = wrap16(*now_r);
		lp[out + 1] = wrap16(*now_j);
		out += 2;
		*prev_index = 0;
		*now_r = 0;
		*now_j = 0;
	}
	return out;
}

/* rtl_fm.c:411-440.  On the strided sequence s_k = data[2k] the k-th output is
 *   (s[2k-5] + 5 (s[2k-4] + s[2k-1]) + 10 (s[2k-3] + s[2k-2]) + s[2k]) >> 4
 * where s[-5..-1] = hist[1..5]; outputs exist for 4k < length.  The operands are
 * int16 (each shuffle through a..f truncates nothing new), the sum is int, the
 * arithmetic shift result is truncated to int16 on store.  hist[0..5] afterwards is
 * the last window. */
void rxo_fifth_order_fm(int16_t *data, int length, int16_t hist[6])
{
	int16_t w[6];
	w[0] = hist[1]; w[1] = hist[2]; w[2] = hist[3]; w[3] = hist[4]; w[4] = hist[5];
	w[5] = data[0];
	data[0] = wrap16((w[0] + (w[1] + w[4]) * 5 + (w[2] + w[3]) * 10 + w[5]) >> 4);
	for (int pos = 4; pos < length; pos += 4) {
		w[0] = w[2]; w[1] = w[3]; w[2] = w[4]; w[3] = w[5];
		w[4] = data[pos - 2];
		w[5] = data[pos];
		data[pos / 2] = wrap16((w[0] + (w[1] + w[4]) * 5 + (w[2] + w[3]) * 10 + w[5]) >> 4);
	}
	memcpy(hist, w, sizeof(w));
}

/* rtl_fm.c:288-300 */
static const int cic9[11][10] = {
	{0},
	{9, -156,  -97, 2798, -15489, 61019, -15489, 2798,  -97, -156},
	{9, -128, -568, 5593, -24125, 74126, -24125, 5593, -568, -128},
	{9, -129, -639, 6187, -26281, 77511, -26281, 6187, -639, -129},
	{9, -122, -612, 6082, -26353, 77818, -26353, 6082, -612, -122},
	{9, -120, -602, 6015, -26269, 77757, -26269, 6015, -602, -120},
	{9, -120, -582, 5951, -26128, 77542, -26128, 5951, -582, -120},
	{9, -119, -580, 5931, -26094, 77505, -26094, 5931, -580, -119},
	{9, -119, -578, 5921, -26077, 77484, -26077, 5921, -578, -119},
	{9, -119, -577, 5917, -26067, 77473, -26067, 5917, -577, -119},
	{9, -199, -362, 5303, -25505, 77489, -25505, 5303, -362, -199},
};
const int *rxo_cic9_table(int passes) { return (passes >= 0 && passes <= 10) ? cic9[passes] : cic9[0]; }

/* rtl_fm.c:442-465 -- 9-tap symmetric FIR on one interleaved half, output from the
 * history BEFORE the new sample is shifted in; int sum, >>15, int16 store. */
void rxo_generic_fir_fm(int16_t *data, int length, const int *fir, int16_t hist[9])
{
	for (int d = 0; d < length; d += 2) {
		int16_t in = data[d];
		int acc = (hist[0] + hist[8]) * fir[1] + (hist[1] + hist[7]) * fir[2]
		        + (hist[2] + hist[6]) * fir[3] + (hist[3] + hist[5]) * fir[4]
		        + hist[4] * fir[5];
		data[d] = wrap16(acc >> 15);
		memmove(hist, hist + 1, 8 * sizeof(int16_t));
		hist[8] = in;
	}
}

/* rtl_fm.c:485-506 -- pi == 1<<14.  The product pi4*(x -/+ |y|) is a wrapping int32
 * multiply; the division is C's truncating signed division. */
int rxo_fast_atan2(int y, int x)
{
	const int q = 1 << 12;
	int ay, ang;
	if (x == 0 && y == 0)
		return 0;
	ay = y < 0 ? -y : y;
	if (x >= 0)
		ang = q - (int)((unsigned)q * (unsigned)(x - ay)) / (x + ay);
	else
		ang = 3 * q - (int)((unsigned)q * (unsigned)(x + ay)) / (ay - x);
	return y < 0 ? -ang : ang;
}

/* rtl_fm.c:470-474 with b conjugated, as both discriminators call it (480, 511) */
static void mul_conj(int ar, int aj, int br, int bj, int *cr, int *cj)
{
	*cr = (int)((unsigned)ar * (unsigned)br - (unsigned)aj * (unsigned)(-bj));
	*cj = (int)((unsigned)aj * (unsigned)br + (unsigned)ar * (unsigned)(-bj));
}

/* rtl_fm.c:508-513 */
int rxo_polar_disc_fast(int ar, int aj, int br, int bj)
{
	int cr, cj;
	mul_conj(ar, aj, br, bj, &cr, &cj);
	return rxo_fast_atan2(cj, cr);
}

/* rtl_fm.c:476-483 */
int rxo_polar_discriminant(int ar, int aj, int br, int bj)
{
	int cr, cj;
	double angle;
	mul_conj(ar, aj, br, bj, &cr, &cj);
	angle = atan2((double)cj, (double)cr);
	return (int)(angle / 3.14159 * (1 << 14));
}

/* rtl_fm.c:515-526: atan_lut[i] = (int)(atan(i / 2^8) / 3.14159 * 2^14), 131072 entries */
static int *lut_table(void)
{
	static int *lut;
	if (!lut) {
		lut = malloc(131072 * sizeof(int));
		for (int i = 0; i < 131072; i++)
			lut[i] = (int)(atan((double)i / (1 << 8)) / 3.14159 * (1 << 14));
	}
	return lut;
}

/* rtl_fm.c:528-564 */
int rxo_polar_disc_lut(int ar, int aj, int br, int bj)
{
	int cr, cj, x, xa;
	const int *lut = lut_table();
	mul_conj(ar, aj, br, bj, &cr, &cj);
	if (cr == 0 || cj == 0) {
		if (cr == 0 && cj == 0) return 0;
		if (cr == 0 && cj > 0) return 1 << 13;
		if (cr == 0 && cj < 0) return -(1 << 13);
		if (cj == 0 && cr > 0) return 0;
		if (cj == 0 && cr < 0) return 1 << 14;
	}
	x = (int)((unsigned)cj << 8) / cr;
	xa = x < 0 ? -x : x;
	if (xa >= 131072)
		return cj > 0 ? 1 << 13 : -(1 << 13);
	if (x > 0)
		return cj > 0 ? lut[x] : lut[x] - (1 << 14);
	return cj > 0 ? (1 << 14) - lut[-x] : -lut[-x];
}

/* rtl_fm.c:566-582 */
int rxo_esbensen(int ar, int aj, int br, int bj)
{
	int dr = (int)(((unsigned)br - (unsigned)ar) * 2u);
	int dj = (int)(((unsigned)bj - (unsigned)aj) * 2u);
	int cj = (int)((unsigned)bj * (unsigned)dr - (unsigned)br * (unsigned)dj);
	int den = (int)((unsigned)ar * (unsigned)ar + (unsigned)aj * (unsigned)aj + 1u);
	return (int)(2608u * (unsigned)cj) / den;
}

/* rtl_fm.c:739-757 */
int rxo_rms(const int16_t *samples, int len, int step)
{
	long p = 0, t = 0;
	for (int i = 0; i < len; i += step) {
		long s = samples[i];
		t += s;
		p += s * s;
	}
	double dc = (double)(t * step) / (double)len;
	double err = (double)(t * 2) * dc - dc * dc * len;
	return (int)sqrt(((double)p - err) / len);
}

/* rtl_fm.c:617-665 */
int rxo_simple_demod(int mode, const int16_t *lp, int lp_len, int output_scale, int16_t *result)
{
	if (mode == 4) {
		memcpy(result, lp, (size_t)lp_len * sizeof(int16_t));
		return lp_len;
	}
	for (int i = 0; i < lp_len; i += 2) {
		int pcm;
		if (mode == 1) {
			pcm = lp[i] * lp[i] + lp[i + 1] * lp[i + 1];
			result[i / 2] = wrap16((int16_t)sqrt(pcm) * output_scale);
		} else {
			pcm = mode == 2 ? lp[i] + lp[i + 1] : lp[i] - lp[i + 1];
			result[i / 2] = wrap16((int16_t)pcm * output_scale);
		}
	}
	return lp_len / 2;
}

/* rtl_fm.c:684-697 */
void rxo_dc_block_audio(int16_t *result, int n, int adc_block_const, int *dc_avg)
{
	int64_t sum = 0;
	for (int i = 0; i < n; i++)
		sum += result[i];
	int avg = (int)(sum / n);
	avg = (avg + *dc_avg * adc_block_const) / (adc_block_const + 1);
	for (int i = 0; i < n; i++)
		result[i] = wrap16(result[i] - avg);
	*dc_avg = avg;
}

/* rtl_fm.c:584-615 -- sample 0 of every call goes through the libm discriminator
 * against the carried previous sample; the rest use the selected one (only 0 and 1 are
 * on the path this oracle covers).  pre_r/pre_j <- last sample. */
int rxo_fm_demod(const int16_t *lp, int lp_len, int custom_atan, int *pre_r, int *pre_j, int16_t *result)
{
	result[0] = wrap16(rxo_polar_discriminant(lp[0], lp[1], *pre_r, *pre_j));
	for (int i = 2; i < lp_len - 1; i += 2) {
		int pcm = custom_atan == 1 ? rxo_polar_disc_fast(lp[i], lp[i + 1], lp[i - 2], lp[i - 1])
		        : custom_atan == 2 ? rxo_polar_disc_lut(lp[i], lp[i + 1], lp[i - 2], lp[i - 1])
		        : custom_atan == 3 ? rxo_esbensen(lp[i], lp[i + 1], lp[i - 2], lp[i - 1])
		        : rxo_polar_discriminant(lp[i], lp[i + 1], lp[i - 2], lp[i - 1]);
		result[i / 2] = wrap16(pcm);
	}
	*pre_r = lp[lp_len - 2];
	*pre_j = lp[lp_len - 1];
	return lp_len / 2;
}

/* rtl_fm.c:667-682 -- avg += round-half-away((x - avg) / a) with C truncating
 * division and a/2 as integer; the state is one int shared by the whole process in
 * the reference (function static), explicit here. */
void rxo_deemph(int16_t *result, int n, int a, int *avg)
{
	int h = a / 2;
	for (int i = 0; i < n; i++) {
		int d = result[i] - *avg;
		*avg += (d > 0) ? (d + h) / a : (d - h) / a;
		result[i] = wrap16(*avg);
	}
}

/* rtl_fm.c:389-409 -- fractional boxcar: accumulate, advance a phase by rate_out2 per
 * input, emit sum / (rate_out / rate_out2) (integer ratio) each time the phase reaches
 * rate_out. */
int rxo_low_pass_real(int16_t *result, int n, int rate_out, int rate_out2, int *now_lpr, int *prev_lpr_index)
{
	int out = 0;
	int ratio = rate_out / rate_out2;
	for (int i = 0; i < n; i++) {
		*now_lpr += result[i];
		*prev_lpr_index += rate_out2;
		if (*prev_lpr_index < rate_out)
			continue;
		result[out++] = wrap16(*now_lpr / ratio);
		*prev_lpr_index -= rate_out;
		*now_lpr = 0;
	}
	return out;
}

/* full_demod, rtl_fm.c:759-824, restricted to the fm/wbfm path with squelch, level
 * printing, post_downsample and dc_block_audio off (their defaults, 1084-1115). */
int rxo_fm_full_demod(rxo_fm_state *st, int16_t *lp, int *lp_len, int16_t *out)
{
	int n;
	if (st->downsample_passes) {
		int p = st->downsample_passes;
		for (int i = 0; i < p; i++) {
			rxo_fifth_order_fm(lp, *lp_len >> i, st->lp_i_hist[i]);
			rxo_fifth_order_fm(lp + 1, (*lp_len >> i) - 1, st->lp_q_hist[i]);
		}
		*lp_len >>= p;
		if (st->comp_fir_size == 9 && p <= 10) {
			rxo_generic_fir_fm(lp, *lp_len, rxo_cic9_table(p), st->droop_i_hist);
			rxo_generic_fir_fm(lp + 1, *lp_len - 1, rxo_cic9_table(p), st->droop_q_hist);
		}
	} else {
		*lp_len = rxo_low_pass(lp, *lp_len, st->downsample, &st->now_r, &st->now_j, &st->prev_index);
	}
	if (st->squelch_level) {                                           /* rtl_fm.c:781-790 */
		if (rxo_rms(lp, *lp_len, 1) < st->squelch_level) {
			st->squelch_hits++;
			memset(lp, 0, (size_t)*lp_len * sizeof(int16_t));
		} else {
			st->squelch_hits = 0;
		}
	}
	if (st->mode == 0) {
		n = rxo_fm_demod(lp, *lp_len, st->custom_atan, &st->pre_r, &st->pre_j, out);
	} else {
		n = rxo_simple_demod(st->mode, lp, *lp_len, st->output_scale, out);
		if (st->mode == 4)
			return n;                                                  /* rtl_fm.c:809-811 */
	}
	if (st->post_downsample > 1)
		n = rxo_low_pass_simple(out, n, st->post_downsample);              /* rtl_fm.c:814-815 */
	if (st->deemph)
		rxo_deemph(out, n, st->deemph_a, &st->deemph_avg);
	if (st->dc_block_audio)
		rxo_dc_block_audio(out, n, st->adc_block_const, &st->dc_avg);
	if (st->rate_out2 > 0)
		n = rxo_low_pass_real(out, n, st->rate_out, st->rate_out2, &st->now_lpr, &st->prev_lpr_index);
	return n;
}

/* rtl_fm.c:373-387.  `len` must be a multiple of `step` (the reference says so and reads past the end otherwise);
 * the write of signal2[len/step + 1] is kept. */
int rxo_low_pass_simple(int16_t *signal2, int len, int step)
{
	int i, i2, sum;
	for (i = 0; i < len; i += step) {
		sum = 0;
		for (i2 = 0; i2 < step; i2++)
			sum += (int)signal2[i + i2];
		signal2[i / step] = (int16_t)sum;
	}
	signal2[i / step + 1] = signal2[i / step];
	return len / step;
}

/* rtl_fm.c:699-721 */
void rxo_dc_block_raw(int16_t *buf, int len, int rdc_block_const, int *dc_avgI, int *dc_avgQ)
{
	int64_t sumI = 0, sumQ = 0;
	int avgI, avgQ;
	for (int i = 0; i < len; i += 2) {
		sumI += buf[i];
		sumQ += buf[i + 1];
	}
	avgI = (int)(sumI / (len / 2));
	avgQ = (int)(sumQ / (len / 2));
	avgI = (avgI + *dc_avgI * rdc_block_const) / (rdc_block_const + 1);
	avgQ = (avgQ + *dc_avgQ * rdc_block_const) / (rdc_block_const + 1);
	for (int i = 0; i < len; i += 2) {
		buf[i] = (int16_t)(buf[i] - avgI);
		buf[i + 1] = (int16_t)(buf[i + 1] - avgQ);
	}
	*dc_avgI = avgI;
	*dc_avgQ = avgQ;
}

/* rtlsdr_callback pre-stage, rtl_fm.c:839-857, then full_demod */
int rxo_fm_block(rxo_fm_state *st, const int16_t *in, int len, int16_t *lp, int *lp_len_out, int16_t *out)
{
	int lp_len = len, n;
	for (int i = 0; i < len; i++)
		lp[i] = rxo_scale_sample((st->mute && i < st->mute) ? 0 : in[i]);
	st->mute = 0;
	if (st->dc_block_raw)                                                  /* rtl_fm.c:850-852 */
		rxo_dc_block_raw(lp, len, st->rdc_block_const, &st->dc_avgI, &st->dc_avgQ);
	if (!st->offset_tuning)
		rxo_rotate_90(lp, (uint32_t)len);
	n = rxo_fm_full_demod(st, lp, &lp_len, out);
	if (lp_len_out)
		*lp_len_out = lp_len;
	return n;
}

long rxo_fm_stream(rxo_fm_state *st, const int16_t *in, size_t n_blocks, int block_len,
                   int16_t *out, int *per_block_len)
{
	int16_t *lp = malloc((size_t)block_len * sizeof(int16_t));
	int16_t *res = malloc((size_t)block_len * sizeof(int16_t));
	long total = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		int n = rxo_fm_block(st, in + b * (size_t)block_len, block_len, lp, NULL, res);
		memcpy(out + total, res, (size_t)n * sizeof(int16_t));
		if (per_block_len)
			per_block_len[b] = n;
		total += n;
	}
	free(lp);
	free(res);
	return total;
}

/* ======================================================================= rx_power */

/* rtl_power.c:240-254 */
void rxo_sine_table(int log2n, int16_t *sinewave)
{
	int n = 1 << log2n;
	for (int i = 0; i < n * 3 / 4; i++) {
		double d = (double)i * 2.0 * M_PI / n;
		sinewave[i] = (int16_t)(int)round(32767 * sin(d));
	}
}

/* rtl_power.c:256-262 -- ((a*b >> 14) + 1) >> 1 written as the reference does
 * (low bit added back), result truncated to int16 by the return type. */
int16_t rxo_fix_mpy(int16_t a, int16_t b)
{
	int c = ((int)a * (int)b) >> 14;
	return wrap16((c >> 1) + (c & 1));
}

static unsigned bitrev(unsigned v, int bits)
{
	unsigned r = 0;
	for (int i = 0; i < bits; i++)
		r |= ((v >> i) & 1u) << (bits - 1 - i);
	return r;
}

/* rtl_power.c:264-320 -- in-place radix-2 decimation-in-time on interleaved int16
 * IQ: bit-reversal permutation, then m stages; every stage halves its inputs
 * (`shift` is always 1, 294), twiddles are the sine table halved AFTER negation
 * (299-301), each of the four products is rounded by FIX_MPY and every store
 * truncates to int16. */
int rxo_fix_fft(int16_t *iq, int m, const int16_t *sinewave)
{
	int n = 1 << m;
	for (int a = 1; a < n; a++) {          /* rtl_power.c:275-290 */
		int b = (int)bitrev((unsigned)a, m);
		if (b <= a)
			continue;
		int16_t t;
		t = iq[2 * a]; iq[2 * a] = iq[2 * b]; iq[2 * b] = t;
		t = iq[2 * a + 1]; iq[2 * a + 1] = iq[2 * b + 1]; iq[2 * b + 1] = t;
	}
	for (int s = 0; s < m; s++) {          /* rtl_power.c:293-318 */
		int half = 1 << s, k = m - 1 - s;
		for (int t = 0; t < half; t++) {
			int j = t << k;
			int16_t wr = sinewave[j + n / 4];
			int16_t wi = wrap16(-sinewave[j]);
			wr >>= 1;
			wi >>= 1;
			for (int lo = t; lo < n; lo += 2 * half) {
				int hi = lo + half;
				int16_t tr = wrap16(rxo_fix_mpy(wr, iq[2 * hi]) - rxo_fix_mpy(wi, iq[2 * hi + 1]));
				int16_t ti = wrap16(rxo_fix_mpy(wr, iq[2 * hi + 1]) + rxo_fix_mpy(wi, iq[2 * hi]));
				int16_t qr = iq[2 * lo] >> 1;
				int16_t qi = iq[2 * lo + 1] >> 1;
				iq[2 * hi] = wrap16(qr - tr);
				iq[2 * hi + 1] = wrap16(qi - ti);
				iq[2 * lo] = wrap16(qr + tr);
				iq[2 * lo + 1] = wrap16(qi + ti);
			}
		}
	}
	return 0;
}

/* rtl_power.c:609-624 -- sum every other element over `length` int16 but divide by
 * `length` (not length/2): about half the mean is removed; nothing if it rounds to 0. */
void rxo_remove_dc(int16_t *data, int length)
{
	int64_t sum = 0;
	for (int i = 0; i < length; i += 2)
		sum += data[i];
	int16_t ave = (int16_t)(sum / (int64_t)length);
	if (ave == 0)
		return;
	for (int i = 0; i < length; i += 2)
		data[i] = wrap16(data[i] - ave);
}

/* rtl_power.c:582-607 -- stateless: three eased-in outputs from the first six
 * samples, then the same [1,5,10,10,5,1]>>4 window as rx_fm's, int temporaries. */
void rxo_fifth_order_power(int16_t *data, int length)
{
	int a = data[0], b = data[2], c = data[4], d = data[6], e = data[8], f = data[10];
	data[0] = wrap16(((a + b) * 10 + (c + d) * 5 + d + f) >> 4);
	data[2] = wrap16(((b + c) * 10 + (a + d) * 5 + e + f) >> 4);
	data[4] = wrap16((a + (b + e) * 5 + (c + d) * 10 + f) >> 4);
	for (int pos = 12; pos < length; pos += 4) {
		a = c; b = d; c = e; d = f;
		e = data[pos - 2];
		f = data[pos];
		data[pos / 2] = wrap16((a + (b + e) * 5 + (c + d) * 10 + f) >> 4);
	}
}

/* rtl_power.c:626-654 -- first nine samples pass through and seed the history */
void rxo_generic_fir_power(int16_t *data, int length, const int *fir)
{
	int hist[9];
	for (int k = 0; k < 9; k++)
		hist[k] = data[2 * k];
	for (int d = 18; d < length; d += 2) {
		int in = data[d];
		int acc = (hist[0] + hist[8]) * fir[1] + (hist[1] + hist[7]) * fir[2]
		        + (hist[2] + hist[6]) * fir[3] + (hist[3] + hist[5]) * fir[4]
		        + hist[4] * fir[5];
		data[d] = wrap16(acc >> 15);
		memmove(hist, hist + 1, 8 * sizeof(int));
		hist[8] = in;
	}
}

/* rtl_power.c:322-401 window shapes, 1034-1037 quantisation */
static double win_value(const char *name, int i, int length)
{
	double n1 = (double)(length - 1);
	if (!strcmp(name, "hamming"))
		return 25.0 / 46.0 - (21.0 / 46.0) * cos(2 * i * M_PI / n1);
	if (!strcmp(name, "blackman"))
		return 7938.0 / 18608.0 - (9240.0 / 18608.0) * cos(2 * i * M_PI / n1)
		     + (1430.0 / 18608.0) * cos(4 * i * M_PI / n1);
	if (!strcmp(name, "blackman-harris") || !strcmp(name, "youssef")) {
		double w = 0.35875 - 0.48829 * cos(2 * i * M_PI / n1) + 0.14128 * cos(4 * i * M_PI / n1)
		         - 0.01168 * cos(6 * i * M_PI / n1);
		if (name[0] == 'y')
			w *= pow(M_E, (-0.0025 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
		return w;
	}
	if (!strcmp(name, "hann-poisson"))
		return 0.5 * (1 - cos(2 * M_PI * i / n1)) * pow(M_E, (-2.0 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
	if (!strcmp(name, "bartlett")) {
		double w = (i - n1 / 2) / ((double)length / 2);
		if (w < 0)
			w = -w;
		return 1 - w;
	}
	return 1.0;   /* rectangle, kaiser */
}

int rxo_window_coefs(const char *name, int length, int *coefs)
{
	for (int i = 0; i < length; i++)
		coefs[i] = (int)(256 * win_value(name, i, length));
	return 0;
}

/* rtl_power.c:403-429 */
void rxo_rms_power(const int16_t *buf, int buf_len, int peak_hold, int64_t *avg0, int *samples)
{
	int64_t p = 0, t = 0;
	for (int i = 0; i < buf_len; i++) {
		int s = buf[i];
		t += s;
		p += (int64_t)s * s;
	}
	double dc = (double)t / (double)buf_len;
	double err = (double)(t * 2) * dc - dc * dc * buf_len;
	p -= (int64_t)round(err);
	if (!peak_hold)
		*avg0 += p;
	else if (p > *avg0)
		*avg0 = p;
	*samples += 1;
}

/* the body of scanner()'s per-tune loop after the read, rtl_power.c:709-770 */
void rxo_power_tune(const rxo_power_cfg *cfg, const int16_t *buf16, int16_t *work, int64_t *avg, int *samples)
{
	int n = 1 << cfg->bin_e, ds = cfg->downsample, len = cfg->buf_len;
	if (n == 1) {
		rxo_rms_power(buf16, len, cfg->peak_hold, &avg[0], samples);
		return;
	}
	memcpy(work, buf16, (size_t)len * sizeof(int16_t));                 /* 715-720 */
	if (cfg->boxcar && ds > 1) {                                        /* 723-733 */
		int src = 2, dst = 0;
		while (src < len) {
			work[dst] = wrap16(work[dst] + work[src]);
			work[dst + 1] = wrap16(work[dst + 1] + work[src + 1]);
			work[src] = 0;
			work[src + 1] = 0;
			src += 2;
			if (src % (ds * 2) == 0)
				dst += 2;
		}
	} else if (cfg->downsample_passes) {                                /* 734-743 */
		int j;
		for (j = 0; j < cfg->downsample_passes; j++) {
			rxo_fifth_order_power(work, len >> j);
			rxo_fifth_order_power(work + 1, (len >> j) - 1);
		}
		if (cfg->comp_fir_size == 9 && cfg->downsample_passes <= 10) {
			rxo_generic_fir_power(work, len >> j, rxo_cic9_table(cfg->downsample_passes));
			rxo_generic_fir_power(work + 1, (len >> j) - 1, rxo_cic9_table(cfg->downsample_passes));
		}
	}
	rxo_remove_dc(work, len / ds);                                      /* 744-745 */
	rxo_remove_dc(work + 1, len / ds - 1);
	for (int off = 0; off < len / ds; off += 2 * n) {                   /* 747-770 */
		for (int j = 0; j < n; j++) {
			work[off + 2 * j] = wrap16((int32_t)work[off + 2 * j] * cfg->window_coefs[j]);
			work[off + 2 * j + 1] = wrap16((int32_t)work[off + 2 * j + 1] * cfg->window_coefs[j]);
		}
		rxo_fix_fft(work + off, cfg->bin_e, cfg->sinewave);
		for (int j = 0; j < n; j++) {
			int64_t re = work[off + 2 * j], im = work[off + 2 * j + 1];
			int64_t pw = re * re + im * im;                             /* real_conj 664-668 */
			if (!cfg->peak_hold)
				avg[j] += pw;
			else if (pw > avg[j])
				avg[j] = pw;
		}
		*samples += ds;
	}
}

/* rtl_power.c:774-817 minus the strftime prefix added by main (1046-1048) */
int rxo_csv_row(char *dst, size_t cap, int64_t freq, int rate, int bin_e, int downsample, double crop,
                int64_t *avg, int *samples)
{
	int len = 1 << bin_e, ds = downsample;
	size_t pos = 0;
	if (bin_e > 0) {
		avg[0] = avg[1];
		for (int i = 0; i < len / 2; i++) {
			int64_t t = avg[i];
			avg[i] = avg[i + len / 2];
			avg[i + len / 2] = t;
		}
	}
	int bin_count = (int)((double)len * (1.0 - crop));
	int bw2 = (int)(((double)rate * (double)bin_count) / (len * 2 * ds));
	pos += (size_t)snprintf(dst + pos, cap - pos, "%lli, %lli, %.2f, %i, ", (long long)freq - bw2,
	                        (long long)freq + bw2, (double)rate / (double)(len * ds), *samples);
	int i1 = 0 + (int)((double)len * crop * 0.5);
	int i2 = (len - 1) - (int)((double)len * crop * 0.5);
	for (int i = i1; i <= i2 && pos < cap; i++) {
		double dbm = (double)avg[i];
		dbm /= (double)rate;
		dbm /= (double)*samples;
		dbm = 10 * log10(dbm);
		pos += (size_t)snprintf(dst + pos, cap - pos, "%.2f, ", dbm);
	}
	double last = (double)avg[i2] / ((double)rate * (double)*samples);
	if (bin_e == 0)
		last = (double)avg[0] / ((double)rate * (double)*samples);
	last = 10 * log10(last);
	if (pos < cap)
		pos += (size_t)snprintf(dst + pos, cap - pos, "%.2f\n", last);
	for (int i = 0; i < len; i++)
		avg[i] = 0;
	*samples = 0;
	return (int)pos;
}

/* =============================================================== channeliser (extension, see rx_oracle.h) */

void rxo_chan_block(const rxo_chan_cfg *cfg, const int16_t *in, int len, int *pre, int16_t *out, size_t out_stride)
{
	const int n = 1 << cfg->bin_e, windows = len / 2 / n;
	int16_t *win = malloc((size_t)2 * n * sizeof(int16_t));
	int16_t *lp = malloc((size_t)cfg->n_channels * 2 * windows * sizeof(int16_t));
	for (int w = 0; w < windows; w++) {
		memcpy(win, in + (size_t)w * 2 * n, (size_t)2 * n * sizeof(int16_t));
		rxo_fix_fft(win, cfg->bin_e, cfg->sinewave);
		for (int c = 0; c < cfg->n_channels; c++) {
			int bin = (cfg->first_bin + c) & (n - 1);
			lp[((size_t)c * windows + w) * 2] = win[2 * bin];
			lp[((size_t)c * windows + w) * 2 + 1] = win[2 * bin + 1];
		}
	}
	for (int c = 0; c < cfg->n_channels; c++)
		rxo_fm_demod(lp + (size_t)c * windows * 2, 2 * windows, cfg->custom_atan, &pre[2 * c], &pre[2 * c + 1],
		             out + (size_t)c * out_stride);
	free(win);
	free(lp);
}

/* The channeliser's second definition, SURVEY.md section 8(f)2 to the letter: per channel an integer NCO, then the reference's low_pass
 * (rtl_fm.c:351-371) at downsample = N, fm_demod per channel.  The reference has no mixer beyond rotate16_90, so the NCO is specified
 * here from the reference's own fixed-point pieces: the block goes through the callback's scale (rtl_fm.c:845-848, no rotation), sample n
 * of a window is multiplied by e^(-j 2 pi k n / N), k = (first_bin + c) mod N, with cos / sin taken from the reference's Sinewave table
 * (rtl_power.c:240-254; the half period it holds, mirrored with a sign for the other half) and every one of the four products rounded by
 * FIX_MPY (rtl_power.c:256-262) and stored as int16; low_pass then sums N mixed samples per output in int and stores them as int16
 * (wrapping for an in-channel carrier above a quarter of full scale, as low_pass itself would at that decimation). */
static void nco_tw(const int16_t *sinewave, int n, int p, int *c, int *s)
{
	const int h = n / 2, q = p & (h - 1);                    /* cos(t) = sin(t + pi/2); the second half period is the first, negated */
	const int cc = sinewave[q + n / 4], ss = sinewave[q];
	*c = p >= h ? -cc : cc;
	*s = p >= h ? -ss : ss;
}

void rxo_chan_nco_block(const rxo_chan_cfg *cfg, const int16_t *in, int len, int *pre, int16_t *out, size_t out_stride)
{
	const int n = 1 << cfg->bin_e, windows = len / 2 / n;
	int16_t *lp = malloc((size_t)2 * windows * sizeof(int16_t) + 4);
	for (int ch = 0; ch < cfg->n_channels; ch++) {
		const int k = (cfg->first_bin + ch) & (n - 1);
		for (int w = 0; w < windows; w++) {
			int now_r = 0, now_j = 0;                          /* low_pass's accumulators: a window is exactly one output (prev_index returns to 0) */
			for (int i = 0; i < n; i++) {
				const int16_t *x = in + ((size_t)w * n + i) * 2;
				const int16_t xr = rxo_scale_sample(x[0]), xi = rxo_scale_sample(x[1]);
				int c, s;
				nco_tw(cfg->sinewave, n, (int)(((long long)k * i) & (n - 1)), &c, &s);
				const int16_t yr = (int16_t)(rxo_fix_mpy(xr, (int16_t)c) + rxo_fix_mpy(xi, (int16_t)s));
				const int16_t yi = (int16_t)(rxo_fix_mpy(xi, (int16_t)c) - rxo_fix_mpy(xr, (int16_t)s));
				now_r += yr;
				now_j += yi;
			}
			lp[2 * w] = (int16_t)now_r;
			lp[2 * w + 1] = (int16_t)now_j;
		}
		rxo_fm_demod(lp, 2 * windows, cfg->custom_atan, &pre[2 * ch], &pre[2 * ch + 1], out + (size_t)ch * out_stride);
	}
	free(lp);
}

/* =============================================================== rx_sdr output formats, rx_fm WAV header */

/* rtl_sdr.c:368-370.  The fp64 sum reaches 128 for x >= 32665, which an int8 cannot hold; the reference
 * build (gcc, x86-64) truncates the converted integer to its low byte (-128), stated here explicitly. */
void rxo_sdr_cs16_to_cs8(const int16_t *in, size_t n, int8_t *out)
{
	for (size_t i = 0; i < n; i++)
		out[i] = (int8_t)(uint8_t)(int)(in[i] / 32767.0 * 128.0 + 0.4);
}

/* rtl_sdr.c:376-378 */
void rxo_sdr_cs16_to_cu8(const int16_t *in, size_t n, uint8_t *out)
{
	for (size_t i = 0; i < n; i++)
		out[i] = (uint8_t)(in[i] / 32767.0 * 128.0 + 127.4);
}

/* rtl_sdr.c:384-386 */
void rxo_sdr_cs16_to_cf32(const int16_t *in, size_t n, float *out)
{
	for (size_t i = 0; i < n; i++)
		out[i] = in[i] * 1.0f / 32767;
}

/* rtl_sdr.c:356-363 */
void rxo_sdr_cs12_to_cs16(const uint8_t *in, size_t n_elems, int16_t *out)
{
	for (size_t i = 0; i < n_elems; i++) {
		uint8_t b0 = in[3 * i], b1 = in[3 * i + 1], b2 = in[3 * i + 2];
		out[2 * i] = (int16_t)((b1 << 12) | (b0 << 4));
		out[2 * i + 1] = (int16_t)((b2 << 8) | (b1 & 0xf0));
	}
}

/* rtl_fm.c:1174-1206 */
void rxo_wav_header(int rate, int raw_mode, uint8_t out[44])
{
	int b_rate = rate * 2, channels = 1, align = 2;
	if (raw_mode) {
		channels = 2;
		align = 4;
		b_rate *= 2;
	}
	uint8_t *p = out;
	memcpy(p, "RIFF", 4); p += 4;
	memset(p, 0xFF, 4); p += 4;
	memcpy(p, "WAVE", 4); p += 4;
	memcpy(p, "fmt ", 4); p += 4;
	*p++ = 0x10; *p++ = 0; *p++ = 0; *p++ = 0;
	*p++ = 1; *p++ = 0;
	*p++ = (uint8_t)channels; *p++ = 0;
	for (int i = 0; i < 4; i++) *p++ = (uint8_t)((rate >> (8 * i)) & 0xFF);
	for (int i = 0; i < 4; i++) *p++ = (uint8_t)((b_rate >> (8 * i)) & 0xFF);
	*p++ = (uint8_t)align; *p++ = 0;
	*p++ = 0x10; *p++ = 0;
	memcpy(p, "data", 4); p += 4;
	memset(p, 0xFF, 4);
}
