/* rxgpu_power.c -- host side of the rx_power path: range planner, tables, batched scan
 * object, the scanner()/csv_dbm() drop-ins.  C over the HIP C API; sample arithmetic is in
 * power_kernels.hip.
 *
 * Reference call sites replaced (under /root/reference/src/rtl_power.c):
 *   scanner(channel)   1040 (definition 670-772; the compute part 709-770)
 *   csv_dbm(&tunes[i]) 1049 (definition 774-817)
 * and the host-side set-up they depend on: frequency_range 431-543, sine_table 240-254,
 * window functions 322-401 with the quantisation at 1034-1037.
 */
#include "rxgpu_internal.h"
#include "rxgpu_ref_structs.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXIMUM_RATE 2800000     /* rtl_power.c:74 */
#define MINIMUM_RATE 1000000     /* rtl_power.c:75 */
#define DEFAULT_BUF_LENGTH 16384 /* rtl_power.c:71 */
#define MAX_TUNES 10000          /* rtl_power.c:111 */

/* ------------------------------------------------------------------ host tables */

/* atofs, convenience.c:65-88: a float with an optional k/K, M, G suffix */
static double parse_suffixed(const char *s)
{
	size_t len = strlen(s);
	double mult = 1.0;
	char tmp[64];
	if (!len)
		return 0.0;
	switch (s[len - 1]) {
	case 'g': case 'G': mult = 1e9; break;
	case 'm': case 'M': mult = 1e6; break;
	case 'k': case 'K': mult = 1e3; break;
	default: return atof(s);
	}
	if (len > sizeof(tmp))
		len = sizeof(tmp);
	memcpy(tmp, s, len - 1);
	tmp[len - 1] = 0;
	return atof(tmp) * mult;
}

/* ---- sweep geometry: what frequency_range (rtl_power.c:431-543) derives from "lower:upper:bin", as a closed form.
 *
 * The reference finds the hop count and the bin exponent by counting upwards until a condition holds.  Both conditions are
 * monotone in the counter -- a hop's share of the span shrinks as the hops grow, a bin narrows as the exponent grows -- so each
 * result is "the least value for which the condition holds", which a bisection over the same arithmetic expression gives
 * without walking the range (the reference's own note asks for exactly that: "todo, replace loop with algebra/log2").  Every
 * expression that decides a result keeps the reference's types and operation order (int64 quotient first, then one double
 * division by 1 - crop, truncation), because the geometry has to come out identical: tests/golden/plans.npz holds nine plans
 * produced by the reference's frequency_range, tests/test_oracle_vs_ref.py sweeps random ranges against the reference itself. */

/* least v in [lo, hi] with holds(v, ctx), for a predicate that never turns false again once true; hi + 1 if it never holds */
static int64_t least_holding(int64_t lo, int64_t hi, int (*holds)(int64_t, const void *), const void *ctx)
{
	const int64_t none = hi + 1;
	if (!holds(hi, ctx))
		return none;
	while (lo < hi) {
		const int64_t mid = lo + (hi - lo) / 2;
		if (holds(mid, ctx))
			hi = mid;
		else
			lo = mid + 1;
	}
	return lo;
}

struct hop_ctx { int64_t span; double keep; };             /* keep = 1 - crop */
/* the rate one of `hops` equal pieces of the span needs once the cropped edges are added back */
static int64_t hop_rate(int64_t hops, const struct hop_ctx *c) { return (int64_t)((double)(c->span / hops) / c->keep); }
static int hop_fits(int64_t hops, const void *ctx) { return hop_rate(hops, ctx) <= MAXIMUM_RATE; }

struct bin_ctx { int64_t rate, ds; double widest; };
static double bin_width(int64_t e, const struct bin_ctx *c) { return (double)c->rate / (double)(((int64_t)1 << e) * c->ds); }
static int bin_fits(int64_t e, const void *ctx) { return bin_width(e, ctx) <= ((const struct bin_ctx *)ctx)->widest; }

int rxgpu_power_plan_range(const char *range, double crop, int boxcar, rxgpu_power_plan *plan)
{
	enum { HOPS_TRIED = 1499, BIN_E_MAX = 21 };
	char text[192];
	if (!range || !plan || strlen(range) >= sizeof(text))
		return rxgpu_fail(RXGPU_EINVAL, "bad range string");
	strcpy(text, range);
	char *second = strchr(text, ':');
	char *third = second ? strchr(second + 1, ':') : NULL;
	if (!third)
		return rxgpu_fail(RXGPU_EINVAL, "range must be lower:upper:bin_size");
	*second++ = 0;
	*third++ = 0;
	const int64_t lower = (int64_t)parse_suffixed(text), upper = (int64_t)parse_suffixed(second);
	const int64_t widest_bin = (int64_t)parse_suffixed(third);
	const struct hop_ctx hc = { upper - lower, 1.0 - crop };

	/* hops: as few as fit the tuner's widest rate; a span no hop count up to the limit covers keeps the last count tried and no tunes */
	int64_t hops = least_holding(1, HOPS_TRIED, hop_fits, &hc);
	const int covered = hops <= HOPS_TRIED;
	int64_t piece = hc.span / (covered ? hops : HOPS_TRIED);   /* Hz each hop contributes to the sweep */
	int64_t rate = hop_rate(covered ? hops : HOPS_TRIED, &hc); /* Hz each hop samples */
	if (!covered)
		hops = 0;

	/* a span narrower than the tuner's slowest rate: one hop, oversampled by a whole factor and decimated back */
	int64_t ds = 1;
	int ds_passes = 0;
	if (rate < MINIMUM_RATE) {
		if (rate <= 0)
			return rxgpu_fail(RXGPU_EINVAL, "unsupported bandwidth");
		hops = 1;
		ds = MAXIMUM_RATE / rate;
		rate *= ds;
	}
	if (!boxcar && ds > 1) {
		/* the fifth_order cascade halves per pass: the largest power of two within the factor (floor(log2) of an integer below 2^22
		 * is its top bit whichever way it is computed) */
		ds_passes = 63 - __builtin_clzll((unsigned long long)ds);
		ds = (int64_t)1 << ds_passes;
		rate = (int)((double)(piece * ds) / hc.keep);
	}

	/* bins: the fewest (a power of two, 2 .. 2^21) that are no wider than asked; 2^21 if even those are wider */
	const struct bin_ctx bc = { rate, ds, (double)widest_bin };
	int64_t bin_e = least_holding(1, BIN_E_MAX, bin_fits, &bc);
	if (bin_e > BIN_E_MAX)
		bin_e = BIN_E_MAX;

	/* bins of a whole tuner bandwidth or more: no transform, one rms_power bin per hop, nothing cropped */
	if (widest_bin >= MINIMUM_RATE) {
		piece = rate = widest_bin;
		hops = hc.span / piece;
		bin_e = 0;
		crop = 0;
	}
	if (hops > MAX_TUNES)
		return rxgpu_fail(RXGPU_EINVAL, "bandwidth too wide");
	const int one_block = 2 * (1 << bin_e) * (int)ds;          /* int16 of one transform's input */
	plan->tune_count = (int)hops;
	plan->bin_e = (int)bin_e;
	plan->buf_len = one_block < DEFAULT_BUF_LENGTH ? DEFAULT_BUF_LENGTH : one_block;
	plan->downsample = (int)ds;
	plan->downsample_passes = ds_passes;
	plan->rate = (int)rate;
	plan->first_freq = lower + piece / 2;                      /* hop i is centred on lower + i * piece + piece / 2 */
	plan->bw_seen = piece;
	plan->crop = crop;
	return RXGPU_OK;
}

/* sine_table, rtl_power.c:240-254 */
int rxgpu_sine_table(int log2n, int16_t *sinewave)
{
	int n, i;
	if (log2n < 0 || log2n > 21 || !sinewave)
		return rxgpu_fail(RXGPU_EINVAL, "bad sine table size");
	n = 1 << log2n;
	for (i = 0; i < n * 3 / 4; i++) {
		double d = (double)i * 2.0 * M_PI / n;
		sinewave[i] = (int16_t)(int)round(32767 * sin(d));
	}
	return RXGPU_OK;
}

/* the window shapes of rtl_power.c:322-401 by their -w names (881-897) */
static double window_value(const char *name, int i, int length)
{
	const double N1 = (double)(length - 1);
	if (!strcmp(name, "hamming"))
		return 25.0 / 46.0 - (21.0 / 46.0) * cos(2 * i * M_PI / N1);
	if (!strcmp(name, "blackman"))
		return 7938.0 / 18608.0 - (9240.0 / 18608.0) * cos(2 * i * M_PI / N1) + (1430.0 / 18608.0) * cos(4 * i * M_PI / N1);
	if (!strcmp(name, "blackman-harris"))
		return 0.35875 - 0.48829 * cos(2 * i * M_PI / N1) + 0.14128 * cos(4 * i * M_PI / N1) - 0.01168 * cos(6 * i * M_PI / N1);
	if (!strcmp(name, "hann-poisson"))
		return 0.5 * (1 - cos(2 * M_PI * i / N1)) * pow(M_E, (-2.0 * (double)abs((int)(N1 - 1 - 2 * i))) / N1);
	if (!strcmp(name, "youssef")) {
		double w = 0.35875 - 0.48829 * cos(2 * i * M_PI / N1) + 0.14128 * cos(4 * i * M_PI / N1) - 0.01168 * cos(6 * i * M_PI / N1);
		return w * pow(M_E, (-0.0025 * (double)abs((int)(N1 - 1 - 2 * i))) / N1);
	}
	if (!strcmp(name, "bartlett")) {
		double L = (double)length, w = (i - N1 / 2) / (L / 2);
		if (w < 0)
			w = -w;
		return 1 - w;
	}
	return 1.0;   /* rectangle, kaiser (385-389) and, like the reference, anything unknown */
}

int rxgpu_window_coefs(const char *name, int length, int *coefs)
{
	if (!name || !coefs || length < 1)
		return rxgpu_fail(RXGPU_EINVAL, "bad window request");
	for (int i = 0; i < length; i++)
		coefs[i] = (int)(256 * window_value(name, i, length));     /* rtl_power.c:1036 */
	return RXGPU_OK;
}

/* ------------------------------------------------------------------ batched scan */

struct rxgpu_power_scan {
	rxgpu_power_params p;
	int max_tunes;
	int *window_dev;
	uint32_t *twiddle_dev;
	int *fir_dev;
	int16_t *work[2];
	size_t work_cap;
	uint32_t *bx_head, *bx_tail;  /* boxcar through the rx_fm decimator: per-span seam partials */
	int *regn_part;               /* -F register cascade: every wave's share of remove_dc's sums (rxk_pw_fifth_regn4) */
	size_t regn_part_cap;
	size_t bx_cap;
	long long *rms_t, *rms_p;
	size_t rms_cap;
	uint32_t *big_scratch;     /* N > 2^15: FFT blocks in HBM */
	int *big_dc;
	size_t big_cap_blocks, big_dc_cap;
	int64_t *big_partial;      /* N = 2^14, 2^15: per-group partial spectra of the two-launch transform */
	size_t big_partial_cap;
};

/* rtl_fm.c:288-300 == rtl_power.c:213-225 */
static const int cic_9_tables[11][10] = {
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

#define PW_LDS_BIN_E 15       /* 2^15 complex samples = 128 KiB of the 160 KiB LDS */
#define PW_MAX_BIN_E 21       /* frequency_range never asks for more, rtl_power.c:485 */

/* tw[0 .. n/2): the twiddles exactly as fix_fft forms them, rtl_power.c:297-301 (halve AFTER negating), packed
 * (wr, wi); tw[n/2 .. n): the same doubled, (2wr, 2wi) -- what the packed butterfly multiplies by
 * (fft_device.h); both still fit an int16 because of the halving.  n + 2 entries are allocated by the callers. */
void rxgpu_twiddle_table(const int16_t *sinewave, int n, uint32_t *tw)
{
	for (int j = 0; j < n / 2; j++) {
		int16_t wr = sinewave[j + n / 4];
		int16_t wi = (int16_t)(-sinewave[j]);
		wr >>= 1;
		wi >>= 1;
		tw[j] = ((uint32_t)(uint16_t)wr) | ((uint32_t)(uint16_t)wi << 16);
		tw[n / 2 + j] = ((uint32_t)(uint16_t)(wr * 2)) | ((uint32_t)(uint16_t)(wi * 2) << 16);
	}
	tw[n] = tw[n + 1] = 0;
}

int rxgpu_power_scan_create(rxgpu_power_scan **out, const rxgpu_power_params *p, int max_tunes,
                            const int *window_coefs, const int16_t *sinewave)
{
	int rc;
	rxgpu_power_scan *s;
	if (!out || !p || max_tunes < 1)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_scan_create: bad arguments");
	if (p->bin_e < 0 || p->bin_e > 21 || p->buf_len < 2 || (p->buf_len & 1) || p->downsample < 1)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_scan_create: bad geometry");
	if (p->bin_e > PW_MAX_BIN_E)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "FFT of 2^%d points: the reference stops at 2^%d (rtl_power.c:485)", p->bin_e, PW_MAX_BIN_E);
	if (p->bin_e > 0 && (!window_coefs || !sinewave))
		return rxgpu_fail(RXGPU_EINVAL, "window and sine tables are required for bin_e > 0");
	if (p->bin_e > 0 && p->buf_len < 2 * (1 << p->bin_e))
		return rxgpu_fail(RXGPU_EINVAL, "buf_len %d shorter than one FFT block", p->buf_len);
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	rxgpu_knobs_reload();                            /* the object keeps the kernel variants chosen now */
	s = calloc(1, sizeof(*s));
	if (!s)
		return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
	s->p = *p;
	s->max_tunes = max_tunes;
	if (p->bin_e > 0) {
		const int n = 1 << p->bin_e;
		uint32_t *tw = malloc((size_t)(n + 2) * 4);
		if (!tw) { free(s); return rxgpu_fail(RXGPU_ENOMEM, "out of host memory"); }
		rxgpu_twiddle_table(sinewave, n, tw);
		if (hipMalloc((void **)&s->window_dev, (size_t)n * 4) != hipSuccess ||
		    hipMalloc((void **)&s->twiddle_dev, (size_t)(n + 2) * 4) != hipSuccess ||
		    hipMalloc((void **)&s->fir_dev, 10 * 4) != hipSuccess ||
		    hipMemcpy(s->window_dev, window_coefs, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(s->twiddle_dev, tw, (size_t)(n + 2) * 4, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(s->fir_dev, cic_9_tables[p->downsample_passes <= 10 ? p->downsample_passes : 0], 40, hipMemcpyHostToDevice) != hipSuccess) {
			free(tw);
			rxgpu_power_scan_destroy(s);
			return rxgpu_fail(RXGPU_ENOMEM, "device table allocation failed");
		}
		free(tw);
	}
	*out = s;
	return RXGPU_OK;
}

void rxgpu_power_scan_destroy(rxgpu_power_scan *s)
{
	if (!s)
		return;
	hipFree(s->window_dev); hipFree(s->twiddle_dev); hipFree(s->fir_dev);
	hipFree(s->work[0]); hipFree(s->work[1]);
	hipFree(s->bx_head); hipFree(s->bx_tail); hipFree(s->regn_part);
	hipFree(s->big_scratch); hipFree(s->big_dc); hipFree(s->big_partial);
	hipFree(s->rms_t); hipFree(s->rms_p);
	free(s);
}

int rxgpu_power_scan_run(rxgpu_power_scan *s, const int16_t *d_in, int passes, int tunes,
                         int64_t *d_avg, int32_t *d_samples)
{
	if (!s || !d_in || !d_avg || !d_samples || passes < 1 || tunes < 1)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_scan_run: bad arguments");
	hipStream_t st = rxgpu_hip_stream();
	const rxgpu_power_params *p = &s->p;
	const size_t n_bufs = (size_t)passes * (size_t)tunes;
	const int buf_len = p->buf_len, ds = p->downsample, ds_p = p->downsample_passes;

	if (p->bin_e == 0) {                                   /* rms_power, rtl_power.c:710-713 */
		if (s->rms_cap < n_bufs) {
			hipFree(s->rms_t); hipFree(s->rms_p);
			s->rms_t = s->rms_p = NULL; s->rms_cap = 0;
			RX_HIP(hipMalloc((void **)&s->rms_t, n_bufs * 8));
			RX_HIP(hipMalloc((void **)&s->rms_p, n_bufs * 8));
			s->rms_cap = n_bufs;
		}
		rxgpu_prof_begin("pw_rms");
		RX_K(rxk_pw_rms_sums(st, d_in, n_bufs, buf_len, s->rms_t, s->rms_p));
		RX_K(rxk_pw_rms_apply(st, s->rms_t, s->rms_p, passes, tunes, buf_len, p->peak_hold, (long long *)d_avg, d_samples));
		rxgpu_prof_end("pw_rms");
		return RXGPU_OK;
	}

	const int16_t *fft_in = d_in;
	int eff_len = buf_len;
	int dc_sums_done = 0;                             /* the downsampler left remove_dc's sums in big_dc (rxk_pw_fifth_regn4) */
	size_t fft_tune_stride = (size_t)buf_len, fft_pass_stride = (size_t)tunes * (size_t)buf_len;
	if (ds > 1 && (p->boxcar || ds_p)) {
		const size_t need = n_bufs * (size_t)buf_len * 2;
		if (s->work_cap < need) {
			hipFree(s->work[0]); hipFree(s->work[1]);
			s->work[0] = s->work[1] = NULL; s->work_cap = 0;
			RX_HIP(hipMalloc((void **)&s->work[0], need));
			RX_HIP(hipMalloc((void **)&s->work[1], need));
			s->work_cap = need;
		}
		rxgpu_prof_begin("pw_downsample");
		if (p->boxcar) {                                   /* rtl_power.c:723-733 */
			/* slots the transform reads: eff_len / 2 rounded up to whole FFT blocks (rms_power never comes here) */
			const int n_fft = 1 << p->bin_e, eff = buf_len / ds;
			const int n_read = (eff + 2 * n_fft - 1) / (2 * n_fft) * n_fft;
			const unsigned long long nc = (unsigned long long)buf_len / 2, T = (unsigned long long)n_bufs * nc;
			/* (eff % (2 * n_fft): with a partial last FFT block the transform reads past the tune's compact output into the next tune's;
			 * the plain kernel below zero-fills up to n_read like the in-place original.  The reference's planner never makes that
			 * geometry, rxgpu_power_scan_create accepts it.) */
			if (ds >= 4 && ds <= RXK_DEC_MAX_DS && nc % (unsigned long long)ds == 0 && T % 4 == 0 && eff % (2 * n_fft) == 0) {
				/* whole windows per buffer: the sums over the concatenated buffers are low_pass (rtl_fm.c:351-371) on an already
				 * scaled, unrotated stream -- the rx_fm decimator, then its per-span seam entries; output compact, eff_len per buffer */
				const size_t n_spans = (size_t)((T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN) + 1;
				if (s->bx_cap < n_spans) {
					hipFree(s->bx_head); hipFree(s->bx_tail);
					s->bx_head = s->bx_tail = NULL; s->bx_cap = 0;
					RX_HIP(hipMalloc((void **)&s->bx_head, n_spans * 4 + 16 + n_spans * 32));     /* + four {I, Q} wave sums per span (rxk_pw_boxcar_sums) */
					RX_HIP(hipMalloc((void **)&s->bx_tail, n_spans * 4));
					s->bx_cap = n_spans;
				}
#ifdef RXGPU_NO_BOX_SUMS                                   /* (scratch build for the A/B) */
				if (0) {
#else
				if (nc % RXK_DEC_SPAN == 0 && p->bin_e >= 14 && p->bin_e <= 21) {
#endif
					/* buffers of whole spans in front of the large-N transform: remove_dc's sums ride in the decimator (every output a span
					 * stores) and in the seam kernel (the one it leaves), where the transform's own dc pass would have put them */
					if (s->big_dc_cap < n_bufs) {
						hipFree(s->big_dc);
						s->big_dc = NULL; s->big_dc_cap = 0;
						RX_HIP(hipMalloc((void **)&s->big_dc, n_bufs * 24 + 64));
						s->big_dc_cap = n_bufs;
					}
					long long *sums = rxk_pw_dc_sums(s->big_dc, n_bufs);
					RX_HIP(hipMemsetAsync(sums, 0, n_bufs * 16, st));
					int *wave_sums = (int *)(s->bx_head + ((s->bx_cap + 1) & ~(size_t)1));   /* behind the head entries, 8-byte aligned */
					RX_K(rxk_pw_boxcar_sums(st, d_in, T, ds, (uint32_t *)s->work[0], s->bx_head, s->bx_tail, wave_sums));
					RX_K(rxk_pw_boxcar_seams(st, (uint32_t *)s->work[0], s->bx_head, s->bx_tail, T, ds, sums, wave_sums, (unsigned)(nc / RXK_DEC_SPAN)));
					dc_sums_done = 1;
				} else {
					RX_K(rxk_fm_decimate(st, d_in, T, ds, 0, 1, 0, (uint32_t *)s->work[0], s->bx_head, s->bx_tail, 0, NULL, 0));
					RX_K(rxk_pw_boxcar_seams(st, (uint32_t *)s->work[0], s->bx_head, s->bx_tail, T, ds, NULL, NULL, 1));
				}
				fft_tune_stride = (size_t)eff;
				fft_pass_stride = (size_t)tunes * (size_t)eff;
			} else {
				RX_K(rxk_pw_boxcar(st, d_in, s->work[0], n_bufs, buf_len, ds, n_read));
			}
			fft_in = s->work[0];
		} else if (ds_p >= 2 && ds_p <= 4 && (buf_len / 2) % (4 << ds_p) == 0 && buf_len / 2 >= (64 << ds_p) && ((size_t)d_in & 15u) == 0 && (buf_len / 2) % 4 == 0 &&
		           (p->comp_fir_size == 9 || p->comp_fir_size == 0) && n_bufs < ((size_t)1 << 31)) {
			/* four passes (ds = 16): the register cascade -- passes, droop FIR and remove_dc's sums in one launch, the 1/16-rate buffers
			 * written once (rtl_power.c:734-745).  The sums go where the large-N transform's dc pass would have put them. */
			const int fir = p->comp_fir_size == 9;
			long long *sums = NULL;
			const int eff = buf_len / ds;
			if (p->bin_e >= 14 && p->bin_e <= 21 && eff % (2 << p->bin_e) == 0) {
				if (s->big_dc_cap < (size_t)passes * (size_t)tunes) {
					hipFree(s->big_dc);
					s->big_dc = NULL; s->big_dc_cap = 0;
					RX_HIP(hipMalloc((void **)&s->big_dc, (size_t)passes * (size_t)tunes * 24 + 64));
					s->big_dc_cap = (size_t)passes * (size_t)tunes;
				}
				sums = rxk_pw_dc_sums(s->big_dc, (size_t)passes * (size_t)tunes);
				const size_t parts = (size_t)rxk_pw_fifth_regn_parts(n_bufs, (unsigned)(buf_len / 2), ds_p);
				if (s->regn_part_cap < parts) {
					hipFree(s->regn_part);
					s->regn_part = NULL; s->regn_part_cap = 0;
					RX_HIP(hipMalloc((void **)&s->regn_part, parts * 8));
					s->regn_part_cap = parts;
				}
				dc_sums_done = 1;
			}
			RX_K(rxk_pw_fifth_regn(st, d_in, n_bufs, (unsigned)(buf_len / 2), (unsigned)(buf_len / 2), ds_p, fir ? s->fir_dev : NULL, fir ? cic_9_tables[ds_p] : NULL,
			                        s->work[0], (unsigned)(buf_len / 2), sums, s->regn_part));
			fft_in = s->work[0];
		} else {                                           /* rtl_power.c:734-743 */
			const int16_t *src = d_in;
			int n_in = buf_len / 2, which = 0, j0 = 0;
			if (ds_p > 4 && (buf_len / 2) % 64 == 0 && buf_len / 2 >= 1024 && ((size_t)d_in & 15u) == 0 && n_bufs < ((size_t)1 << 31)) {
				/* more than four passes: the first four -- 15/16 of the cascade's samples -- in the register kernel (no FIR, no sums there: those come
				 * behind the LAST pass), the rest on the 1/16-rate buffers below.  Every pass is stateless and eased in (rtl_power.c:582-607), so a
				 * later pass only needs the level-4 buffers right from their first sample, which k_pw_fifth_fix sees to */
				RX_K(rxk_pw_fifth_regn(st, d_in, n_bufs, (unsigned)(buf_len / 2), (unsigned)(buf_len / 2), 4, NULL, NULL, s->work[0], (unsigned)(buf_len / 2), NULL, NULL));
				src = s->work[0];
				n_in >>= 4;
				which = 1;
				j0 = 4;
			}
			for (int j = j0; j < ds_p;) {
				/* up to three passes per launch while the pass input is whole tiles; otherwise one pass at a time */
				int fuse = ds_p - j < 3 ? ds_p - j : 3;
				while (fuse > 0 && n_in % RXK_FIFTH_TILE)
					fuse = 0;
				if (fuse) {
					RX_K(rxk_pw_fifth_fused(st, src, n_bufs, (unsigned)n_in, (unsigned)(buf_len / 2), fuse, s->work[which], (unsigned)(buf_len / 2)));
					n_in >>= fuse;
					j += fuse;
				} else {
					RX_K(rxk_pw_fifth(st, src, s->work[which], n_bufs, n_in, buf_len / 2, buf_len / 2));
					n_in = (n_in + 1) / 2;
					j++;
				}
				src = s->work[which];
				which ^= 1;
			}
			if (p->comp_fir_size == 9 && ds_p <= 10) {
				RX_K(rxk_pw_droop(st, src, s->work[which], n_bufs, (buf_len >> ds_p) / 2, buf_len / 2, s->fir_dev));
				src = s->work[which];
			}
			fft_in = src;
		}
		rxgpu_prof_end("pw_downsample");
		eff_len = buf_len / ds;
	}
	/* enough workgroups to fill 256 CUs several times over, few enough that the per-group
	 * int64 accumulators amortise the global atomics */
	int groups = (4096 + tunes - 1) / tunes;
	/* few tunes (a narrow sweep with fine bins): thousands of passes land on the same N bins.  Fewer, longer groups, and their
	 * spectra go to a partial buffer that one reduction folds into avg[] instead of int64 atomics from every group */
	const int few = p->bin_e >= 5 && p->bin_e <= 13 && eff_len % (2 << p->bin_e) == 0 && tunes <= 64;
	if (few)
		groups = (1024 + tunes - 1) / tunes;
	if (groups > passes) groups = passes;
	if (groups < 1) groups = 1;
	const int ppg = (passes + groups - 1) / groups;
	const int n_blocks = (eff_len + 2 * (1 << p->bin_e) - 1) / (2 * (1 << p->bin_e));
	if (few) {
		const size_t fpw = p->bin_e >= 12 ? 1 : (size_t)4096 >> p->bin_e;
		const size_t need = (size_t)((passes + ppg - 1) / ppg) * (size_t)tunes * fpw << p->bin_e;
		if (s->big_partial_cap < need && need * 8 <= ((size_t)1 << 29)) {
			hipFree(s->big_partial);
			s->big_partial = NULL; s->big_partial_cap = 0;
			if (hipMalloc((void **)&s->big_partial, need * 8) == hipSuccess)
				s->big_partial_cap = need;
		}
	}
	int samples_done = 0;
	rxgpu_prof_begin("pw_fft");
	/* N = 2^14 .. 2^21 (the reference's limit) with whole blocks: the register-blocked transform in two to four launches over a scratch
	 * copy (rxk_pw_fft_mid: radix-16 passes through HBM until a sub-transform fits a workgroup); a buffer that is no whole number of
	 * transforms (the reference's planner never makes one; rxgpu_power_scan_create accepts it) takes the one-launch-per-radix-2-stage
	 * network for N > 2^15 and the LDS radix-2 kernel below that */
	const int mid = p->bin_e >= 14 && p->bin_e <= 21 && eff_len % (2 << p->bin_e) == 0;
	if (p->bin_e > PW_LDS_BIN_E || mid) {
		const size_t total = (size_t)passes * (size_t)tunes * (size_t)n_blocks, n = (size_t)1 << p->bin_e;
		size_t want = ((size_t)1 << 28) / n;                /* up to 1 GiB of scratch */
		if (want > total) want = total;
		if (mid && want < (size_t)tunes * (size_t)n_blocks)
			want = (size_t)tunes * (size_t)n_blocks;       /* a launch of the two-kernel path covers whole passes */
		if (want < 1) want = 1;
		if (s->big_cap_blocks < want) {
			hipFree(s->big_scratch);
			s->big_scratch = NULL; s->big_cap_blocks = 0;
			RX_HIP(hipMalloc((void **)&s->big_scratch, want * n * 4));
			s->big_cap_blocks = want;
		}
		if (s->big_dc_cap < (size_t)passes * (size_t)tunes) {
			hipFree(s->big_dc);
			s->big_dc = NULL; s->big_dc_cap = 0;
			/* 2 ints per (pass, tune), then the int64 sums the reduction accumulates (power_kernels.hip pwb_dc) */
			RX_HIP(hipMalloc((void **)&s->big_dc, (size_t)passes * (size_t)tunes * 24 + 64));
			s->big_dc_cap = (size_t)passes * (size_t)tunes;
		}
		if (mid) {
			/* per-group partial spectra instead of atomics on avg[] (one tune: every pass lands on the same N bins) */
			const size_t need = ((size_t)RXK_PWM_TARGET_WG / (n / 4096) + (size_t)tunes * (size_t)n_blocks) * n;
			if (s->big_partial_cap < need && need * 8 <= ((size_t)1 << 30)) {
				hipFree(s->big_partial);
				s->big_partial = NULL; s->big_partial_cap = 0;
				if (hipMalloc((void **)&s->big_partial, need * 8) == hipSuccess)
					s->big_partial_cap = need;
			}
			RX_K(rxk_pw_fft_mid(st, fft_in, fft_tune_stride, fft_pass_stride, passes, tunes, p->bin_e, eff_len,
			                    s->window_dev, s->twiddle_dev, p->peak_hold, s->big_scratch, s->big_cap_blocks, s->big_dc, (long long *)d_avg,
			                    (long long *)s->big_partial, s->big_partial_cap, dc_sums_done, d_samples, n_blocks * ds * passes));   /* + rtl_power.c:769 */
			samples_done = 1;
		}
		else
			RX_K(rxk_pw_fft_big(st, fft_in, fft_tune_stride, fft_pass_stride, passes, tunes, p->bin_e, eff_len,
			                    s->window_dev, s->twiddle_dev, p->peak_hold, s->big_scratch, s->big_cap_blocks, s->big_dc, (long long *)d_avg));
	} else {
		RX_K(rxk_pw_fft(st, fft_in, fft_tune_stride, fft_pass_stride, passes, tunes, p->bin_e, eff_len, eff_len,
		                s->window_dev, s->twiddle_dev, p->peak_hold, ppg, (long long *)d_avg,
		                few ? (long long *)s->big_partial : NULL, few ? s->big_partial_cap : 0));
	}
	rxgpu_prof_end("pw_fft");
	if (!samples_done)
		RX_K(rxk_pw_samples(st, d_samples, tunes, n_blocks * ds * passes));   /* rtl_power.c:769 */
	return RXGPU_OK;
}

/* ------------------------------------------------------------------ drop-ins */

/* scanner() is called once per sweep for the life of the process with the same geometry and tables (rtl_power.c:1040-1046):
 * the scan object, its device tables and the device buffers are kept between calls and only rebuilt when the geometry or a
 * table changes.
 *
 * The integrators stay on the device between calls.  Nothing reads ts->avg[] / ts->samples between two scanner() calls -- the
 * reference only looks at them in csv_dbm, once per report interval (rtl_power.c:1045-1050) -- so rxgpu_scan uploads the tunes'
 * buf16 (9.8 MB at the configs[2] geometry), adds the sweep into device-resident accumulators that start from zero, and
 * returns without waiting; rxgpu_scan_sync (called by rxgpu_csv_dbm for a tuning_state of the sweep, or explicitly in front of the
 * reference's own csv_dbm) brings the accumulated sums back ONCE per interval and merges them into the caller's arrays:
 * avg += delta, or MAX for peak hold (every term is a non-negative power, so a running maximum that starts from zero
 * commutes with the caller's), samples += delta.  The first version moved all of avg[] both ways on every call: 2 x 19.6 MB
 * for 9.8 MB of input.
 *
 * Deferred accumulation is OPT-IN (rxgpu_scan_deferred(1), or $RXGPU_SCAN_DEFERRED=1 read at the first rxgpu_scan): by default every
 * rxgpu_scan ends with the merge, so the structs are current when it returns ("exactly as the CPU") and the library keeps no pointer of
 * the caller's past the call.  In deferred mode the library keeps `tunes` between calls, and the rule is: it only ever writes through a
 * pointer the caller has handed to the call that is running -- rxgpu_scan_sync(tunes, n), rxgpu_csv_dbm(&tunes[i]) -- so a pending
 * interval that meets another array, count or geometry FAILS rxgpu_scan (RXGPU_EINVAL: sync first), and one that is still pending at
 * rxgpu_shutdown / rxgpu_power_dropin_release is dropped with a line on stderr: the old array may be gone by then. */
/* A table of caller buffers page-locked in place -- one per tune, malloc'd once by frequency_range and never freed (rtl_power.c:518-531) -- with
 * their device-visible addresses in an array a kernel reads. */
struct zc_table {
	const void **host;               /* [cap] the buffer each entry was resolved for */
	void **dev;                      /* [cap] its device-visible address (host copy of the table) */
	unsigned char *owned;            /* [cap] page-locked HERE (released with the cache), not by the caller's rxgpu_pin */
	unsigned char *zero;             /* [cap] (avg rows) the row holds zeros: rxgpu_csv_dbm cleared it after the last merge and nothing else writes it */
	int zero_hint;                   /* where rxgpu_csv_dbm's next row is expected */
	void **d_tab;                    /* the table on the device */
	int count;                       /* entries resolved; 0 = none / not usable */
	int failed;                      /* a buffer could not be page-locked: the copying path from then on (until the geometry changes) */
	unsigned gen;
};

/* Where rxgpu_scan / rxgpu_scan_sync spend their time (host clock, $RXGPU_DROPIN_TIMING=1; rxgpu_scan_timing reads and clears):
 *   0 scan: geometry check + the table of page-locked rows     1 scan: gather launch (or staging memcpy + H2D enqueue)
 *   2 scan: enqueue of the transforms                           3 scan: wait until the gather has read the caller's buffers
 *   4 sync: D2H of the accumulators (wait)                      5 sync: merge into the caller's avg[] / samples
 *   6 scans, 7 syncs */
static double g_st[8];
static int g_st_on = -1;
static double st_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
static int st_on(void)
{
	if (g_st_on < 0) {
		const char *e = rxgpu_knob("RXGPU_DROPIN_TIMING");
		g_st_on = e && atoi(e) > 0;
	}
	return g_st_on;
}
int rxgpu_scan_timing(double *us, int n)
{
	g_st_on = -1;                                        /* the switch is looked at again (rxgpu_knobs_reload in between) */
	for (int i = 0; i < n && i < 8; i++) {
		us[i] = g_st[i];
		g_st[i] = 0;
	}
	return n < 8 ? n : 8;
}

static struct {
	rxgpu_power_scan *s;
	rxgpu_power_params p;
	int tune_cap;
	int *window_copy;
	int16_t *sine_copy;
	int16_t *d_in[2];                /* two sweeps of input in rotation: sweep k+1 is staged while sweep k is transformed */
	int16_t *h_in[2];                /* pinned staging: tunes[i].buf16 are separate mallocs of the caller */
	hipEvent_t ev_in[2];
	int ev_valid[2];
	unsigned long long calls;
	int64_t *d_avg;                  /* accumulated since the last sync, from zero */
	int32_t *d_samples;
	int64_t *h_avg;                  /* pinned: download of the accumulators (+ samples behind them) */
	/* zero-copy input: the caller's tunes[i].buf16 are malloc'd once and never freed (rtl_power.c:518-531) -- page-locked in place the first
	 * time they are seen, their device-visible addresses in a table the gather kernel reads (k_pw_gather_rows) */
	struct zc_table zin;             /* the tunes' buf16: the gather's rows */
	struct zc_table zavg;            /* the tunes' avg[]: rxgpu_scan_sync's merge runs on them in place (round 6) */
	int zc_last;                     /* the last rxgpu_scan read its input zero-copy (rxgpu_scan_zero_copy) */
	int zc_sync_last;                /* the last merge ran on the caller's avg[] in place */
	hipEvent_t ev_gather;
	hipEvent_t ev_fft;               /* the transforms of the latest sweep are enqueued behind this */
	struct tuning_state *tunes;      /* whose sums the accumulators hold */
	int tune_count;
	int dirty;
	int deferred;                    /* -1: not decided yet ($RXGPU_SCAN_DEFERRED at the first scan), 0: every scan merges, 1: rxgpu_scan_sync does */
	long syncs;
} g_scan = { .deferred = -1 };
static pthread_mutex_t g_scan_lock = PTHREAD_MUTEX_INITIALIZER;

static void zc_release(struct zc_table *z)
{
	if (z->owned && z->host)
		for (int i = 0; i < z->count; i++)
			if (z->owned[i]) {
				rxgpu_pin_changed();
				(void)hipHostUnregister((void *)z->host[i]);
				z->owned[i] = 0;
			}
	(void)hipGetLastError();
	z->count = 0;
}

static void zc_free(struct zc_table *z)
{
	zc_release(z);
	free(z->host); free(z->dev); free(z->owned); free(z->zero);
	hipFree(z->d_tab);
	memset(z, 0, sizeof(*z));
}

static int zc_alloc(struct zc_table *z, int cap)
{
	z->host = calloc((size_t)cap, sizeof(*z->host));
	z->dev = calloc((size_t)cap, sizeof(*z->dev));
	z->owned = calloc((size_t)cap, 1);
	z->zero = calloc((size_t)cap, 1);
	return z->host && z->dev && z->owned && z->zero && hipMalloc((void **)&z->d_tab, (size_t)cap * sizeof(void *)) == hipSuccess;
}

static const void *tune_buf16(const struct tuning_state *t) { return t->buf16; }
static const void *tune_avg(const struct tuning_state *t) { return t->avg; }

/* Device-visible addresses of every tune's buffer (`field`: buf16 or avg), page-locking those that are not yet, table uploaded.  Returns 1 with
 * *row0 = the table row of tunes[0], or 0 = this call takes the copying path.
 * LIFETIME (include/rxgpu.h, rxgpu_scan): a buffer that has been handed to rxgpu_scan stays allocated until rxgpu_scan_release() -- it is
 * page-locked in place, and an allocation that dies under its registration leaves pinned pages behind which a later allocation at the same
 * address would silently alias.  The reference never frees them (rtl_power.c:518-531); ctypes callers release before their arrays die.
 * A call on a SUB-ARRAY of the registered sweep (the drop-in's missed-read path: rxgpu_scan(&tunes[i], j - i)) is looked up in the table and
 * goes through the rows it already has; a shorter call with buffers the table does not know is copied -- neither touches the
 * registrations of the full sweep. */
static int zc_resolve(struct zc_table *z, const void *(*field)(const struct tuning_state *), struct tuning_state *tunes, int tune_count, size_t row_bytes,
                      hipStream_t st, int *row0)
{
	const char *e = rxgpu_knob("RXGPU_SCAN_ZC");
	*row0 = 0;
	if ((e && e[0] == '0') || z->failed || (row_bytes & 15u) || !z->host)
		return 0;
	const unsigned gen = rxgpu_pin_generation();
	if (z->count >= tune_count && z->gen == gen) {
		int k0 = 0;
		while (k0 + tune_count <= z->count && z->host[k0] != field(&tunes[0]))
			k0++;
		int same = k0 + tune_count <= z->count;
		for (int i = 0; same && i < tune_count; i++)
			same = z->host[k0 + i] == field(&tunes[i]);
		if (same) {
			*row0 = k0;
			return 1;
		}
		if (tune_count < z->count)
			return 0;                                    /* part of a sweep, through buffers of its own: copied; the sweep's table stays */
	}
	zc_release(z);
	memset(z->zero, 0, (size_t)tune_count);
	for (int i = 0; i < tune_count; i++) {
		void *a = NULL, *a_end = NULL;
		void *b = (void *)field(&tunes[i]);
		int owned = 0;
		if (hipHostGetDevicePointer(&a, b, 0) != hipSuccess || hipHostGetDevicePointer(&a_end, (char *)b + row_bytes - 1, 0) != hipSuccess) {
			(void)hipGetLastError();
			a = NULL;
			/* exactly the bytes that are used, not the allocation (buf16: buf_len * 4, rtl_power.c:526) and not rounded out to pages */
			if (hipHostRegister(b, row_bytes, hipHostRegisterDefault) == hipSuccess) {
				owned = 1;
				if (hipHostGetDevicePointer(&a, b, 0) != hipSuccess)
					a = NULL;
			}
		}
		z->host[i] = b;
		z->dev[i] = a;
		z->owned[i] = (unsigned char)owned;
		z->count = i + 1;
		if (!a || ((size_t)a & 15u)) {
			(void)hipGetLastError();
			zc_release(z);
			z->failed = 1;
			return 0;
		}
	}
	rxgpu_pin_changed();
	/* both tables pin: each one's generation is what IT left behind (the other table's registrations are this library's own and do not move) */
	z->gen = rxgpu_pin_generation();
	if (z == &g_scan.zin && g_scan.zavg.count) g_scan.zavg.gen = z->gen;
	if (z == &g_scan.zavg && g_scan.zin.count) g_scan.zin.gen = z->gen;
	if (hipMemcpyAsync(z->d_tab, z->dev, (size_t)tune_count * sizeof(void *), hipMemcpyHostToDevice, st) != hipSuccess ||
	    hipStreamSynchronize(st) != hipSuccess) {            /* dev[] is pageable: the copy has read it when this returns */
		(void)hipGetLastError();
		zc_release(z);
		z->failed = 1;
		return 0;
	}
	return 1;
}

static void scan_cache_drop(void)
{
	zc_free(&g_scan.zin);
	zc_free(&g_scan.zavg);
	if (g_scan.ev_gather) hipEventDestroy(g_scan.ev_gather);
	if (g_scan.ev_fft) hipEventDestroy(g_scan.ev_fft);
	rxgpu_power_scan_destroy(g_scan.s);
	free(g_scan.window_copy); free(g_scan.sine_copy);
	for (int k = 0; k < 2; k++) {
		hipFree(g_scan.d_in[k]);
		if (g_scan.h_in[k]) hipHostFree(g_scan.h_in[k]);
		if (g_scan.ev_in[k]) hipEventDestroy(g_scan.ev_in[k]);
	}
	hipFree(g_scan.d_avg); hipFree(g_scan.d_samples);
	if (g_scan.h_avg) hipHostFree(g_scan.h_avg);
	const int deferred = g_scan.deferred;
	const long syncs = g_scan.syncs;
	memset(&g_scan, 0, sizeof(g_scan));
	g_scan.deferred = deferred;
	g_scan.syncs = syncs;
}

/* accumulators -> `tunes`, the array the running call was handed (the one the pending sums belong to: the callers check);
 * the device side starts from zero again */
static int scan_sync_locked(struct tuning_state *tunes)
{
	if (!g_scan.dirty)
		return RXGPU_OK;
	hipStream_t st = rxgpu_hip_stream();
	const int tc = g_scan.tune_count;
	const size_t n = (size_t)1 << g_scan.p.bin_e;
	int32_t *h_samples = (int32_t *)(g_scan.h_avg + (size_t)tc * n);
	const int timing = st_on();
	double t_a = timing ? st_now() : 0;
	/* The merge IN PLACE (round 6): the tunes' avg[] -- one malloc each, never freed, like buf16 -- are page-locked once and ONE launch adds (or
	 * maxes) the device's accumulators into them across PCIe, reading and writing the host rows and zeroing the accumulators as it goes: the link
	 * carries 19.6 MB each way at once and the host adds nothing but the 599 sample counts.  Before: D2H of the accumulators, then 2.45 M int64
	 * additions on this thread -- 1.2-1.5 ms per interval at the configs[2] geometry, two thirds of it the additions. */
	int row0 = 0;
	const int zc = zc_resolve(&g_scan.zavg, tune_avg, tunes, tc, n * 8, st, &row0);
	g_scan.zc_sync_last = zc;
	if (zc) {
		/* rows this library's own csv_dbm zeroed after the last merge (rtl_power.c:815-817 does the same) and that nothing has written since
		 * -- the reference touches avg[] in scanner() and csv_dbm only -- need not be READ across the link: sum and maximum with zero are the
		 * accumulator itself, the merge is then one write stream (19.6 MB one way, not both) */
		int all_zero = 1;
		for (int i = 0; i < tc; i++) {
			all_zero &= g_scan.zavg.zero[row0 + i];
			g_scan.zavg.zero[row0 + i] = 0;
		}
		rxgpu_prof_begin("pw_zc_merge");
		if (rxk_pw_merge_rows(st, (void *const *)g_scan.zavg.d_tab + row0, tc, n * 8, (long long *)g_scan.d_avg, g_scan.p.peak_hold, all_zero) != 0)
			return rxgpu_fail(RXGPU_ENODEV, "rxgpu_scan_sync: merge launch failed: %s", hipGetErrorString(hipGetLastError()));
		rxgpu_prof_end("pw_zc_merge");
	} else {
		if (g_scan.zavg.zero && g_scan.zavg.count)
			memset(g_scan.zavg.zero, 0, (size_t)g_scan.zavg.count);      /* the host adds below: no row stays known-zero */
		RX_HIP(hipMemcpyAsync(g_scan.h_avg, g_scan.d_avg, (size_t)tc * n * 8, hipMemcpyDeviceToHost, st));
		RX_HIP(hipMemsetAsync(g_scan.d_avg, 0, (size_t)tc * n * 8, st));
	}
	RX_HIP(hipMemcpyAsync(h_samples, g_scan.d_samples, (size_t)tc * 4, hipMemcpyDeviceToHost, st));
	RX_HIP(hipMemsetAsync(g_scan.d_samples, 0, (size_t)tc * 4, st));
	RX_HIP(hipStreamSynchronize(st));
	if (timing) { const double t_b = st_now(); g_st[4] += t_b - t_a; t_a = t_b; }
	for (int i = 0; i < tc; i++) {
		if (!zc) {
			int64_t *avg = tunes[i].avg;
			const int64_t *delta = g_scan.h_avg + (size_t)i * n;
			if (g_scan.p.peak_hold) {
				for (size_t j = 0; j < n; j++)
					if (delta[j] > avg[j])
						avg[j] = delta[j];
			} else {
				for (size_t j = 0; j < n; j++)
					avg[j] += delta[j];
			}
		}
		tunes[i].samples += h_samples[i];
	}
	if (timing) { g_st[5] += st_now() - t_a; g_st[7] += 1; }
	g_scan.dirty = 0;
	g_scan.tunes = NULL;                             /* nothing pending: no pointer of the caller's is kept */
	g_scan.syncs++;
	rxgpu_prof_collect();
	return RXGPU_OK;
}

void rxgpu_power_dropin_release(void)
{
	pthread_mutex_lock(&g_scan_lock);
	if (g_scan.dirty)                                    /* deferred mode only; the array it belongs to may be gone: never written here */
		fprintf(stderr, "librxgpu: dropping the sums of a sweep interval over %d tunes that was never merged "
		        "(rxgpu_scan_sync before rxgpu_shutdown)\n", g_scan.tune_count);
	scan_cache_drop();
	pthread_mutex_unlock(&g_scan_lock);
}

void rxgpu_scan_release(void) { rxgpu_power_dropin_release(); }

static int scan_locked(struct tuning_state *tunes, int tune_count, const int *window_coefs,
                       const int16_t *sinewave, int boxcar, int comp_fir_size, int peak_hold)
{
	int rc;
	rxgpu_power_params p;
	/* scanner() uses tunes[0]'s geometry for every tune, rtl_power.c:676-678 */
	memset(&p, 0, sizeof(p));
	p.bin_e = tunes[0].bin_e;
	p.buf_len = tunes[0].buf_len;
	p.downsample = tunes[0].downsample;
	p.downsample_passes = tunes[0].downsample_passes;
	p.boxcar = boxcar;
	p.comp_fir_size = comp_fir_size;
	p.peak_hold = peak_hold;
	const size_t n = (size_t)1 << p.bin_e;
	const size_t n_sine = n * 3 / 4 ? n * 3 / 4 : 1;
	int same = g_scan.s && !memcmp(&p, &g_scan.p, sizeof(p)) && tune_count <= g_scan.tune_cap;
	if (same && p.bin_e > 0)
		same = window_coefs && sinewave && !memcmp(g_scan.window_copy, window_coefs, n * sizeof(int)) &&
		       !memcmp(g_scan.sine_copy, sinewave, n_sine * sizeof(int16_t));
	if (g_scan.deferred < 0) {
		const char *e = rxgpu_knob("RXGPU_SCAN_DEFERRED");
		g_scan.deferred = e && atoi(e) > 0;
	}
	/* a pending interval (deferred mode) and another sweep geometry, tuning_state array or count: the sums belong to an array this
	 * call was not handed -- it may have been freed -- so they are not written anywhere; the caller has to merge them first */
	if (g_scan.dirty && (!same || tunes != g_scan.tunes || tune_count != g_scan.tune_count))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_scan: %d tunes of another sweep are still accumulated on the device "
		                  "(deferred mode): call rxgpu_scan_sync on that array first", g_scan.tune_count);
	if (!same) {
		scan_cache_drop();
		if ((rc = rxgpu_power_scan_create(&g_scan.s, &p, tune_count, window_coefs, sinewave)) != RXGPU_OK)
			return rc;
		g_scan.p = p;
		g_scan.tune_cap = tune_count;
		if (p.bin_e > 0) {
			g_scan.window_copy = malloc(n * sizeof(int));
			g_scan.sine_copy = malloc(n_sine * sizeof(int16_t));
			if (!g_scan.window_copy || !g_scan.sine_copy) {
				scan_cache_drop();
				return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
			}
			memcpy(g_scan.window_copy, window_coefs, n * sizeof(int));
			memcpy(g_scan.sine_copy, sinewave, n_sine * sizeof(int16_t));
		}
		const size_t in_bytes = (size_t)tune_count * p.buf_len * 2;
		if (hipMalloc((void **)&g_scan.d_in[0], in_bytes) != hipSuccess || hipMalloc((void **)&g_scan.d_in[1], in_bytes) != hipSuccess ||
		    hipMalloc((void **)&g_scan.d_avg, (size_t)tune_count * n * 8) != hipSuccess ||
		    hipMalloc((void **)&g_scan.d_samples, (size_t)tune_count * 4 + 4) != hipSuccess ||
		    hipHostMalloc((void **)&g_scan.h_avg, (size_t)tune_count * n * 8 + (size_t)tune_count * 4, 0) != hipSuccess ||
		    hipHostMalloc((void **)&g_scan.h_in[0], in_bytes, 0) != hipSuccess || hipHostMalloc((void **)&g_scan.h_in[1], in_bytes, 0) != hipSuccess ||
		    hipEventCreateWithFlags(&g_scan.ev_in[0], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&g_scan.ev_in[1], hipEventDisableTiming) != hipSuccess ||
		    hipMemset(g_scan.d_avg, 0, (size_t)tune_count * n * 8) != hipSuccess ||
		    hipMemset(g_scan.d_samples, 0, (size_t)tune_count * 4 + 4) != hipSuccess ||
		    hipEventCreateWithFlags(&g_scan.ev_fft, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&g_scan.ev_gather, hipEventDisableTiming) != hipSuccess) {
			scan_cache_drop();
			return rxgpu_fail(RXGPU_ENOMEM, "rxgpu_scan: buffer allocation failed");
		}
		if (!zc_alloc(&g_scan.zin, tune_count) || !zc_alloc(&g_scan.zavg, tune_count)) {
			scan_cache_drop();
			return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
		}
	}
	g_scan.tunes = tunes;
	g_scan.tune_count = tune_count;
	hipStream_t st = rxgpu_hip_stream();
	const int timing = st_on();
	double t_a = timing ? st_now() : 0, t_b;
	const int k = (int)(g_scan.calls++ & 1);
	int row0 = 0;
	const int zc = zc_resolve(&g_scan.zin, tune_buf16, tunes, tune_count, (size_t)p.buf_len * 2, st, &row0);
	if (timing) { t_b = st_now(); g_st[0] += t_b - t_a; t_a = t_b; }
	if (zc) {
		/* one launch reads every tune's page-locked buf16 across PCIe into the scan's input; the caller refills buf16 as soon as this
		 * call returns (the next sweep's readStream, rtl_power.c:693-704), so the call waits for the gather -- not for the scan.
		 * On the COPY stream (round 6): behind the transforms of the sweep before on the compute stream it started 64 us late; the input
		 * buffers rotate, buffer k was last read by the transforms of two sweeps ago (ev_in) */
		hipStream_t sc = rxgpu_hip_stream3();
		if (g_scan.ev_valid[k])
			RX_HIP(hipStreamWaitEvent(sc, g_scan.ev_in[k], 0));
		rxgpu_prof_begin_on("pw_zc_gather", sc);
		if (rxk_pw_gather_rows(sc, (const void *const *)g_scan.zin.d_tab + row0, tune_count, (size_t)p.buf_len * 2, g_scan.d_in[k]) != 0)
			return rxgpu_fail(RXGPU_ENODEV, "rxgpu_scan: gather launch failed: %s", hipGetErrorString(hipGetLastError()));
		rxgpu_prof_end_on("pw_zc_gather", sc);
		RX_HIP(hipEventRecord(g_scan.ev_gather, sc));
		RX_HIP(hipStreamWaitEvent(st, g_scan.ev_gather, 0));
	} else {
		/* the caller's scattered buffers into pinned staging (one copy instead of one per tune); the staging of two sweeps ago has long been read */
		if (g_scan.ev_valid[k])
			RX_HIP(hipEventSynchronize(g_scan.ev_in[k]));
		for (int i = 0; i < tune_count; i++)
			memcpy(g_scan.h_in[k] + (size_t)i * p.buf_len, tunes[i].buf16, (size_t)p.buf_len * 2);
		RX_HIP(hipMemcpyAsync(g_scan.d_in[k], g_scan.h_in[k], (size_t)tune_count * p.buf_len * 2, hipMemcpyHostToDevice, st));
	}
	if (timing) { t_b = st_now(); g_st[1] += t_b - t_a; t_a = t_b; }
	if ((rc = rxgpu_power_scan_run(g_scan.s, g_scan.d_in[k], 1, tune_count, g_scan.d_avg, g_scan.d_samples)) != RXGPU_OK)
		return rc;
	if (timing) { t_b = st_now(); g_st[2] += t_b - t_a; t_a = t_b; }
	RX_HIP(hipEventRecord(g_scan.ev_in[k], st));
	g_scan.ev_valid[k] = 1;
	g_scan.dirty = 1;
	g_scan.zc_last = zc;
	if (zc && g_scan.deferred)
		RX_HIP(hipEventSynchronize(g_scan.ev_gather));
	if (timing) { g_st[3] += st_now() - t_a; g_st[6] += 1; }
	if (!g_scan.deferred)
		return scan_sync_locked(tunes);
	return RXGPU_OK;
}

int rxgpu_scan(struct tuning_state *tunes, int tune_count, const int *window_coefs,
               const int16_t *sinewave, int boxcar, int comp_fir_size, int peak_hold)
{
	int rc;
	if (!tunes || tune_count < 1)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_scan: no tunes");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	pthread_mutex_lock(&g_scan_lock);
	rc = scan_locked(tunes, tune_count, window_coefs, sinewave, boxcar, comp_fir_size, peak_hold);
	pthread_mutex_unlock(&g_scan_lock);
	return rc;
}

int rxgpu_scan_sync(struct tuning_state *tunes, int tune_count)
{
	int rc = RXGPU_OK;
	pthread_mutex_lock(&g_scan_lock);
	if (g_scan.dirty) {
		if (!tunes)
			rc = rxgpu_fail(RXGPU_EINVAL, "rxgpu_scan_sync: pass the tuning_state array of the pending sweep (%d tunes)", g_scan.tune_count);
		else if (tunes != g_scan.tunes || tune_count != g_scan.tune_count)
			rc = rxgpu_fail(RXGPU_EINVAL, "rxgpu_scan_sync: the accumulated sweep belongs to another tuning_state array (%d tunes)", g_scan.tune_count);
		else
			rc = scan_sync_locked(tunes);
	}
	pthread_mutex_unlock(&g_scan_lock);
	return rc;
}

int rxgpu_scan_deferred(int on)
{
	int rc = RXGPU_OK;
	pthread_mutex_lock(&g_scan_lock);
	if (g_scan.dirty && !on)
		rc = rxgpu_fail(RXGPU_EINVAL, "rxgpu_scan_deferred(0): %d tunes are still accumulated on the device, rxgpu_scan_sync first", g_scan.tune_count);
	else
		g_scan.deferred = on ? 1 : 0;
	pthread_mutex_unlock(&g_scan_lock);
	return rc;
}

long rxgpu_scan_syncs(void) { return g_scan.syncs; }
int rxgpu_scan_zero_copy(void) { return g_scan.zc_last; }
int rxgpu_scan_sync_in_place(void) { return g_scan.zc_sync_last; }

/* One CSV row for a tuning_state, byte for byte what csv_dbm prints (rtl_power.c:774-817) -- a restatement, because
 * the text has to be identical: the bins are read through the index map the reference's in-place edits amount to
 * (bin 0 takes bin 1's value, then the two halves trade places), every floating-point expression keeps the
 * reference's operation order (two successive divisions per bin, one division by the product for the trailing
 * column), and like the reference the row's accumulators are cleared afterwards. */
static int64_t csv_bin(const struct tuning_state *ts, int len, int i)
{
	if (ts->bin_e == 0)
		return ts->avg[i];
	int src = i + len / 2;
	if (src >= len)
		src -= len;
	return ts->avg[src == 0 ? 1 : src];
}

void rxgpu_csv_dbm(struct tuning_state *ts, void *file)
{
	FILE *f = (FILE *)file;
	/* the row of a sweep whose sums are still on the device: bring them home first (once per interval -- the next rows find
	 * nothing pending) */
	pthread_mutex_lock(&g_scan_lock);
	if (g_scan.dirty && ts >= g_scan.tunes && ts < g_scan.tunes + g_scan.tune_count && scan_sync_locked(g_scan.tunes) != RXGPU_OK)
		fprintf(stderr, "rxgpu_csv_dbm: %s\n", rxgpu_last_error());
	pthread_mutex_unlock(&g_scan_lock);
	const int len = 1 << ts->bin_e, ds = ts->downsample;
	const int kept = (int)((double)len * (1.0 - ts->crop));
	const int half_bw = (int)(((double)ts->rate * (double)kept) / (len * 2 * ds));
	const int skip = (int)((double)len * ts->crop * 0.5);
	const int first = skip, last = (len - 1) - skip;
	fprintf(f, "%lli, %lli, %.2f, %i, ", (long long)ts->freq - half_bw, (long long)ts->freq + half_bw,
	        (double)ts->rate / (double)(len * ds), ts->samples);
	for (int i = first; i <= last; i++) {
		double v = (double)csv_bin(ts, len, i);
		v /= (double)ts->rate;
		v /= (double)ts->samples;
		fprintf(f, "%.2f, ", 10 * log10(v));
	}
	/* the trailing column repeats the last bin with the other operation order (rtl_power.c:808-813) */
	const double tail = (double)csv_bin(ts, len, ts->bin_e == 0 ? 0 : last) / ((double)ts->rate * (double)ts->samples);
	fprintf(f, "%.2f\n", 10 * log10(tail));
	memset(ts->avg, 0, (size_t)len * sizeof(ts->avg[0]));
	ts->samples = 0;
	(void)rxgpu_scan_rows_cleared(ts, 1);                  /* (the next merge into this row need not read it: scan_sync_locked) */
}

int rxgpu_scan_rows_cleared(const struct tuning_state *tunes, int tune_count)
{
	int marked = 0;
	if (!tunes || tune_count < 1)
		return 0;
	pthread_mutex_lock(&g_scan_lock);
	struct zc_table *z = &g_scan.zavg;
	for (int i = 0; i < tune_count && z->zero; i++)
		for (int k = 0; k < z->count; k++) {
			const int idx = (z->zero_hint + k) % z->count;
			if (z->host[idx] == (const void *)tunes[i].avg) {
				z->zero[idx] = 1;
				z->zero_hint = idx + 1;
				marked++;
				break;
			}
		}
	pthread_mutex_unlock(&g_scan_lock);
	return marked;
}
