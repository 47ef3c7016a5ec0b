/* rxgpu_chan.c -- host side of the rx_fm channeliser (extension; BASELINE configs[4]).
 *
 * The reference has one demod_state ("multiple of these, eventually", rtl_fm.c:189) and no mixer beyond
 * rotate16_90.  The channeliser is specified entirely from reference primitives: every window of
 * N = 2^bin_e capture samples through fix_fft (rtl_power.c:264-320) -- the bank of "mix by k*fs/N and
 * boxcar-sum N samples" channels, i.e. low_pass (rtl_fm.c:351-371) at ds = N for every offset at once,
 * with the reference's own fixed-point scaling -- and fm_demod (rtl_fm.c:584-615) per channel with its
 * own carried pre_r/pre_j, callback block after callback block.
 */
#include "rxgpu_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct rxgpu_chan {
	rxgpu_chan_params p;
	size_t max_windows;
	uint32_t *twiddle_dev;
	uint32_t *nco_tw_dev;            /* p.nco: the full period of the NCO, N packed (cos, sin) from the reference's Sinewave table */
	int *pre_up;                     /* carried (pre_r, pre_j) per channel as uploaded from pre_host (a run that is not chained to one before it) */
	int *pre_host;
	int *audio_dev[2], *audio_host;  /* per channel {deemph avg, now_lpr, prev_lpr_index}: in / out */
	int16_t *audio_y;                /* [n_channels][max_windows]: the de-emphasised samples in front of the resampler (k_ch_audio), or the demodulated rows (segmented form) */
	void *audio_ctab;                /* the (segment, channel) form of the audio stages: chunk tables [n_channels][chunks] ... */
	int *audio_seg;                  /* ... and every chunk's start state [n_channels][chunks] */
	/* what a run leaves for its retirement, two of them in rotation (rxgpu_chan_run_async keeps up to two runs in flight) */
	struct chan_slot {
		int live;
		rxk_fm_dev *dev, *dev_host;  /* flag count (device / its pinned copy, read back behind the run's kernels) */
		unsigned long long *flag_list;
		uint32_t *chan_lp;           /* the bins the sparse / dense demodulator pass reads -- and a host fix-up after it */
		int *pre_out_dev;            /* the run's carries out: the next run's carries in, on the device */
		int *pre_in_host, *pre_out_host;   /* pinned copies: carries in (a channel's first window re-evaluated on the host), carries out */
		int16_t *rows;               /* where the run's demodulated samples went */
		size_t rstride;
		unsigned long long total;
		int fused;
		hipEvent_t done;
	} run[2];
	unsigned long long seq;
	int chained;                     /* the previous enqueued run's pre_out_dev holds the carries (else pre_host does) */
	unsigned long long *flag_host;
	size_t last_windows;
	long fixups, fixups_pending;
};

static int chan_slots_alloc(rxgpu_chan *s, size_t nc)
{
	for (int k = 0; k < 2; k++) {
		struct chan_slot *r = &s->run[k];
		if (hipMalloc((void **)&r->dev, sizeof(rxk_fm_dev)) != hipSuccess ||
		    hipHostMalloc((void **)&r->dev_host, sizeof(rxk_fm_dev), 0) != hipSuccess ||
		    hipMalloc((void **)&r->flag_list, RXK_FLAG_CAP * 8) != hipSuccess ||
		    hipMalloc((void **)&r->chan_lp, nc * s->max_windows * 4) != hipSuccess ||
		    hipMalloc((void **)&r->pre_out_dev, nc * 8) != hipSuccess ||
		    hipHostMalloc((void **)&r->pre_in_host, nc * 8, 0) != hipSuccess ||
		    hipHostMalloc((void **)&r->pre_out_host, nc * 8, 0) != hipSuccess ||
		    hipEventCreateWithFlags(&r->done, hipEventDisableTiming) != hipSuccess)
			return RXGPU_ENOMEM;
	}
	return RXGPU_OK;
}

int rxgpu_chan_create(rxgpu_chan **out, const rxgpu_chan_params *p, size_t max_blocks, size_t block_len, const int16_t *sinewave)
{
	int rc;
	rxgpu_chan *s;
	if (!out || !p || !sinewave || !max_blocks || block_len < 2 || (block_len & 1))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_chan_create: bad arguments");
	if (p->bin_e < 1 || p->bin_e > 15)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "window of 2^%d samples outside 2^1..2^15", p->bin_e);
	const size_t n = (size_t)1 << p->bin_e;
	if ((block_len / 2) % n)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "block of %zu samples is not a whole number of %zu-sample windows", block_len / 2, n);
	if (p->n_channels < 1 || (size_t)p->n_channels > n || p->first_bin < 0 || (size_t)p->first_bin >= n)
		return rxgpu_fail(RXGPU_EINVAL, "channels [%d, %d) do not fit %zu bins", p->first_bin, p->first_bin + p->n_channels, n);
	if (p->custom_atan != 0 && p->custom_atan != 1)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "channeliser: -A std or fast only");
	if (p->deemph && p->deemph_a < 1)
		return rxgpu_fail(RXGPU_EINVAL, "deemph_a %d < 1", p->deemph_a);
	if (p->rate_out2 > 0 && (p->rate_out < p->rate_out2 || p->rate_out <= 0))
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "low_pass_real needs rate_out >= rate_out2 > 0 (got %d, %d)", p->rate_out, p->rate_out2);
	if (p->nco != 0 && p->nco != 1)
		return rxgpu_fail(RXGPU_EINVAL, "nco %d is neither 0 (bins of fix_fft) nor 1 (NCO -> low_pass)", p->nco);
	if (p->nco && (p->bin_e < 3 || p->bin_e > 12))
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "NCO mode: windows of 2^3 .. 2^12 samples (window and table are staged in LDS)");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	rxgpu_knobs_reload();                            /* the object keeps the kernel variants chosen now */
	s = calloc(1, sizeof(*s));
	if (!s)
		return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
	s->p = *p;
	s->max_windows = max_blocks * (block_len / 2 / n);
	uint32_t *tw = malloc((n + 2) * 4);
	if (!tw) { free(s); return rxgpu_fail(RXGPU_ENOMEM, "out of host memory"); }
	rxgpu_twiddle_table(sinewave, (int)n, tw);
	const size_t nc = (size_t)p->n_channels;
	/* chunk tables of the segmented audio stages: at most 8 segments of 256 chunks per channel, and never more chunks than the longest row holds */
	size_t ctab_per_channel = 2048;
	if (p->deemph && p->deemph_a >= 2 && p->deemph_a <= 64) {
		const size_t w8 = (size_t)((rxgpu_deemph_warm64(p->deemph_a) + 7) & ~7);
		if ((s->max_windows + w8 - 1) / w8 < ctab_per_channel)
			ctab_per_channel = (s->max_windows + w8 - 1) / w8 + 1;
	}
	if (hipMalloc((void **)&s->twiddle_dev, (n + 2) * 4) != hipSuccess ||
	    hipMalloc((void **)&s->pre_up, nc * 8) != hipSuccess ||
	    hipMalloc((void **)&s->audio_dev[0], nc * 12) != hipSuccess || hipMalloc((void **)&s->audio_dev[1], nc * 12) != hipSuccess ||
	    hipHostMalloc((void **)&s->audio_host, nc * 12, 0) != hipSuccess ||
	    ((p->rate_out2 > 0 || p->deemph) && hipMalloc((void **)&s->audio_y, nc * s->max_windows * 2) != hipSuccess) ||
	    (p->deemph && (hipMalloc(&s->audio_ctab, nc * ctab_per_channel * 16) != hipSuccess || hipMalloc((void **)&s->audio_seg, nc * ctab_per_channel * 4) != hipSuccess)) ||
	    chan_slots_alloc(s, nc) != RXGPU_OK ||
	    hipHostMalloc((void **)&s->flag_host, RXK_FLAG_CAP * 8, 0) != hipSuccess ||
	    hipHostMalloc((void **)&s->pre_host, nc * 8, 0) != hipSuccess ||
	    hipMemcpy(s->twiddle_dev, tw, (n + 2) * 4, hipMemcpyHostToDevice) != hipSuccess) {
		free(tw);
		rxgpu_chan_destroy(s);
		return rxgpu_fail(RXGPU_ENOMEM, "channeliser workspace allocation failed");
	}
	free(tw);
	if (p->nco) {
		/* cos(t) = sin(t + pi/2); the table holds 3/4 of a period (rtl_power.c:240-254): the second half period is the first, negated */
		uint32_t *full = malloc(n * 4);
		if (!full) { rxgpu_chan_destroy(s); return rxgpu_fail(RXGPU_ENOMEM, "out of host memory"); }
		for (size_t q = 0; q < n; q++) {
			const size_t h = n / 2, r = q & (h - 1);
			int co = sinewave[r + n / 4], si = sinewave[r];
			if (q >= h) { co = -co; si = -si; }
			full[q] = ((uint32_t)co & 0xffffu) | ((uint32_t)si << 16);
		}
		const int bad = hipMalloc((void **)&s->nco_tw_dev, n * 4) != hipSuccess || hipMemcpy(s->nco_tw_dev, full, n * 4, hipMemcpyHostToDevice) != hipSuccess;
		free(full);
		if (bad) { rxgpu_chan_destroy(s); return rxgpu_fail(RXGPU_ENOMEM, "channeliser workspace allocation failed"); }
	}
	memset(s->pre_host, 0, nc * 8);
	memset(s->audio_host, 0, nc * 12);
	*out = s;
	return RXGPU_OK;
}

void rxgpu_chan_destroy(rxgpu_chan *s)
{
	if (!s)
		return;
	hipFree(s->twiddle_dev); hipFree(s->nco_tw_dev); hipFree(s->pre_up);
	for (int k = 0; k < 2; k++) {
		struct chan_slot *r = &s->run[k];
		if (r->done) { hipEventSynchronize(r->done); hipEventDestroy(r->done); }
		hipFree(r->dev); hipFree(r->flag_list); hipFree(r->chan_lp); hipFree(r->pre_out_dev);
		if (r->dev_host) hipHostFree(r->dev_host);
		if (r->pre_in_host) hipHostFree(r->pre_in_host);
		if (r->pre_out_host) hipHostFree(r->pre_out_host);
	}
	hipFree(s->audio_dev[0]); hipFree(s->audio_dev[1]); hipFree(s->audio_y); hipFree(s->audio_ctab); hipFree(s->audio_seg);
	if (s->audio_host) hipHostFree(s->audio_host);
	if (s->flag_host) hipHostFree(s->flag_host);
	if (s->pre_host) hipHostFree(s->pre_host);
	free(s);
}

int rxgpu_chan_set_carry(rxgpu_chan *s, const int *pre)
{
	if (!s || !pre)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	if (s->run[0].live || s->run[1].live)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_chan_set_carry with runs in flight: rxgpu_chan_wait first");
	memcpy(s->pre_host, pre, (size_t)s->p.n_channels * 8);
	s->chained = 0;
	return RXGPU_OK;
}

int rxgpu_chan_get_carry(rxgpu_chan *s, int *pre)
{
	if (!s || !pre)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	/* pre_host is brought up to date when a run is retired (chan_drain): with runs in flight it still holds an older run's carries */
	if (s->run[0].live || s->run[1].live)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_chan_get_carry with runs in flight: rxgpu_chan_wait first");
	memcpy(pre, s->pre_host, (size_t)s->p.n_channels * 8);
	return RXGPU_OK;
}

int rxgpu_chan_set_audio_carry(rxgpu_chan *s, const int *audio)
{
	if (!s || !audio)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	memcpy(s->audio_host, audio, (size_t)s->p.n_channels * 12);
	return RXGPU_OK;
}

int rxgpu_chan_get_audio_carry(rxgpu_chan *s, int *audio)
{
	if (!s || !audio)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	if (s->run[0].live || s->run[1].live)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_chan_get_audio_carry with runs in flight: rxgpu_chan_wait first");
	memcpy(audio, s->audio_host, (size_t)s->p.n_channels * 12);
	return RXGPU_OK;
}

/* -1 while runs are in flight: the count is settled when a run is retired (rxgpu_chan_wait), like the carries */
long rxgpu_chan_host_fixups(const rxgpu_chan *s) { return !s ? 0 : (s->run[0].live || s->run[1].live) ? -1 : s->fixups; }

/* samples until trajectories from the two ends of the int16 range are fewer than `a` apart (then at most one adjacent pair
 * of candidates merges per sample, which is what the mask tracking relies on): the gap g shrinks by at least floor(g / a) per
 * sample (k_fm_deemph_scan's argument).  a <= 64, so the candidates also fit the 64-bit mask. */
int rxgpu_deemph_warm64(int a)
{
	int n = 0;
	long long g = 65535;
	while (g >= a && n < (1 << 20)) {
		g -= g / a;
		n++;
	}
	return (n + 8) / 8 * 8;
}

static int disc_host(int ar, int aj, int br, int bj)
{
	int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
	return (int)(atan2((double)cj, (double)cr) / 3.14159 * (1 << 14));
}

/* enqueue one run into slot `k`: carries in from the run before it (on the device) or from pre_host, FFT bank (+ fused discriminator),
 * the sparse / dense demodulator pass, flag count and carries out copied to pinned memory behind them */
static int chan_enqueue(rxgpu_chan *s, int k, const int16_t *d_iq, unsigned long long total, unsigned long long wpb, int16_t *rows, size_t rstride)
{
	hipStream_t st = rxgpu_hip_stream();
	struct chan_slot *r = &s->run[k];
	const size_t nc = (size_t)s->p.n_channels;
	const int *pre_in;
	memset(r->dev_host, 0, sizeof(*r->dev_host));
	RX_HIP(hipMemsetAsync(r->dev, 0, sizeof(rxk_fm_dev), st));
	if (s->chained) {
		pre_in = s->run[k ^ 1].pre_out_dev;
	} else {
		RX_HIP(hipMemcpyAsync(s->pre_up, s->pre_host, nc * 8, hipMemcpyHostToDevice, st));
		pre_in = s->pre_up;
	}
	RX_HIP(hipMemcpyAsync(r->pre_in_host, pre_in, nc * 8, hipMemcpyDeviceToHost, st));
	/* -A fast with whole groups of windows per block: fm_demod runs inside the FFT kernel for all but each group's first window */
	const int fused = s->p.nco ? 0 : rxk_ch_fused_ok(s->p.bin_e, wpb, s->p.custom_atan, s->p.n_channels);
	rxgpu_prof_begin("ch_fft");
	if (s->p.nco)                                         /* SURVEY 8(f)2's literal definition: NCO -> low_pass at downsample N, per channel */
		RX_K(rxk_ch_nco(st, d_iq, total, s->p.bin_e, s->nco_tw_dev, s->p.first_bin, s->p.n_channels, r->chan_lp));
	else
		RX_K(rxk_ch_fft(st, d_iq, total, s->p.bin_e, s->twiddle_dev, s->p.first_bin, s->p.n_channels, r->chan_lp, fused, rows, rstride,
		                r->pre_out_dev));
	rxgpu_prof_end("ch_fft");
	rxgpu_prof_begin("ch_demod");
	RX_K(rxk_ch_demod(st, r->chan_lp, total, wpb, s->p.n_channels, s->p.custom_atan, pre_in, r->pre_out_dev, rows, rstride,
	                  r->dev, r->flag_list, fused));
	rxgpu_prof_end("ch_demod");
	RX_HIP(hipMemcpyAsync(r->dev_host, r->dev, sizeof(rxk_fm_dev), hipMemcpyDeviceToHost, st));
	RX_HIP(hipMemcpyAsync(r->pre_out_host, r->pre_out_dev, nc * 8, hipMemcpyDeviceToHost, st));
	RX_HIP(hipEventRecord(r->done, st));
	r->rows = rows; r->rstride = rstride; r->total = total; r->fused = fused;
	r->live = 1;
	s->chained = 1;
	return RXGPU_OK;
}

/* wait for the run of slot `k`; libm samples the device could not decide go through the host's libm (like rxgpu_fm.c) and are patched into
 * the run's rows -- the carries are the channels' last bins, which no fix-up changes, so the run behind it needs nothing redone */
static int chan_retire(rxgpu_chan *s, int k)
{
	struct chan_slot *r = &s->run[k];
	if (!r->live)
		return RXGPU_OK;
	RX_HIP(hipEventSynchronize(r->done));
	r->live = 0;
	rxgpu_prof_collect();
	const size_t nc = (size_t)s->p.n_channels;
	const unsigned long long total = r->total;
	const int fused = r->fused;
	const int cnt = r->dev_host->flag_cnt;
	if (cnt) {
		if (cnt > RXK_FLAG_CAP)
			return rxgpu_fail(RXGPU_EUNSUPPORTED, "%d undecided libm discriminator samples (cap %d)", cnt, RXK_FLAG_CAP);
		RX_HIP(hipMemcpy(s->flag_host, r->flag_list, (size_t)cnt * 8, hipMemcpyDeviceToHost));
		for (int i = 0; i < cnt; i++) {
			unsigned long long gid = s->flag_host[i], c = gid / total, t = gid - c * total;
			uint32_t a, b;
			int br, bj;
			/* fused: the FFT kernel kept [channel][run] first windows, then [channel][run] last windows (flagged samples are block starts,
			 * hence run starts); else the dense [channel][window] array */
			const unsigned long long runs = fused ? total / (unsigned long long)fused : 0, q = fused ? t / (unsigned long long)fused : 0;
			const uint32_t *pa = fused ? r->chan_lp + c * runs + q : r->chan_lp + gid;
			const uint32_t *pb = fused ? r->chan_lp + nc * runs + c * runs + q - 1 : r->chan_lp + gid - 1;
			RX_HIP(hipMemcpy(&a, pa, 4, hipMemcpyDeviceToHost));
			if (t) {
				RX_HIP(hipMemcpy(&b, pb, 4, hipMemcpyDeviceToHost));
				br = (int16_t)(b & 0xffff); bj = (int16_t)(b >> 16);
			} else {
				br = r->pre_in_host[2 * c]; bj = r->pre_in_host[2 * c + 1];
			}
			int16_t v = (int16_t)disc_host((int16_t)(a & 0xffff), (int16_t)(a >> 16), br, bj);
			RX_HIP(hipMemcpy(r->rows + c * r->rstride + t, &v, 2, hipMemcpyHostToDevice));
		}
		s->fixups_pending += cnt;
	}
	return RXGPU_OK;
}

static int chan_check_run(rxgpu_chan *s, const int16_t *d_iq, size_t n_blocks, size_t block_len, int16_t *d_out, size_t out_stride,
                          unsigned long long *wpb_out, unsigned long long *total_out)
{
	if (!s || !d_iq || !d_out || !n_blocks)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_chan_run: bad arguments");
	const size_t n = (size_t)1 << s->p.bin_e;
	if (block_len < 2 || (block_len & 1) || (block_len / 2) % n)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "block of %zu samples is not a whole number of %zu-sample windows", block_len / 2, n);
	const unsigned long long wpb = block_len / 2 / n, total = wpb * n_blocks;
	if (total > s->max_windows)
		return rxgpu_fail(RXGPU_ECAPACITY, "channeliser created for %zu windows, run asks %llu", s->max_windows, total);
	if (out_stride < total)
		return rxgpu_fail(RXGPU_ECAPACITY, "out_stride %zu shorter than %llu windows", out_stride, total);
	*wpb_out = wpb; *total_out = total;
	return RXGPU_OK;
}

/* the two runs in flight, older first; the carries of the newest come home */
static int chan_drain(rxgpu_chan *s)
{
	int rc;
	const int newest = (int)((s->seq + 1) & 1);              /* slot of run seq - 1 */
	if ((rc = chan_retire(s, newest ^ 1)) != RXGPU_OK || (rc = chan_retire(s, newest)) != RXGPU_OK)
		return rc;
	if (s->seq && s->chained)
		memcpy(s->pre_host, s->run[newest].pre_out_host, (size_t)s->p.n_channels * 8);
	s->fixups = s->fixups_pending;
	s->fixups_pending = 0;
	return RXGPU_OK;
}

int rxgpu_chan_run_async(rxgpu_chan *s, const int16_t *d_iq, size_t n_blocks, size_t block_len, int16_t *d_out, size_t out_stride)
{
	int rc;
	unsigned long long wpb, total;
	if ((rc = chan_check_run(s, d_iq, n_blocks, block_len, d_out, out_stride, &wpb, &total)) != RXGPU_OK)
		return rc;
	if (s->p.deemph || s->p.rate_out2 > 0)
		/* the audio stages start from samples a host fix-up may still change: such a channeliser runs call by call */
		return rxgpu_chan_run(s, d_iq, n_blocks, block_len, d_out, out_stride, &s->last_windows);
	const int k = (int)(s->seq & 1);
	if ((rc = chan_retire(s, k)) != RXGPU_OK)                /* the run two enqueues ago: its slot is this run's */
		return rc;
	if ((rc = chan_enqueue(s, k, d_iq, total, wpb, d_out, out_stride)) != RXGPU_OK)
		return rc;
	s->seq++;
	s->last_windows = (size_t)total;
	return RXGPU_OK;
}

int rxgpu_chan_wait(rxgpu_chan *s, size_t *windows_out)
{
	int rc;
	if (!s)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	if ((rc = chan_drain(s)) != RXGPU_OK)
		return rc;
	if (windows_out)
		*windows_out = s->last_windows;
	return RXGPU_OK;
}

int rxgpu_chan_run(rxgpu_chan *s, const int16_t *d_iq, size_t n_blocks, size_t block_len, int16_t *d_out, size_t out_stride,
                   size_t *windows_out)
{
	int rc;
	unsigned long long wpb, total;
	if ((rc = chan_check_run(s, d_iq, n_blocks, block_len, d_out, out_stride, &wpb, &total)) != RXGPU_OK)
		return rc;
	hipStream_t st = rxgpu_hip_stream();
	const size_t nc = (size_t)s->p.n_channels;
	if ((rc = chan_drain(s)) != RXGPU_OK)                    /* runs a caller left in flight */
		return rc;
	/* per-channel audio stages: which form serves this run is known before anything is launched -- the (segment, channel) grid reads the
	 * demodulated rows from a buffer of its own (audio_y) and writes the audio to d_out, k_ch_audio works on d_out in place */
	const int audio_on = s->p.deemph || s->p.rate_out2 > 0;
	int serial = s->p.deemph && (s->p.deemph_a < 2 || s->p.deemph_a > 64);
	for (size_t c = 0; c < nc && s->p.deemph && !serial; c++)
		if (s->audio_host[3 * c] < -32768 || s->audio_host[3 * c] > 32767)
			serial = 1;
	const int warm = s->p.deemph && !serial ? rxgpu_deemph_warm64(s->p.deemph_a) : 8;
	const int seg = audio_on && s->p.deemph && !serial && s->audio_y &&
	                rxk_ch_audio_seg_ok(total, warm, s->p.rate_out, s->p.rate_out2 > 0 ? s->p.rate_out2 : 0);
	int16_t *const rows = seg ? s->audio_y : d_out;                 /* where the demodulated samples go */
	const size_t rstride = seg ? s->max_windows : out_stride;
	const int k = (int)(s->seq & 1);
	if ((rc = chan_enqueue(s, k, d_iq, total, wpb, rows, rstride)) != RXGPU_OK)
		return rc;
	s->seq++;
	if ((rc = chan_drain(s)) != RXGPU_OK)
		return rc;
	unsigned long long per_channel = total;
	if (audio_on) {
		/* per-channel audio stages on the finished (and, where needed, host-corrected) demodulated rows */
		unsigned long long J = total;
		if (s->p.rate_out2 > 0) {
			const int p0 = s->audio_host[2];                  /* the phase advances alike in every channel */
			for (size_t c = 0; c < nc; c++)
				if (s->audio_host[3 * c + 2] != p0 || p0 < 0 || p0 >= s->p.rate_out)
					return rxgpu_fail(RXGPU_EINVAL, "prev_lpr_index must be the same in [0, rate_out) for every channel");
			J = ((unsigned long long)p0 + total * (unsigned long long)s->p.rate_out2) / (unsigned long long)s->p.rate_out;
		}
		RX_HIP(hipMemcpyAsync(s->audio_dev[0], s->audio_host, nc * 12, hipMemcpyHostToDevice, st));
		rxgpu_prof_begin("ch_audio");
		if (seg)
			/* rows long enough to cut into segments: chunk tables per (segment, channel), composed in a second level; low_pass_real inline */
			RX_K(rxk_ch_audio_seg(st, rows, rstride, d_out, out_stride, total, s->p.n_channels, s->p.deemph_a, warm, s->p.rate_out,
			                      s->p.rate_out2 > 0 ? s->p.rate_out2 : 0, s->audio_dev[0], s->audio_dev[1], s->audio_ctab, s->audio_seg));
		else
			RX_K(rxk_ch_audio(st, d_out, out_stride, total, s->p.n_channels, s->p.deemph, s->p.deemph_a, warm, serial, s->p.rate_out,
			                  s->p.rate_out2 > 0 ? s->p.rate_out2 : 0, J, s->audio_dev[0], s->audio_dev[1], s->audio_y, s->max_windows));
		rxgpu_prof_end("ch_audio");
		RX_HIP(hipMemcpyAsync(s->audio_host, s->audio_dev[1], nc * 12, hipMemcpyDeviceToHost, st));
		RX_HIP(hipStreamSynchronize(st));
		rxgpu_prof_collect();
		per_channel = J;
	}
	s->last_windows = (size_t)per_channel;
	if (windows_out)
		*windows_out = (size_t)per_channel;
	return RXGPU_OK;
}
