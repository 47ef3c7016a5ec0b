/* rxgpu_fm.c -- host side of the rx_fm path: the batched stream object and the two drop-in
 * entry points.  C over the HIP C API; all arithmetic on samples happens in fm_kernels.hip.
 *
 * Reference call sites replaced (under /root/reference/src):
 *   full_demod(d)                rtl_fm.c:923  (definition 759-824)
 *   rtlsdr_callback(buf,len,ctx) rtl_fm.c:899  (definition 828-863)
 */
#include "rxgpu_internal.h"
#include "rxgpu_ref_structs.h"
#include <math.h>
#include <limits.h>
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define DEEMPH_CHUNK_MIN 128           /* chunk = power of two >= warm: one lane per chunk, 64 chunks per workgroup */
#define DEEMPH_LEVELS 8
/* tables the single-workgroup top walk stages in LDS.  Kept small (<= 16 KiB of LDS, 256 threads) so that
 * the workgroup finds a slot on a CU that the pipelined decimator of the next run is saturating. */
#define DEEMPH_TOPCAP(group) ((group) == 16 ? 224 : 56)

/* geometry of one run, everything the host can know without the device */
struct run_geom {
	unsigned long long n, T, M, K, J;
	int passes, ds, p0, pr0, rotate, fast, post;
	int rdc_fused;                       /* -E rdc: the decimator subtracts the block averages itself (no corrected copy of the capture) */
	int literal;                         /* -F on blocks that are not whole tiles: the per-block int16-indexed kernels */
	int lf;                              /* literal: int16 count of a block's lowpassed[] after the cascade, (2n) >> passes (may be odd) */
};

struct rxgpu_fm_stream {
	rxgpu_fm_params p;
	rxgpu_fm_carry carry;
	size_t max_blocks, block_len;        /* capacity */
	size_t max_T, max_M;
	/* device workspaces */
	uint32_t *lp_raw[2], *head[2], *tail[2];   /* decimator outputs, double-buffered across pipelined runs */
	uint32_t *lp;
	const uint32_t *lp_final;            /* where the last run left the final decimated IQ */
	uint32_t *cas[2];                    /* fifth_order ping-pong */
	uint32_t *seams;                     /* per block: 3 levels x 5 history samples for the fused passes */
	uint32_t *cas_a[2], *seams_a[2];     /* raw input: the first fused group runs on stream A like the decimator, double-buffered */
	uint32_t *edges_a[2];                /* ... and where it also demodulates: each block's first and last FIR output */
	int last_fuse_a, last_fuse_dd;       /* how the previous run split its histories between the seam stream and stream B */
	hipEvent_t ev_up;                    /* carries uploaded on stream B -> stream A may read the cascade history */
	hipEvent_t ev_seam[2];               /* the seam histories of a run's first fused group are in place (stream 4 -> stream A) */
	int16_t *pcm_buf[2], *pcm, *y;       /* pcm: the buffer the run in hand uses (double-buffered like lp_raw) */
	int16_t *rdc_buf, *pcm_post;         /* -E rdc: the corrected, rotated capture; -o: pcm after low_pass_simple */
	long long *rdc_sums;
	int *rdc_avg, *rdc_state;
	hipEvent_t ev_rdc;
	int *chunk_pre;                                /* per chunk: its start state for each candidate of its workgroup (scan -> apply) */
	void *ctab;                                    /* tiled path: compact 16-byte chunk tables (scan_t -> up0 / down0) */
	int *cstart;                                   /* tiled path: exact start state per chunk (down0 -> apply_rs_t) */
	int tiled;                                     /* pcm[] of the run in hand is in the tiled layout: chunk log2, else 0 */
	int row_audio;                                 /* the run in hand is one short block whose audio stages are rxk_fm_row_audio's one launch */
	int *lvl_tab, *lvl_lo, *lvl_gap, *lvl_start;   /* tree levels, packed back to back */
	size_t lvl_cap;
	/* libm samples the device could not decide, per run slot (seq & 1): count + records on the device, mirrors in pinned memory */
	int *flag_cnt_dev;                   /* [2] */
	rxk_flag_rec *flag_rec_dev;          /* [2][RXK_FLAG_CAP] */
	int *flag_cnt_host;                  /* [2], pinned */
	rxk_flag_rec *flag_rec_host;         /* [RXK_FLAG_CAP], pinned */
	int *snap_dev;                       /* [2][4]: the audio-stage carries-in of the run in each slot */
	int *atan_lut;                       /* -A lut table (rtl_fm.c:515-526), only when custom_atan == 2 */
	int *below;                          /* squelch verdict per block */
	long long *dc_sums;                  /* dc_block_audio: per-block sums and means */
	int *dc_avgs;
	int *below_host;
	rxk_fm_dev *dev;
	int16_t *hist_dev;                   /* [10][12] cascade hist in, [10][12] out, [18] droop in, [18] out */
	int *fir_dev;                        /* 10 ints */
	int fir_loaded;                      /* cascade depth whose cic_9_tables row is in fir_dev (0: none) */
	/* pinned host mirrors */
	rxk_fm_dev *dev_host;
	int16_t *hist_host;
	/* device staging for run_host: three input chunks in rotation (a chunk must stay put until its run is retired,
	 * i.e. until the run two behind it is enqueued), one output buffer for the whole call */
	int16_t *stage_in[3], *stage_out;
	size_t stage_in_cap, stage_out_cap;
	hipEvent_t ev_h2d[3];
	/* de-emphasis geometry */
	int group, warm, lo0, hi0, gap_w;
	int chunk;                           /* de-emphasis scan: samples per chunk */
	int topcap_override;                 /* $RXGPU_DEEMPH_TOPCAP: forces the multi-level scan (tests) */
	/* plan knobs, read once at creation (rxgpu_knob): $RXGPU_DEEMPH_CHUNK, $RXGPU_HOST_CHUNK */
	int k_deemph_chunk;
	unsigned long long k_host_chunk;
	/* one_stream (the drop-in's per-block streams): stream A, stream B and the seam stream are ONE stream -- a run of one block has nothing to
	 * overlap, and every hop between streams is an event the next kernel waits behind (round 6: 84 -> see profiles/r06_dropin_latency.txt) */
	int one_stream;
	int16_t *lp_mirror;                  /* one_stream runs: a page-locked host mirror the run's decimated IQ (lp_final) is written into at the end of the run */
	int flag_all;                        /* $RXGPU_FLAG_ALL: every libm discriminator sample goes through the host re-evaluation (tests) */
	int allow_empty;                     /* the drop-in: a block that yields no decimated sample is legal (the struct-memory reads the
	                                      * reference then makes are reproduced by rxgpu_full_demod on the real struct) */
	int16_t *lit[2];                     /* literal -F path: one block's int16 lowpassed[], ping-pong */
	long fixups;
	/* pipelining state: at most two runs in flight, run `seq` uses buffer set / slot seq & 1 */
	hipEvent_t ev_dec[2], ev_small[2], ev_disc[2];
	int ev_small_valid[2];
	unsigned long long seq;              /* runs enqueued so far */
	int pending;                         /* runs enqueued and not yet waited for */
	int carry_lost;                      /* a sequence failed while being retired: the host copy of the carries is stale until set_carry */
	int chained;                         /* device carries are ahead of the host copy */
	int h_prev_index, h_prev_lpr_index;  /* the two carries the host can track in closed form */
	struct run_rec {
		int live;                        /* enqueued, not retired */
		unsigned long long seq;
		struct run_geom g;
		int tiled, row_audio;
		int16_t *d_out, *pcm;
		rxk_fm_blocks blk;
		size_t n_blocks;
	} rec[2];
	struct run_geom last;
	rxk_fm_blocks blk;                   /* block ownership of the run in hand */
	size_t last_n_blocks;
};

/* rtl_fm.c:288-300 */
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

#define HIST_CAS_IN   0
#define HIST_CAS_OUT  (10 * 12)
#define HIST_DROOP_IN (20 * 12)
#define HIST_DROOP_OUT (20 * 12 + 18)
#define HIST_TOTAL    (20 * 12 + 36)

static int validate_params(const rxgpu_fm_params *p)
{
	if (p->downsample_passes < 0 || p->downsample_passes > 10)
		return rxgpu_fail(RXGPU_EINVAL, "downsample_passes %d outside 0..10", p->downsample_passes);
	if (!p->downsample_passes && p->downsample < 1)
		return rxgpu_fail(RXGPU_EINVAL, "downsample %d < 1", p->downsample);
	if (p->custom_atan < 0 || p->custom_atan > 3)
		return rxgpu_fail(RXGPU_EINVAL, "custom_atan %d outside 0..3", p->custom_atan);
	if (p->mode < RXGPU_MODE_FM || p->mode > RXGPU_MODE_RAW)
		return rxgpu_fail(RXGPU_EINVAL, "mode %d outside 0..4", p->mode);
	if (p->dc_block_audio && p->adc_block_const < 0)
		return rxgpu_fail(RXGPU_EINVAL, "adc_block_const %d < 0", p->adc_block_const);
	if (p->post_downsample < 0 || p->post_downsample > 16)                 /* MAXIMUM_OVERSAMPLE, rtl_fm.c:79 */
		return rxgpu_fail(RXGPU_EINVAL, "post_downsample %d outside 0..16", p->post_downsample);
	if (p->dc_block_raw && p->rdc_block_const < 0)
		return rxgpu_fail(RXGPU_EINVAL, "rdc_block_const %d < 0", p->rdc_block_const);
	if (p->dc_block_raw && p->prescaled)
		return rxgpu_fail(RXGPU_EINVAL, "dc_block_raw works on the raw capture (it is part of the callback), not on prescaled input");
	if (p->deemph && p->deemph_a < 1)
		return rxgpu_fail(RXGPU_EINVAL, "deemph_a %d < 1", p->deemph_a);
	if (p->rate_out2 > 0 && (p->rate_out < p->rate_out2 || p->rate_out <= 0))
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "low_pass_real needs rate_out >= rate_out2 > 0 (got %d, %d)", p->rate_out, p->rate_out2);
	return RXGPU_OK;
}

/* worst-case samples until trajectories from the two ends of [lo0,hi0] are < a apart:
 * the gap g shrinks by at least floor(g/a) per sample (see k_fm_deemph_scan) */
static int deemph_warm(int a, long long range, int *gap_end)
{
	int n = 0;
	long long g = range;
	while (g >= a) {
		g -= g / a;
		n++;
	}
	if (gap_end)
		*gap_end = (int)g;            /* < a: what the bound has come down to (k_fm_deemph_scan_t takes it as the candidate count) */
	return n + 1;
}

static void deemph_geometry(rxgpu_fm_stream *s)
{
	int a = s->p.deemph_a, avg = s->carry.deemph_avg;
	s->lo0 = avg < -32768 ? avg : -32768;
	s->hi0 = avg > 32767 ? avg : 32767;
	s->group = a <= 16 ? 16 : (a <= 64 ? 64 : 0);
	if (a < 2 || avg < -32768 || avg > 32767)
		s->group = 0;                 /* a == 1 or a carried state outside int16: the serial kernel */
	s->gap_w = 0;
	s->warm = s->group ? (deemph_warm(a, (long long)s->hi0 - s->lo0, &s->gap_w) + 7) / 8 * 8 : 0;   /* whole 16-byte reads */
	s->chunk = DEEMPH_CHUNK_MIN;
	while (s->chunk < s->warm)
		s->chunk *= 2;
	if (s->chunk > 1024)
		s->group = 0;                 /* cannot happen for a <= 64 and an int16 state; the serial kernel if it does */
}

int rxgpu_fm_stream_create(rxgpu_fm_stream **out, const rxgpu_fm_params *params, size_t max_blocks, size_t block_len)
{
	int rc;
	rxgpu_fm_stream *s;
	if (!out || !params || !max_blocks || block_len < 2 || (block_len & 1))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_fm_stream_create: bad arguments");
	if ((rc = validate_params(params)) != RXGPU_OK)
		return rc;
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	s = calloc(1, sizeof(*s));
	if (!s)
		return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
	s->p = *params;
	if (!s->p.output_scale)
		s->p.output_scale = 1;
	{
		/* the knobs that shape this stream's launch plan, fixed for its life: chained runs never see one flip */
		rxgpu_knobs_reload();
		const char *e = rxgpu_knob("RXGPU_DEEMPH_TOPCAP");
		s->topcap_override = (e && atoi(e) > 0) ? atoi(e) : 0;
		e = rxgpu_knob("RXGPU_FLAG_ALL");
		s->flag_all = (e && atoi(e) > 0) ? atoi(e) : 0;
		e = rxgpu_knob("RXGPU_DEEMPH_CHUNK");
		s->k_deemph_chunk = e ? atoi(e) : 0;
		e = rxgpu_knob("RXGPU_HOST_CHUNK");
		s->k_host_chunk = e ? strtoull(e, NULL, 10) : 0;
	}
	s->max_blocks = max_blocks;
	s->block_len = block_len;
	s->max_T = max_blocks * (block_len / 2);
	if (params->downsample_passes)
		s->max_M = max_blocks * (((block_len / 2) >> params->downsample_passes) + 1);
	else if (params->downsample < 1)
		s->max_M = s->max_T + 2;
	else
		s->max_M = s->max_T / (size_t)params->downsample + 2;
	size_t n_wg = (s->max_T + RXK_DEC_SPAN - 1) / RXK_DEC_SPAN + 1;
	size_t n_chunks = (s->max_M + DEEMPH_CHUNK_MIN - 1) / DEEMPH_CHUNK_MIN + 1;   /* the smallest chunk = the most tables */
#define DMALLOC(ptr, bytes) do { if (hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) { \
	rxgpu_fm_stream_destroy(s); return rxgpu_fail(RXGPU_ENOMEM, "hipMalloc(%zu) failed", (size_t)(bytes)); } } while (0)
	for (int i = 0; i < 2; i++) {
		DMALLOC(s->lp_raw[i], s->max_M * 4);
		DMALLOC(s->head[i], n_wg * 4);
		DMALLOC(s->tail[i], n_wg * 4);
		if (hipEventCreateWithFlags(&s->ev_dec[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_disc[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_small[i], hipEventDisableTiming) != hipSuccess) {
			rxgpu_fm_stream_destroy(s);
			return rxgpu_fail(RXGPU_ENODEV, "hipEventCreate failed");
		}
	}
	DMALLOC(s->lp, s->max_M * 4);
	const size_t pcm_cap = (s->max_M + 64 * 256 - 1) / (64 * 256) * (64 * 256) + 64 * 256;   /* whole tiles of the tiled layout */
	DMALLOC(s->pcm_buf[0], pcm_cap * 2);
	DMALLOC(s->pcm_buf[1], pcm_cap * 2);
	s->pcm = s->pcm_buf[0];
	DMALLOC(s->y, s->max_M * 2);
	const size_t n_l0 = n_chunks / RXK_DEEMPH_FAN + 2;                                    /* the tiled path's first level: 16 chunks per table */
	s->lvl_cap = n_l0 + n_l0 / (RXK_DEEMPH_FAN - 1) + 2 * DEEMPH_LEVELS + 2;             /* level 0 + all composites */
	DMALLOC(s->chunk_pre, n_chunks * (size_t)(params->deemph_a <= 16 ? 16 : 64) * 4);
	DMALLOC(s->ctab, n_chunks * 16);
	DMALLOC(s->cstart, n_chunks * 4);
	if (params->dc_block_raw) {
		DMALLOC(s->rdc_buf, s->max_T * 4);
		DMALLOC(s->rdc_sums, max_blocks * 16);
		DMALLOC(s->rdc_avg, max_blocks * 8);
		DMALLOC(s->rdc_state, 8);
		if (hipEventCreateWithFlags(&s->ev_rdc, hipEventDisableTiming) != hipSuccess) {
			rxgpu_fm_stream_destroy(s);
			return rxgpu_fail(RXGPU_ENOMEM, "hipEventCreate failed");
		}
	}
	if (params->post_downsample > 1)
		DMALLOC(s->pcm_post, s->max_M * 2);
	DMALLOC(s->lvl_tab, s->lvl_cap * 64 * 4);
	DMALLOC(s->lvl_lo, s->lvl_cap * 4);
	DMALLOC(s->lvl_gap, s->lvl_cap * 4);
	DMALLOC(s->lvl_start, s->lvl_cap * 4);
	DMALLOC(s->flag_cnt_dev, 2 * sizeof(int));
	DMALLOC(s->flag_rec_dev, 2 * RXK_FLAG_CAP * sizeof(rxk_flag_rec));
	DMALLOC(s->snap_dev, 8 * sizeof(int));
	DMALLOC(s->below, 2 * (max_blocks + 1) * 4);           /* verdicts, then the rms values behind them */
	DMALLOC(s->dc_sums, (max_blocks + 1) * 8);
	DMALLOC(s->dc_avgs, (max_blocks + 1) * 4);
	if (params->custom_atan == 2) {
		/* atan_lut_init, rtl_fm.c:515-526, with the host libm the reference uses */
		int *lut = malloc(131072 * sizeof(int));
		if (!lut) { rxgpu_fm_stream_destroy(s); return rxgpu_fail(RXGPU_ENOMEM, "out of host memory"); }
		for (int i = 0; i < 131072; i++)
			lut[i] = (int)(atan((double)i / (1 << 8)) / 3.14159 * (1 << 14));
		DMALLOC(s->atan_lut, 131072 * sizeof(int));
		hipError_t e = hipMemcpy(s->atan_lut, lut, 131072 * sizeof(int), hipMemcpyHostToDevice);
		free(lut);
		if (e != hipSuccess) { rxgpu_fm_stream_destroy(s); return rxgpu_fail(RXGPU_ENODEV, "atan table upload failed"); }
	}
	DMALLOC(s->dev, sizeof(rxk_fm_dev));
	DMALLOC(s->hist_dev, HIST_TOTAL * 2);
	DMALLOC(s->fir_dev, 10 * 4);
	if (params->downsample_passes) {
		DMALLOC(s->lit[0], block_len * 2 + 64);
		DMALLOC(s->lit[1], block_len * 2 + 64);
		/* pass 0 output is half the input, later passes shrink further: two buffers suffice */
		DMALLOC(s->cas[0], (s->max_T / 2 + max_blocks) * 4);
		DMALLOC(s->cas[1], (s->max_T / 4 + max_blocks) * 4);
		DMALLOC(s->seams, 4 * (max_blocks + 1) * 15 * 4);               /* one region per fused group of a run */
		if (!params->prescaled) {
			const int fuse = params->downsample_passes < 3 ? params->downsample_passes : 3;
			for (int i = 0; i < 2; i++) {
				DMALLOC(s->cas_a[i], ((s->max_T >> fuse) + max_blocks) * 4);
				DMALLOC(s->seams_a[i], (max_blocks + 1) * 25 * 4);             /* up to five levels of five history samples per block, or three and ten tail samples */
				DMALLOC(s->edges_a[i], (max_blocks + 1) * 2 * 4);
			}
			if (hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&s->ev_seam[0], hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&s->ev_seam[1], hipEventDisableTiming) != hipSuccess) {
				rxgpu_fm_stream_destroy(s);
				return rxgpu_fail(RXGPU_ENODEV, "hipEventCreate failed");
			}
		}
	}
	if (hipHostMalloc((void **)&s->dev_host, sizeof(rxk_fm_dev), 0) != hipSuccess ||
	    hipHostMalloc((void **)&s->hist_host, HIST_TOTAL * 2, 0) != hipSuccess ||
	    hipHostMalloc((void **)&s->flag_rec_host, RXK_FLAG_CAP * sizeof(rxk_flag_rec), 0) != hipSuccess ||
	    hipHostMalloc((void **)&s->flag_cnt_host, 2 * sizeof(int), 0) != hipSuccess ||
	    hipHostMalloc((void **)&s->below_host, 2 * (max_blocks + 1) * 4, 0) != hipSuccess) {
		rxgpu_fm_stream_destroy(s);
		return rxgpu_fail(RXGPU_ENOMEM, "hipHostMalloc failed");
	}
	s->flag_cnt_host[0] = s->flag_cnt_host[1] = 0;
	*out = s;
	return RXGPU_OK;
}

void rxgpu_fm_stream_destroy(rxgpu_fm_stream *s)
{
	if (!s)
		return;
	if (rxgpu_hip_stream()) {
		hipStreamSynchronize(rxgpu_hip_stream());
		hipStreamSynchronize(rxgpu_hip_stream2());
		hipStreamSynchronize(rxgpu_hip_stream3());
		hipStreamSynchronize(rxgpu_hip_stream4());
	}
	for (int i = 0; i < 2; i++) {
		hipFree(s->lp_raw[i]); hipFree(s->head[i]); hipFree(s->tail[i]);
		if (s->ev_dec[i]) hipEventDestroy(s->ev_dec[i]);
		if (s->ev_small[i]) hipEventDestroy(s->ev_small[i]);
		if (s->ev_disc[i]) hipEventDestroy(s->ev_disc[i]);
	}
	for (int i = 0; i < 3; i++) {
		hipFree(s->stage_in[i]);
		if (s->ev_h2d[i]) hipEventDestroy(s->ev_h2d[i]);
	}
	hipFree(s->lp);
	hipFree(s->cas[0]); hipFree(s->cas[1]); hipFree(s->seams);
	hipFree(s->lit[0]); hipFree(s->lit[1]);
	hipFree(s->cas_a[0]); hipFree(s->cas_a[1]); hipFree(s->seams_a[0]); hipFree(s->seams_a[1]); hipFree(s->edges_a[0]); hipFree(s->edges_a[1]);
	if (s->ev_up) hipEventDestroy(s->ev_up);
	if (s->ev_seam[0]) hipEventDestroy(s->ev_seam[0]);
	if (s->ev_seam[1]) hipEventDestroy(s->ev_seam[1]);
	hipFree(s->pcm_buf[0]); hipFree(s->pcm_buf[1]); hipFree(s->y);
	
	hipFree(s->lvl_tab); hipFree(s->lvl_lo); hipFree(s->lvl_gap); hipFree(s->lvl_start); hipFree(s->chunk_pre);
	hipFree(s->ctab); hipFree(s->cstart);
	hipFree(s->rdc_buf); hipFree(s->pcm_post); hipFree(s->rdc_sums); hipFree(s->rdc_avg); hipFree(s->rdc_state);
	if (s->ev_rdc) hipEventDestroy(s->ev_rdc);
	hipFree(s->atan_lut); hipFree(s->below); hipFree(s->dc_sums); hipFree(s->dc_avgs);
	if (s->below_host) hipHostFree(s->below_host);
	hipFree(s->flag_cnt_dev); hipFree(s->flag_rec_dev); hipFree(s->snap_dev);
	hipFree(s->dev); hipFree(s->hist_dev); hipFree(s->fir_dev);
	if (s->dev_host) hipHostFree(s->dev_host);
	if (s->hist_host) hipHostFree(s->hist_host);
	if (s->flag_rec_host) hipHostFree(s->flag_rec_host);
	if (s->flag_cnt_host) hipHostFree(s->flag_cnt_host);
	if (s->stage_out) hipFree(s->stage_out);
	free(s);
}

static int finish_runs(rxgpu_fm_stream *s);

int rxgpu_fm_stream_set_carry(rxgpu_fm_stream *s, const rxgpu_fm_carry *c)
{
	int rc;
	if (!s || !c)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	if (s->pending && (rc = finish_runs(s)) != RXGPU_OK && !s->carry_lost)
		return rc;
	s->carry = *c;
	s->carry_lost = 0;
	return RXGPU_OK;
}

int rxgpu_fm_stream_get_carry(rxgpu_fm_stream *s, rxgpu_fm_carry *c)
{
	int rc;
	if (!s || !c)
		return rxgpu_fail(RXGPU_EINVAL, "null argument");
	if (s->pending && (rc = finish_runs(s)) != RXGPU_OK)
		return rc;
	if (s->carry_lost)
		return rxgpu_fail(RXGPU_EINVAL, "the last sequence of runs failed while it was retired: its carries are unknown (set_carry and replay)");
	*c = s->carry;
	return RXGPU_OK;
}

long rxgpu_fm_stream_host_fixups(const rxgpu_fm_stream *s) { return s ? s->fixups : 0; }

/* where sample m of the demodulated stream sits in pcm[]: linear, or (chl2 != 0) the tiled layout of fm_kernels.hip's pcm_index */
static unsigned long long pcm_index_host(unsigned long long m, int chl2)
{
	if (!chl2)
		return m;
	const unsigned k = (unsigned)m & ((1u << chl2) - 1u), c = (unsigned)(m >> chl2) & 63u;
	return (m & ~((64ull << chl2) - 1)) | ((unsigned long long)(k >> 3) << 9) | (c << 3) | (k & 7u);
}

/* polar_discriminant with the host libm, exactly rtl_fm.c:470-483 */
static int polar_discriminant_host(int ar, int aj, int br, int bj)
{
	int cr = (int)((unsigned)ar * (unsigned)br + (unsigned)aj * (unsigned)bj);
	int cj = (int)((unsigned)aj * (unsigned)br - (unsigned)ar * (unsigned)bj);
	double angle = atan2((double)cj, (double)cr);
	return (int)(angle / 3.14159 * (1 << 14));
}

/* de-emphasis + resampler stages on stream st; pcm (M samples) -> d_out.  Re-runnable. */
static int run_audio_stages(rxgpu_fm_stream *s, hipStream_t st, unsigned long long M, unsigned long long J, int16_t *d_out)
{
	const rxgpu_fm_params *p = &s->p;
	const int resample = p->rate_out2 > 0;
	int16_t *deemph_dst = resample ? s->y : d_out;
	const int16_t *pcm = s->pcm;
	if (s->row_audio && M) {
		/* one short block (the drop-in's) whose carries the host knows: both stages in one workgroup with the row in LDS, not the tree's six launches */
		const int avg = s->carry.deemph_avg;
		const int serial = p->deemph && (p->deemph_a < 2 || p->deemph_a > 64 || avg < -32768 || avg > 32767);
		const int warm = p->deemph && !serial ? rxgpu_deemph_warm64(p->deemph_a) : 8;
		rxgpu_prof_begin_on("fm_deemph", st);
		RX_K(rxk_fm_row_audio(st, pcm, NULL, (unsigned)M, p->deemph, p->deemph_a, warm, serial, p->rate_out, resample ? p->rate_out2 : 0, (unsigned)J,
		                      &s->dev->in_deemph_avg, &s->dev->out_deemph_avg, d_out, &s->dev->out_deemph_avg, NULL, NULL, 0));
		rxgpu_prof_end_on("fm_deemph", st);
		return RXGPU_OK;
	}
	if (s->tiled && M) {
		/* the small-decimation chain: lane-per-chunk kernels on the tiled stream, de-emphasis and low_pass_real in two
		 * passes over pcm[] with the tree on compact tables in between; the filtered audio itself never reaches HBM */
		const int g = s->group, chl2 = s->tiled;
		unsigned long long cnt[DEEMPH_LEVELS + 1], off[DEEMPH_LEVELS + 1];
		int top = 0;
		const unsigned long long n_chunks = (M + (1ull << chl2) - 1) >> chl2;
		rxgpu_prof_begin_on("fm_deemph", st);
		RX_K(rxk_fm_deemph_scan_t(st, pcm, M, p->deemph_a, g, chl2, s->warm, s->lo0, s->gap_w, s->ctab, s->dev));
		cnt[0] = (n_chunks + RXK_DEEMPH_FAN - 1) / RXK_DEEMPH_FAN;
		off[0] = 0;
		RX_K(rxk_fm_deemph_up0(st, n_chunks, g, s->ctab, s->lvl_tab, s->lvl_lo, s->lvl_gap));
		const unsigned long long topcap = s->topcap_override ? (unsigned long long)s->topcap_override : (unsigned long long)DEEMPH_TOPCAP(g);
		while (cnt[top] > topcap) {
			if (top == DEEMPH_LEVELS)
				return rxgpu_fail(RXGPU_EUNSUPPORTED, "de-emphasis scan deeper than %d levels", DEEMPH_LEVELS);
			cnt[top + 1] = (cnt[top] + RXK_DEEMPH_FAN - 1) / RXK_DEEMPH_FAN;
			off[top + 1] = off[top] + cnt[top];
			RX_K(rxk_fm_deemph_up(st, cnt[top], g, s->lvl_tab + off[top] * g, s->lvl_lo + off[top], s->lvl_gap + off[top],
			                      s->lvl_tab + off[top + 1] * g, s->lvl_lo + off[top + 1], s->lvl_gap + off[top + 1]));
			top++;
		}
		RX_K(rxk_fm_deemph_top(st, (int)cnt[top], g, s->lvl_tab + off[top] * g, s->lvl_lo + off[top], s->lvl_gap + off[top],
		                       s->lvl_start + off[top], s->dev));
		for (int l = top; l > 0; l--)
			RX_K(rxk_fm_deemph_down(st, cnt[l - 1], g, s->lvl_tab + off[l - 1] * g, s->lvl_lo + off[l - 1],
			                        s->lvl_start + off[l], s->lvl_start + off[l - 1]));
		RX_K(rxk_fm_deemph_down0(st, n_chunks, s->ctab, s->lvl_start, s->cstart));
		rxgpu_prof_end_on("fm_deemph", st);
		rxgpu_prof_begin_on("fm_resample", st);
		RX_K(rxk_fm_deemph_apply_rs_t(st, pcm, M, p->deemph_a, chl2, s->cstart, p->rate_out, p->rate_out2, d_out, s->dev));
		rxgpu_prof_end_on("fm_resample", st);
		return RXGPU_OK;
	}
	if (p->post_downsample > 1) {
		/* rtl_fm.c:814-815; run_geometry made sure every block's length is a multiple of the step */
		M /= (unsigned long long)p->post_downsample;
		RX_K(rxk_fm_post_downsample(st, s->pcm, M, p->post_downsample, s->pcm_post));
		pcm = s->pcm_post;
	}
	const int16_t *audio = pcm;
	if (p->deemph && M) {
		rxgpu_prof_begin_on("fm_deemph", st);
		if (s->group) {
			/* tree scan over chunk maps: level 0 = composites of RXK_DEEMPH_WG_CHUNKS chunk tables (made by the scan
			 * kernel itself), level l+1 = composites of RXK_DEEMPH_FAN level-l tables */
			const int g = s->group;
			unsigned long long cnt[DEEMPH_LEVELS + 1], off[DEEMPH_LEVELS + 1];
			int top = 0;
			const unsigned long long n_chunks = (M + s->chunk - 1) / s->chunk;
			cnt[0] = (n_chunks + RXK_DEEMPH_WG_CHUNKS - 1) / RXK_DEEMPH_WG_CHUNKS;
			off[0] = 0;
			RX_K(rxk_fm_deemph_scan(st, pcm, M, p->deemph_a, g, s->chunk, s->warm, s->lo0, s->hi0,
			                        s->chunk_pre, s->lvl_tab, s->lvl_lo, s->lvl_gap, s->dev));
			const unsigned long long topcap = s->topcap_override ? (unsigned long long)s->topcap_override : (unsigned long long)DEEMPH_TOPCAP(g);
			while (cnt[top] > topcap) {
				if (top == DEEMPH_LEVELS)
					return rxgpu_fail(RXGPU_EUNSUPPORTED, "de-emphasis scan deeper than %d levels", DEEMPH_LEVELS);
				cnt[top + 1] = (cnt[top] + RXK_DEEMPH_FAN - 1) / RXK_DEEMPH_FAN;
				off[top + 1] = off[top] + cnt[top];
				RX_K(rxk_fm_deemph_up(st, cnt[top], g, s->lvl_tab + off[top] * g, s->lvl_lo + off[top], s->lvl_gap + off[top],
				                      s->lvl_tab + off[top + 1] * g, s->lvl_lo + off[top + 1], s->lvl_gap + off[top + 1]));
				top++;
			}
			RX_K(rxk_fm_deemph_top(st, (int)cnt[top], g, s->lvl_tab + off[top] * g, s->lvl_lo + off[top], s->lvl_gap + off[top],
			                       s->lvl_start + off[top], s->dev));
			for (int l = top; l > 0; l--)
				RX_K(rxk_fm_deemph_down(st, cnt[l - 1], g, s->lvl_tab + off[l - 1] * g, s->lvl_lo + off[l - 1],
				                        s->lvl_start + off[l], s->lvl_start + off[l - 1]));
			RX_K(rxk_fm_deemph_apply(st, pcm, M, p->deemph_a, g, s->chunk, s->chunk_pre, s->lvl_lo, s->lvl_start, deemph_dst));
		} else {
			RX_K(rxk_fm_deemph_serial(st, pcm, M, p->deemph_a, deemph_dst, s->dev));
		}
		rxgpu_prof_end_on("fm_deemph", st);
		audio = deemph_dst;
	}
	if (p->dc_block_audio && M) {
		/* rtl_fm.c:818: in place on whatever holds the audio now */
		int16_t *dst = (int16_t *)audio;
		if (audio == pcm) {
			/* pcm[] stays pristine (a host fix-up may have to redo these stages) */
			dst = resample ? s->y : d_out;
			RX_HIP(hipMemcpyAsync(dst, pcm, M * 2, hipMemcpyDeviceToDevice, st));
		}
		rxk_fm_blocks blk = s->blk;
		blk.post = p->post_downsample;
		RX_K(rxk_fm_dc_block(st, dst, M, blk, p->adc_block_const, s->dc_sums, s->dc_avgs, s->dev));
		audio = dst;
	}
	if (resample) {
		rxgpu_prof_begin_on("fm_resample", st);
		RX_K(rxk_fm_resample(st, audio, M, p->rate_out, p->rate_out2, J, d_out, s->dev));
		rxgpu_prof_end_on("fm_resample", st);
	} else if (audio == pcm && M) {
		/* no stage wrote d_out yet: the demodulator output is the result */
		RX_HIP(hipMemcpyAsync(d_out, pcm, M * 2, hipMemcpyDefault, st));    /* d_out may be a page-locked host mirror (the drop-in) */
	}
	if (!(p->deemph && M) || !resample)
		RX_K(rxk_fm_passthrough_carry(st, s->dev, !(p->deemph && M), !resample));
	return RXGPU_OK;
}

static int run_geometry(rxgpu_fm_stream *s, size_t n_blocks, size_t block_len, size_t out_cap, struct run_geom *g)
{
	const rxgpu_fm_params *p = &s->p;
	g->n = block_len / 2;                                    /* complex samples per block */
	g->T = g->n * n_blocks;
	if (g->T > s->max_T)
		return rxgpu_fail(RXGPU_ECAPACITY, "stream created for %zu samples, run asks %llu", s->max_T, g->T);
	g->passes = p->downsample_passes;
	g->ds = g->passes ? 1 : p->downsample;
	g->p0 = g->passes ? 0 : s->h_prev_index;
	g->pr0 = s->h_prev_lpr_index;
	g->rotate = !p->prescaled && !p->offset_tuning && !p->dc_block_raw;   /* the -E rdc pre-pass rotates */
	g->K = 0;
	g->rdc_fused = 0;
	g->fast = 0;
	g->literal = 0;
	g->lf = 0;
	if (g->passes) {
		if ((g->n % (1ull << g->passes)) || (g->n >> (g->passes - 1)) < 16) {
			/* not whole tiles of the cascade: rtl_fm.c:764-769 taken literally on each block's int16 array (lp_len >> i may be odd) */
			g->literal = 1;
			g->lf = (int)((2 * g->n) >> g->passes);
			g->K = (unsigned long long)(g->lf / 2);          /* result_len = lp_len / 2, rtl_fm.c:614 */
			if (g->lf < 2 && !s->allow_empty)
				return rxgpu_fail(RXGPU_EUNSUPPORTED, "a block of %llu samples leaves %d int16 after %d fifth_order passes: the reference then takes "
				                  "pre_r/pre_j from in front of lowpassed[] (rtl_fm.c:612-613), which only the drop-in on the real struct can reproduce",
				                  g->n, g->lf, g->passes);
		} else {
			g->K = g->n >> g->passes;
		}
		g->M = g->K * n_blocks;
	} else {
		if (g->p0 < 0 || g->p0 >= g->ds)
			return rxgpu_fail(RXGPU_EINVAL, "prev_index %d outside [0,%d)", g->p0, g->ds);
		g->M = ((unsigned long long)g->p0 + g->T) / (unsigned long long)g->ds;
		if (g->n < (unsigned long long)g->ds && (n_blocks > 1 || (!g->M && !s->allow_empty)))
			return rxgpu_fail(RXGPU_EUNSUPPORTED, "blocks of %llu samples are shorter than downsample %d: a block that completes no window makes the "
			                  "reference's fm_demod read pre_r/pre_j from in front of lowpassed[] (rtl_fm.c:612-613); only the drop-in, block by block "
			                  "on the real struct, reproduces that", g->n, g->ds);
		g->fast = g->ds >= 4 && g->ds <= RXK_DEC_MAX_DS && (g->n % 4) == 0 && g->n >= (unsigned long long)g->ds;
		/* -E rdc in front of the span decimator, blocks of whole spans: the averages are subtracted there (rxk_fm_decimate_rdc) */
		g->rdc_fused = p->dc_block_raw && !p->prescaled && g->fast && g->ds > RXK_DEC_SMALL_MAX && g->n % RXK_DEC_SPAN == 0;
		if (g->rdc_fused)
			g->rotate = !p->offset_tuning;
	}
	if (g->M > s->max_M)
		return rxgpu_fail(RXGPU_ECAPACITY, "workspace too small for %llu decimated samples", g->M);
	if (!g->M && !s->allow_empty)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "run produces no decimated sample");
	if (!g->M && p->dc_block_audio && p->mode != RXGPU_MODE_RAW)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "-E adc on a block without a demodulated sample: the reference divides by result_len == 0 (rtl_fm.c:693)");
	g->J = g->M;
	g->post = (p->post_downsample > 1 && p->mode != RXGPU_MODE_RAW) ? p->post_downsample : 1;
	if (g->post > 1) {
		/* low_pass_simple (rtl_fm.c:373-387) on a length that is no multiple of the step: its last loop turn sums past `len` -- into what
		 * earlier blocks left in result[] -- but writes that sum to signal2[len / step], BEHIND the len / step values it returns: what a
		 * block hands on is its complete groups, the remainder is dropped, not carried.  One block per run (the drop-in's shape) is served
		 * that way; a multi-block run keeps the condition, because its blocks are cut out of ONE demodulated sequence by cumulative counts. */
		const unsigned long long per = g->passes ? g->K : g->n / (unsigned long long)g->ds;
		if (n_blocks > 1 && ((!g->passes && g->n % (unsigned long long)g->ds) || per % (unsigned long long)g->post))
			return rxgpu_fail(RXGPU_EUNSUPPORTED, "-o %d on a run of several blocks needs every block's demodulated length to be a multiple of it "
			                  "(block of %llu samples, downsample %d); block by block (the drop-in) any length goes", g->post, g->n, g->ds);
		g->J = g->M / (unsigned long long)g->post;
	}
	if (p->mode == RXGPU_MODE_RAW) {
		/* raw_demod: result = lowpassed, rtl_fm.c:658-665, 809-811 (an odd lp_len included on the literal path) */
		g->J = g->literal ? (unsigned long long)g->lf * n_blocks : 2 * g->M;
	} else if (p->rate_out2 > 0) {
		if (g->pr0 < 0 || g->pr0 >= p->rate_out)
			return rxgpu_fail(RXGPU_EINVAL, "prev_lpr_index %d outside [0,rate_out)", g->pr0);
		g->J = ((unsigned long long)g->pr0 + g->J * (unsigned long long)p->rate_out2) / (unsigned long long)p->rate_out;
	}
	if (g->J > out_cap)
		return rxgpu_fail(RXGPU_ECAPACITY, "output needs %llu int16, capacity %zu", g->J, out_cap);
	return RXGPU_OK;
}

static void block_lengths(const rxgpu_fm_stream *s, const struct run_geom *g, size_t n_blocks, int *block_out_len)
{
	const rxgpu_fm_params *p = &s->p;
	unsigned long long j_prev = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		unsigned long long cum = g->passes ? g->K * (b + 1) : ((unsigned long long)g->p0 + g->n * (b + 1)) / (unsigned long long)g->ds;
		if (p->mode != RXGPU_MODE_RAW)
			cum /= (unsigned long long)g->post;
		unsigned long long jj = p->mode == RXGPU_MODE_RAW ? (g->literal ? (unsigned long long)g->lf * (b + 1) : 2 * cum)
			: p->rate_out2 > 0 ? ((unsigned long long)g->pr0 + cum * (unsigned long long)p->rate_out2) / (unsigned long long)p->rate_out : cum;
		block_out_len[b] = (int)(jj - j_prev);
		j_prev = jj;
	}
}

/* Enqueue one run.  The HBM-bound decimator goes on stream A, everything after it (1/ds of the
 * data, latency-bound) on stream B behind an event, so that the next run's decimator overlaps
 * this run's audio stages.  Carries stay on the device between chained runs. */
static int retire_slot(rxgpu_fm_stream *s, int slot);
static hipStream_t fm_sa(const rxgpu_fm_stream *s) { return s->one_stream ? rxgpu_hip_stream2() : rxgpu_hip_stream(); }
static hipStream_t fm_sd(const rxgpu_fm_stream *s) { return s->one_stream ? rxgpu_hip_stream2() : rxgpu_hip_stream4(); }

static int enqueue_run(rxgpu_fm_stream *s, const int16_t *d_iq_in, size_t n_blocks, size_t block_len,
                       int16_t *d_out, const struct run_geom *g)
{
	hipStream_t sa = fm_sa(s), sb = rxgpu_hip_stream2();
	const rxgpu_fm_params *p = &s->p;
	const int16_t *d_iq = d_iq_in;
	int prescaled = p->prescaled;
	int rc;
	/* buffer set / flag slot of this run; the run that used it two enqueues ago is retired first (normally long
	 * finished: its discriminator ran before the previous run's decimator even started) */
	const int db = (int)(s->seq & 1);
	if (s->rec[db].live && (rc = retire_slot(s, db)) != RXGPU_OK)
		return rc;
	if (p->dc_block_raw) {
		/* -E rdc: what the callback does before it hands lowpassed[] over (rtl_fm.c:845-857): scale, dc_block_raw_filter,
		 * rotate16_90 -- written out once, the chain then runs on it as prescaled input */
		RX_HIP(hipMemcpyAsync(s->rdc_state, &s->carry.dc_avgI, 8, hipMemcpyHostToDevice, sa));
		RX_K(rxk_fm_rdc(sa, d_iq_in, n_blocks, g->n, 0, !p->offset_tuning, p->rdc_block_const, s->rdc_state, s->rdc_sums,
		                s->rdc_avg, g->rdc_fused ? NULL : s->rdc_buf));
		RX_HIP(hipEventRecord(s->ev_rdc, sa));
		RX_HIP(hipStreamWaitEvent(sb, s->ev_rdc, 0));
		if (!g->rdc_fused) {
			d_iq = s->rdc_buf;
			prescaled = 1;
		}
	}
	rxk_fm_dev *h = s->dev_host;
	if (p->deemph)
		deemph_geometry(s);
	RX_HIP(hipMemsetAsync(s->flag_cnt_dev + db, 0, sizeof(int), sb));
	/* -F on the raw capture: the first fused group of fifth_order passes reads 8/9 of all the bytes of the run; like the
	 * boxcar decimator it goes on stream A, so that it overlaps the later passes and the audio stages of the run before */
	/* four passes in that group where the cascade has them (the next group then reads 1/16 of the capture, not 1/8; five bought nothing more) */
	const int fuse_a_max = 4;
	const int fuse_a = (g->passes && !g->literal && !prescaled && (g->n % RXK_FIFTH_TILE) == 0 && s->cas_a[0]) ? (g->passes < fuse_a_max ? g->passes : fuse_a_max) : 0;
	/* a three-pass cascade is the whole of the group: the droop FIR and the discriminator ride in the same launch and only pcm leaves it
	 * (squelch, -A std / lut, am / usb / lsb / raw and other FIR sizes take the separate kernels) */
	const int fuse_dd = fuse_a == 3 && g->passes == 3 && p->mode == RXGPU_MODE_FM && !p->squelch_level && p->custom_atan == 1 &&
	                    (p->comp_fir_size == 9 || p->comp_fir_size == 0);
	const int fresh = !s->chained;

	if (!s->chained) {
		/* carries in from the host copy, status cleared */
		memset(h, 0, sizeof(*h));
		h->in_now_r = s->carry.now_r; h->in_now_j = s->carry.now_j; h->in_prev_index = s->carry.prev_index;
		h->in_pre_r = s->carry.pre_r; h->in_pre_j = s->carry.pre_j;
		h->in_deemph_avg = s->carry.deemph_avg;
		h->in_now_lpr = s->carry.now_lpr; h->in_prev_lpr_index = s->carry.prev_lpr_index;
		h->in_dc_avg = s->carry.dc_avg;
		h->out_now_r = h->in_now_r; h->out_now_j = h->in_now_j; h->out_prev_index = h->in_prev_index;
		h->out_pre_r = h->in_pre_r; h->out_pre_j = h->in_pre_j; h->out_dc_avg = h->in_dc_avg;
		RX_HIP(hipMemcpyAsync(s->dev, h, sizeof(*h), hipMemcpyHostToDevice, sb));
		if (g->passes) {
			int16_t *hh = s->hist_host;
			for (int i = 0; i < 10; i++) {
				memcpy(hh + HIST_CAS_IN + i * 12, s->carry.lp_i_hist[i], 12);
				memcpy(hh + HIST_CAS_IN + i * 12 + 6, s->carry.lp_q_hist[i], 12);
			}
			memcpy(hh + HIST_DROOP_IN, s->carry.droop_i_hist, 18);
			memcpy(hh + HIST_DROOP_IN + 9, s->carry.droop_q_hist, 18);
			RX_HIP(hipMemcpyAsync(s->hist_dev, hh, HIST_TOTAL * 2, hipMemcpyHostToDevice, sb));
		}
		RX_K(rxk_fm_carry_advance(sb, s->dev, 0, s->snap_dev + 4 * db));
	} else {
		/* carries out of the previous run become this run's carries in, on the device */
		RX_K(rxk_fm_carry_advance(sb, s->dev, 1, s->snap_dev + 4 * db));
		if (g->passes) {
			/* the histories of the passes stream A runs are advanced there (below), the others here */
			RX_K(rxk_copy_small(sb, s->hist_dev + HIST_CAS_IN + fuse_a * 12, s->hist_dev + HIST_CAS_OUT + fuse_a * 12, (unsigned)(10 - fuse_a) * 12 * 2));
			if (p->comp_fir_size == 9 && !fuse_dd)
				RX_K(rxk_copy_small(sb, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT, 18 * 2));
		}
	}

	s->lp_final = s->lp;
	s->tiled = 0;
	s->pcm = s->pcm_buf[db];                 /* stays intact until the run is retired: a host fix-up patches it */
	rxk_flag_rec *const flag_rec = s->flag_rec_dev + (size_t)db * RXK_FLAG_CAP;
	int *const flag_cnt = s->flag_cnt_dev + db;
	s->blk.first_mode = g->passes ? RXK_FIRST_UNIFORM : RXK_FIRST_LOWPASS;
	s->blk.ds = g->ds; s->blk.p0 = g->p0; s->blk.n = g->n; s->blk.k = g->K; s->blk.n_blocks = n_blocks;
	s->last_n_blocks = n_blocks;
	/* squelch and the non-fm demodulators need the finished lowpassed[] before anything is demodulated */
	const int split = p->squelch_level != 0 || p->mode != RXGPU_MODE_FM;
	int lit_done = 0;                        /* the literal per-block path did squelch and demodulation itself */
	/* de-emphasis + resampler behind an fm discriminator: hand them the demodulated samples in the tiled layout their
	 * lane-per-chunk kernels stream (rxk_fm_deemph_scan_t / _apply_rs_t).  The LDS-staged kernels keep: no resampler behind the de-emphasis,
	 * even a or a < 9 or a > 255 (rxk_fm_deemph_tiled_ok), -o, -E adc, squelch and the non-fm demodulators in front. */
	s->tiled = 0;
	s->row_audio = s->one_stream && fresh && g->M <= RXK_ROW_AUDIO_MAX && g->post == 1 && !p->dc_block_audio && (p->deemph || p->rate_out2 > 0);
	if (!split && !g->literal && p->deemph && s->group && p->rate_out2 > 0 && g->post == 1 && !p->dc_block_audio && !s->row_audio)
	{
		/* $RXGPU_DEEMPH_CHUNK=256 forces the larger chunk where the warm-up would fit 128 (tests of that template; measured: no
		 * gain -- the scan's warm-up weighs less, but the resampler's per-wave staging doubles and with it the LDS a workgroup needs
		 * to find beside the decimator's) */
		const int chunk = (s->k_deemph_chunk == 256 && s->chunk < 256) ? 256 : s->chunk;
		s->tiled = rxk_fm_deemph_tiled_ok(p->deemph_a, s->group, chunk, p->rate_out, p->rate_out2);
	}
	if (!g->passes) {
		const int fused_disc = g->fast && p->custom_atan == 1 && !split;
		/* lowpassed[] is an intermediate of the fused chain: keep only the entries the seam kernel reads
		 * (the drop-in, which must hand lowpassed[] back, runs prescaled) */
		const int lp_sparse = fused_disc && !prescaled && g->ds <= RXK_LP_SPARSE_MAX_DS;
		/* small decimation on the raw capture: the direct kernel (samples staged in LDS, a thread per output, no seams) */
		const int small = g->fast && fused_disc && !prescaled && g->ds >= RXK_DEC_SMALL_MIN && g->ds <= RXK_DEC_SMALL_MAX;
		if (g->fast) {
			s->lp_final = s->lp_raw[db];
			/* buffer set `db` was last read by the audio chain two runs ago */
			if (s->ev_small_valid[db])
				RX_HIP(hipStreamWaitEvent(sa, s->ev_small[db], 0));
			rxgpu_prof_begin_on("fm_decimate", sa);
			if (small)
				RX_K(rxk_fm_decimate_small(sa, d_iq, g->T, g->ds, g->p0, g->rotate, g->M, s->pcm, s->tiled));
			else if (g->rdc_fused)
				RX_K(rxk_fm_decimate_rdc(sa, d_iq, g->T, g->ds, g->p0, g->rotate, s->lp_raw[db], s->head[db], s->tail[db],
				                         lp_sparse, fused_disc ? s->pcm : NULL, s->tiled, s->rdc_avg, (unsigned)(g->n / RXK_DEC_SPAN)));
			else
				RX_K(rxk_fm_decimate(sa, d_iq, g->T, g->ds, g->p0, prescaled, g->rotate, s->lp_raw[db], s->head[db], s->tail[db],
				                     lp_sparse, fused_disc ? s->pcm : NULL, s->tiled));
			rxgpu_prof_end_on("fm_decimate", sa);
			RX_HIP(hipEventRecord(s->ev_dec[db], sa));
			RX_HIP(hipStreamWaitEvent(sb, s->ev_dec[db], 0));
		} else if (g->M) {
			rxgpu_prof_begin_on("fm_decimate_generic", sb);
			RX_K(rxk_fm_decimate_generic(sb, d_iq, g->T, g->ds, g->p0, g->n, prescaled, g->rotate, s->dev, s->lp, g->M));
			rxgpu_prof_end_on("fm_decimate_generic", sb);
		}
		rxgpu_prof_begin_on("fm_disc", sb);
		/* fast path: lp_raw is finished in place (only seam entries change) and becomes the final decimated IQ */
		if (!g->fast && g->M && !split && p->custom_atan == 1 && g->p0 == 0 && g->n % (unsigned long long)g->ds == 0 &&
		    g->M == n_blocks * (g->n / (unsigned long long)g->ds) && ((size_t)s->lp & 15u) == 0)
			/* behind the one-thread-per-output decimator (ds < 4), whole windows per block: every block's first output is every (n / ds)-th one, no
			 * carry is left over -- the four-outputs-per-thread discriminator (no filter) instead of the dense kernel */
			RX_K(rxk_fm_droop_disc(sb, s->lp, g->M, NULL, NULL, NULL, NULL, g->n / (unsigned long long)g->ds, s->pcm, s->tiled, s->dev, flag_rec, flag_cnt,
			                       s->flag_all));
		else if (g->rdc_fused)
			RX_K(rxk_fm_disc_rdc(sb, d_iq, g->T, g->ds, g->p0, g->n, g->rotate, g->fast, s->lp_raw[db], s->head[db], s->tail[db], s->lp_raw[db], g->M,
			                     p->custom_atan, split ? NULL : s->pcm, s->dev, flag_rec, flag_cnt, fused_disc, n_blocks, s->atan_lut, lp_sparse, s->flag_all,
			                     s->tiled, s->rdc_avg));
		else
		RX_K(rxk_fm_disc(sb, d_iq, g->T, g->ds, g->p0, g->n, prescaled, g->rotate, small ? 2 : g->fast, g->fast ? s->lp_raw[db] : s->lp,
		                 s->head[db], s->tail[db], g->fast ? s->lp_raw[db] : s->lp, g->M, RXK_FIRST_LOWPASS, 0, p->custom_atan, 1,
		                 split ? NULL : s->pcm, s->dev, flag_rec, flag_cnt, fused_disc, n_blocks, s->atan_lut, g->fast ? lp_sparse : 0, s->flag_all, s->tiled));
		rxgpu_prof_end_on("fm_disc", sb);
	} else if (g->literal) {
		/* -F on blocks that are not whole tiles: full_demod's cascade, droop FIR, squelch and demodulator block after block on
		 * the block's own int16 array, exactly as the C loops index it (rtl_fm.c:764-776, 781-790, 808); everything stays on the
		 * device, histories and pre_r/pre_j chained through hist_dev / dev.  Rare shapes: one small launch per pass and block. */
		const int passes = g->passes;
		lit_done = 1;
		if (p->comp_fir_size == 9 && s->fir_loaded != passes) {
			RX_HIP(hipMemcpyAsync(s->fir_dev, cic_9_tables[passes], 10 * sizeof(int), hipMemcpyHostToDevice, sb));
			s->fir_loaded = passes;
		}
		const unsigned long long per_out = p->mode == RXGPU_MODE_RAW ? (unsigned long long)g->lf : g->K;
		for (size_t b = 0; b < n_blocks; b++) {
			const int16_t *cur = d_iq + b * block_len;
			int w = 0;
			if (!prescaled) {
				RX_K(rxk_fm_prestage(sb, cur, (unsigned)g->n, g->rotate, s->lit[0]));
				cur = s->lit[0];
				w = 1;
			}
			for (int i = 0; i < passes; i++) {
				RX_K(rxk_fm_fifth_lit(sb, cur, s->lit[w], (int)((2 * g->n) >> i), s->hist_dev + HIST_CAS_IN + i * 12, s->hist_dev + HIST_CAS_OUT + i * 12));
				cur = s->lit[w];
				w ^= 1;
			}
			RX_K(rxk_copy_small(sb, s->hist_dev + HIST_CAS_IN, s->hist_dev + HIST_CAS_OUT, (unsigned)passes * 12 * 2));
			if (p->comp_fir_size == 9) {
				RX_K(rxk_fm_droop_lit(sb, cur, s->lit[w], g->lf, s->fir_dev, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT));
				RX_K(rxk_copy_small(sb, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT, 18 * 2));
				cur = s->lit[w];
				w ^= 1;
			}
			if (p->squelch_level && g->lf > 0)
				RX_K(rxk_fm_squelch_lit(sb, (int16_t *)cur, g->lf, p->squelch_level, s->below + b, s->below + s->max_blocks + 1 + b));
			else if (p->squelch_level) {
				RX_HIP(hipMemsetAsync(s->below + b, 1, sizeof(int), sb));   /* rms() of nothing is (int)NaN = INT_MIN on x86-64: below any level */
				RX_HIP(hipMemsetAsync(s->below + s->max_blocks + 1 + b, 0x80, sizeof(int), sb));   /* 0x80808080: negative, stands for that INT_MIN (block_rms) */
			}
			RX_K(rxk_fm_demod_lit(sb, cur, g->lf, p->mode, p->custom_atan, p->output_scale, p->mode == RXGPU_MODE_RAW ? d_out : s->pcm,
			                      (unsigned long long)b * per_out, s->dev, b > 0, flag_rec, flag_cnt, s->atan_lut, s->flag_all));
			s->lp_final = (const uint32_t *)cur;             /* the drop-in hands the (single) block's lowpassed[] back */
		}
		if (p->mode == RXGPU_MODE_RAW)
			RX_K(rxk_fm_passthrough_carry(sb, s->dev, 1, 1));
	} else {
		/* F3: cascade (first passes fused where the input is raw), F12 optional */
		const int passes = g->passes;
		int fused_dd = 0;
		const void *src = d_iq;
		unsigned n_in = (unsigned)g->n, in_stride = (unsigned)g->n;
		int first_pass = 0;
		if (fuse_a) {
			const int fuse = fuse_a;
			uint32_t *dst = s->cas_a[db];
			/* The group's seam histories need nothing from the previous run but its archive (a few samples, written by ITS seam
			 * kernel): they go on a stream of their own and are ready long before stream A gets to this run -- left on stream A,
			 * the history copy and the seam kernel sat between two HBM-bound launches (50-150 us of an idle chip per run). */
			hipStream_t sd = fm_sd(s);
			if (fresh) {
				RX_HIP(hipEventRecord(s->ev_up, sb));    /* the history upload above went through stream B */
				RX_HIP(hipStreamWaitEvent(sd, s->ev_up, 0));
			} else {
				/* histories this run advances on the seam stream were left there by the previous run's seam kernels -- unless that run
				 * had another shape (a ragged run between whole-tile runs, a shorter first group): then stream B wrote some of them, at
				 * the end of its chain */
				if ((s->last_fuse_a != fuse_a || s->last_fuse_dd != fuse_dd) && s->ev_small_valid[db ^ 1])
					RX_HIP(hipStreamWaitEvent(sd, s->ev_small[db ^ 1], 0));
				RX_K(rxk_copy_small(sd, s->hist_dev + HIST_CAS_IN, s->hist_dev + HIST_CAS_OUT, (unsigned)fuse * 12 * 2));
				if (fuse_dd && p->comp_fir_size == 9)
					RX_K(rxk_copy_small(sd, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT, 18 * 2));
			}
			RX_K(rxk_fm_fifth_seams(sd, d_iq, 0, g->rotate, n_blocks, (unsigned)g->n, fuse, s->hist_dev + HIST_CAS_IN,
			                        s->hist_dev + HIST_CAS_OUT, s->seams_a[db]));
			uint32_t *const tails = s->seams_a[db] + (s->max_blocks + 1) * 15;
			if (fuse_dd)
				RX_K(rxk_fm_fifth_tails(sd, d_iq, g->rotate, n_blocks, (unsigned)g->n, fuse, p->comp_fir_size == 9 ? s->hist_dev + HIST_DROOP_IN : NULL,
				                        p->comp_fir_size == 9 ? s->hist_dev + HIST_DROOP_OUT : NULL, tails));
			RX_HIP(hipEventRecord(s->ev_seam[db], sd));
			RX_HIP(hipStreamWaitEvent(sa, s->ev_seam[db], 0));
			if (s->ev_small_valid[db])                   /* cas_a[db] was last read by the run two enqueues ago */
				RX_HIP(hipStreamWaitEvent(sa, s->ev_small[db], 0));
			rxgpu_prof_begin_on("fm_fifth", sa);
			if (fuse_dd)
				RX_K(rxk_fm_fifth_dd(sa, d_iq, g->rotate, n_blocks, (unsigned)g->n, fuse, s->seams_a[db], tails,
				                     p->comp_fir_size == 9 ? cic_9_tables[passes] : NULL, s->pcm, s->tiled, s->edges_a[db]));
			else
				RX_K(rxk_fm_fifth_fused(sa, d_iq, 0, g->rotate, n_blocks, (unsigned)g->n, fuse, NULL,
				                        s->hist_dev + HIST_CAS_OUT, s->seams_a[db], dst));
			rxgpu_prof_end_on("fm_fifth", sa);
			RX_HIP(hipEventRecord(s->ev_dec[db], sa));
			RX_HIP(hipStreamWaitEvent(sb, s->ev_dec[db], 0));
			if (fuse_dd) {
				/* what needs the carries and the flag list: each block's first sample, pre_r/pre_j out */
				rxgpu_prof_begin_on("fm_disc", sb);
				RX_K(rxk_fm_dd_edges(sb, s->edges_a[db], n_blocks, g->K, s->pcm, s->tiled, s->dev, flag_rec, flag_cnt, s->flag_all));
				rxgpu_prof_end_on("fm_disc", sb);
				fused_dd = 1;
			}
			src = dst;
			n_in = (unsigned)(g->n >> fuse);
			in_stride = n_in;
			first_pass = fuse;
		}
		rxgpu_prof_begin_on("fm_fifth2", sb);
		/* further fused groups on the decimated stream while whole tiles are left (3 passes, or 1 so that the ping-pong buffers
		 * stay distinct): passes 4-6 at 1/8 rate, the 7th of ds = 128 at 1/64 -- the one-thread-per-output kernel below took as
		 * long for that last pass as the three before it */
		for (int grp = 1; fuse_a >= 3 && grp < 4 && first_pass < passes && n_in >= RXK_FIFTH_TILE && (n_in % RXK_FIFTH_TILE) == 0; grp++) {
			const int fuse2 = passes - first_pass >= 3 ? 3 : 1;
			if (first_pass + fuse2 > 7)                      /* the kernel's 32-bit sums assume inputs below 2^14: 128 * 2^6 at most */
				break;
			uint32_t *dst2 = s->cas[(first_pass + fuse2 - 1) & 1];
			RX_K(rxk_fm_fifth_fused(sb, src, 1, 0, n_blocks, n_in, fuse2, s->hist_dev + HIST_CAS_IN + first_pass * 12,
			                        s->hist_dev + HIST_CAS_OUT + first_pass * 12, s->seams + (size_t)grp * (s->max_blocks + 1) * 15, dst2));
			src = dst2;
			n_in >>= fuse2;
			in_stride = n_in;
			first_pass += fuse2;
		}
		for (int i = first_pass; i < passes; i++) {
			uint32_t *dst = s->cas[i & 1];
			unsigned n_out = n_in / 2;
			RX_K(rxk_fm_fifth_pass(sb, src, i == 0, prescaled, g->rotate, n_blocks, n_in, in_stride, dst, n_out,
			                       s->hist_dev + HIST_CAS_IN + i * 12, s->hist_dev + HIST_CAS_OUT + i * 12));
			src = dst;
			n_in = n_out;
			in_stride = n_out;
		}
		rxgpu_prof_end_on("fm_fifth2", sb);
		s->lp_final = fuse_dd ? NULL : (const uint32_t *)src;   /* [n_blocks][K] contiguous == M samples */
		if (p->comp_fir_size == 9 && !fuse_dd) {
			if (s->fir_loaded != passes) {                   /* the table of this cascade depth: once */
				RX_HIP(hipMemcpyAsync(s->fir_dev, cic_9_tables[passes], 10 * sizeof(int), hipMemcpyHostToDevice, sb));
				s->fir_loaded = passes;
			}
			if (!split && p->custom_atan == 1) {
				/* the droop FIR and the discriminator in one pass; the FIR output itself is only kept where somebody reads it
				 * back (the drop-in hands lowpassed[] to its caller) */
				rxgpu_prof_begin_on("fm_droop", sb);
				RX_K(rxk_fm_droop_disc(sb, s->lp_final, g->M, s->fir_dev, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT,
				                       prescaled ? s->lp : NULL, g->K, s->pcm, s->tiled, s->dev, flag_rec, flag_cnt, s->flag_all));
				rxgpu_prof_end_on("fm_droop", sb);
				if (prescaled)
					s->lp_final = s->lp;
				fused_dd = 1;
			} else {
				rxgpu_prof_begin_on("fm_droop", sb);
				RX_K(rxk_fm_droop(sb, s->lp_final, g->M, s->fir_dev, s->hist_dev + HIST_DROOP_IN, s->hist_dev + HIST_DROOP_OUT, s->lp));
				rxgpu_prof_end_on("fm_droop", sb);
				s->lp_final = s->lp;
			}
		}
		if (!split && !fused_dd && p->custom_atan == 1 && p->comp_fir_size != 9 && ((size_t)s->lp_final & 15u) == 0) {
			/* -F without the droop FIR, -A fast: the four-outputs-per-thread discriminator (rxk_fm_droop_disc without a filter) */
			rxgpu_prof_begin_on("fm_disc", sb);
			RX_K(rxk_fm_droop_disc(sb, s->lp_final, g->M, NULL, NULL, NULL, NULL, g->K, s->pcm, s->tiled, s->dev, flag_rec, flag_cnt, s->flag_all));
			rxgpu_prof_end_on("fm_disc", sb);
			fused_dd = 1;
		}
		if (!split && !fused_dd) {
			rxgpu_prof_begin_on("fm_disc", sb);
			RX_K(rxk_fm_disc(sb, d_iq, g->T, 1, 0, g->n, prescaled, g->rotate, 0, s->lp_final, NULL, NULL, NULL, g->M,
			                 RXK_FIRST_UNIFORM, g->K, p->custom_atan, 0, s->pcm, s->dev, flag_rec, flag_cnt, 0, n_blocks, s->atan_lut, 0, s->flag_all, s->tiled));
			rxgpu_prof_end_on("fm_disc", sb);
		}
	}
	if (split && !lit_done) {
		uint32_t *lpw = (uint32_t *)s->lp_final;             /* every producer of lp_final owns it writable */
		if (p->squelch_level)
			RX_K(rxk_fm_squelch(sb, lpw, s->blk, p->squelch_level, s->below, s->below + s->max_blocks + 1));
		if (p->mode == RXGPU_MODE_FM) {
			rxgpu_prof_begin_on("fm_disc", sb);
			RX_K(rxk_fm_disc(sb, d_iq, g->T, g->ds, g->p0, g->n, prescaled, g->rotate, 0, lpw, NULL, NULL, NULL, g->M,
			                 s->blk.first_mode, g->K, p->custom_atan, 0, s->pcm, s->dev, flag_rec, flag_cnt, 0, n_blocks, s->atan_lut, 0, s->flag_all, 0));
			rxgpu_prof_end_on("fm_disc", sb);
		} else if (p->mode == RXGPU_MODE_RAW) {
			RX_HIP(hipMemcpyAsync(d_out, lpw, g->M * 4, hipMemcpyDefault, sb));
			RX_K(rxk_fm_passthrough_carry(sb, s->dev, 1, 1));
		} else {
			RX_K(rxk_fm_simple_demod(sb, lpw, g->M, p->mode, p->output_scale, s->pcm));
		}
	}
	/* every kernel that can flag a libm sample has been enqueued: the count comes back behind them, and the event
	 * tells the host when this run's demodulated samples (and its reads of d_iq) are complete */
	RX_K(rxk_copy_small(sb, s->flag_cnt_host + db, flag_cnt, sizeof(int)));       /* pinned host memory, written by the device */
	RX_HIP(hipEventRecord(s->ev_disc[db], sb));
	if (!g->M) {
		/* no demodulated sample (the drop-in on a tiny block): the audio stages see result_len == 0 and change nothing */
		rc = RXGPU_OK;
		if (p->mode != RXGPU_MODE_RAW)
			RX_K(rxk_fm_passthrough_carry(sb, s->dev, 1, 1));
	} else {
		rc = p->mode == RXGPU_MODE_RAW ? RXGPU_OK : run_audio_stages(s, sb, g->M, g->J, d_out);
	}
	if (rc != RXGPU_OK)
		return rc;
	if (p->squelch_level)
		RX_HIP(hipMemcpyAsync(s->below_host, s->below, (s->max_blocks + 1 + n_blocks) * 4, hipMemcpyDeviceToHost, sb));
	RX_HIP(hipEventRecord(s->ev_small[db], sb));
	s->ev_small_valid[db] = 1;
	/* what the next run needs from this one on the host side is closed-form */
	if (!g->passes)
		s->h_prev_index = (int)(((unsigned long long)g->p0 + g->T) - g->M * (unsigned long long)g->ds);
	if (p->rate_out2 > 0 && p->mode != RXGPU_MODE_RAW)
		s->h_prev_lpr_index = (int)((unsigned long long)g->pr0 + (g->M / (unsigned long long)g->post) * (unsigned long long)p->rate_out2 -
		                            g->J * (unsigned long long)p->rate_out);
	if (s->lp_mirror) {
		/* the drop-in hands lowpassed[] back: lp_len' int16 (odd on some -F shapes), never fewer than two -- every fifth_order pass rewrites
		 * lowpassed[0] and [1] even for an empty block (rtl_fm.c:419-423) */
		const int lp_len_out = g->passes ? (g->literal ? g->lf : (int)(2 * g->K)) : (int)(2 * g->M);
		const size_t back = g->passes ? (size_t)(lp_len_out > 2 ? lp_len_out : 2) : (size_t)lp_len_out;
		RX_K(rxk_copy_mirror(sb, s->lp_mirror, s->lp_final, (unsigned)(back * 2)));
	}
	s->chained = 1;
	s->last_fuse_a = fuse_a;
	s->last_fuse_dd = fuse_dd;
	s->pending++;
	s->last = *g;
	s->rec[db].live = 1;
	s->rec[db].seq = s->seq++;
	s->rec[db].g = *g;
	s->rec[db].d_out = d_out;
	s->rec[db].pcm = s->pcm;
	s->rec[db].tiled = s->tiled;
	s->rec[db].row_audio = s->row_audio;
	s->rec[db].blk = s->blk;
	s->rec[db].n_blocks = n_blocks;
	return RXGPU_OK;
}

/* Settle the libm samples the device left undecided in the run of `slot`, and in the run enqueued behind it (its
 * audio stages consumed carries that are about to change): re-evaluate them with the host libm -- the one the
 * reference uses -- patch pcm[], and redo the audio stages from the snapshot of their carries-in.  Everything else
 * of those runs (decimated IQ, discriminator carries, the other pcm samples) is final already. */
static int fixup_from(rxgpu_fm_stream *s, int slot)
{
	hipStream_t sa = fm_sa(s), sb = rxgpu_hip_stream2();
	int rc;
	RX_HIP(hipStreamSynchronize(sa));
	RX_HIP(hipStreamSynchronize(sb));
	const int other = slot ^ 1;
	int order[2], n = 0;
	order[n++] = slot;
	if (s->rec[other].live && s->rec[other].seq > s->rec[slot].seq)
		order[n++] = other;
	for (int i = 0; i < n; i++) {
		const int q = order[i];
		struct run_rec *r = &s->rec[q];
		const int cnt = s->flag_cnt_host[q];
		if (cnt > RXK_FLAG_CAP)
			return rxgpu_fail(RXGPU_EUNSUPPORTED, "%d undecided libm discriminator samples in one run (at most %d are recorded)", cnt, RXK_FLAG_CAP);
		if (cnt) {
			RX_HIP(hipMemcpy(s->flag_rec_host, s->flag_rec_dev + (size_t)q * RXK_FLAG_CAP, (size_t)cnt * sizeof(rxk_flag_rec), hipMemcpyDeviceToHost));
			for (int k = 0; k < cnt; k++) {
				const rxk_flag_rec *f = &s->flag_rec_host[k];
				const int16_t v = (int16_t)polar_discriminant_host(f->ar, f->aj, f->br, f->bj);
				RX_HIP(hipMemcpy(r->pcm + pcm_index_host(f->m, r->tiled), &v, 2, hipMemcpyHostToDevice));
			}
			s->fixups += cnt;
			s->flag_cnt_host[q] = 0;
		}
		if (s->p.mode == RXGPU_MODE_RAW)
			continue;
		RX_K(rxk_fm_audio_carry(sb, s->dev, i == 0 ? s->snap_dev + 4 * q : NULL));
		s->pcm = r->pcm;
		s->blk = r->blk;
		s->tiled = r->tiled;
		s->row_audio = r->row_audio;
		if ((rc = run_audio_stages(s, sb, r->g.M, r->g.J, r->d_out)) != RXGPU_OK)
			return rc;
	}
	RX_HIP(hipStreamSynchronize(sb));
	return RXGPU_OK;
}

/* The run in `slot` leaves the pipeline: its demodulated samples are complete (ev_disc), so the count of undecided
 * libm samples is known; nearly always zero. */
static int retire_slot(rxgpu_fm_stream *s, int slot)
{
	int rc;
	if (!s->rec[slot].live)
		return RXGPU_OK;
	RX_HIP(hipEventSynchronize(s->ev_disc[slot]));
	if (s->flag_cnt_host[slot] && (rc = fixup_from(s, slot)) != RXGPU_OK)
		return rc;
	s->rec[slot].live = 0;
	return RXGPU_OK;
}

/* Wait for everything enqueued, settle undecided libm samples, read the carries back. */
static int finish_runs(rxgpu_fm_stream *s)
{
	hipStream_t sa = fm_sa(s), sb = rxgpu_hip_stream2();
	rxk_fm_dev *h = s->dev_host;
	const rxgpu_fm_params *p = &s->p;
	int rc;
	if (!s->pending)
		return RXGPU_OK;
	/* oldest first */
	const int first = (s->rec[0].live && s->rec[1].live) ? (s->rec[0].seq < s->rec[1].seq ? 0 : 1) : (s->rec[0].live ? 0 : 1);
	if ((rc = retire_slot(s, first)) != RXGPU_OK || (rc = retire_slot(s, first ^ 1)) != RXGPU_OK) {
		/* the sequence is lost (e.g. more undecided libm samples in one run than are recorded): leave nothing behind that the next
		 * enqueue would trip over again, and say so to whoever asks for the carries -- set_carry + a replay recovers */
		hipStreamSynchronize(sa);
		hipStreamSynchronize(sb);
		hipStreamSynchronize(fm_sd(s));                  /* the seam/history kernels of a -F run that was enqueued ahead */
		s->rec[0].live = s->rec[1].live = 0;
		s->flag_cnt_host[0] = s->flag_cnt_host[1] = 0;
		s->pending = 0;
		s->chained = 0;
		s->carry_lost = 1;
		return rc;
	}
	const struct run_geom *g = &s->last;
	if (g->passes)
		RX_HIP(hipMemcpyAsync(s->hist_host, s->hist_dev, HIST_TOTAL * 2, hipMemcpyDeviceToHost, sb));
	RX_HIP(hipMemcpyAsync(h, s->dev, sizeof(*h), hipMemcpyDeviceToHost, sb));
	RX_HIP(hipStreamSynchronize(sb));
	RX_HIP(hipStreamSynchronize(sa));
	s->pending = 0;
	s->chained = 0;
	rxgpu_prof_collect();
	if (h->err)
		return rxgpu_fail(RXGPU_ENODEV, "device-side invariant violated in the de-emphasis scan (err=%d)", h->err);

	/* carries out */
	if (!g->passes) {
		s->carry.now_r = h->out_now_r; s->carry.now_j = h->out_now_j; s->carry.prev_index = h->out_prev_index;
	} else {
		const int16_t *hh = s->hist_host;
		for (int i = 0; i < g->passes; i++) {
			memcpy(s->carry.lp_i_hist[i], hh + HIST_CAS_OUT + i * 12, 12);
			memcpy(s->carry.lp_q_hist[i], hh + HIST_CAS_OUT + i * 12 + 6, 12);
		}
		if (p->comp_fir_size == 9) {
			memcpy(s->carry.droop_i_hist, hh + HIST_DROOP_OUT, 18);
			memcpy(s->carry.droop_q_hist, hh + HIST_DROOP_OUT + 9, 18);
		}
	}
	s->carry.pre_r = h->out_pre_r; s->carry.pre_j = h->out_pre_j;
	s->carry.deemph_avg = h->out_deemph_avg;
	s->carry.now_lpr = h->out_now_lpr; s->carry.prev_lpr_index = h->out_prev_lpr_index;
	s->carry.dc_avg = h->out_dc_avg;
	if (p->dc_block_raw)
		RX_HIP(hipMemcpy(&s->carry.dc_avgI, s->rdc_state, 8, hipMemcpyDeviceToHost));
	if (p->squelch_level)                                    /* rtl_fm.c:783-789, block after block */
		for (size_t b = 0; b < s->last_n_blocks; b++)
			s->carry.squelch_hits = s->below_host[b] ? s->carry.squelch_hits + 1 : 0;
	s->h_prev_index = s->carry.prev_index;
	s->h_prev_lpr_index = s->carry.prev_lpr_index;
	return RXGPU_OK;
}

int rxgpu_fm_stream_run_async(rxgpu_fm_stream *s, const int16_t *d_iq, size_t n_blocks, size_t block_len,
                              int16_t *d_out, size_t out_cap, size_t *out_len, int *block_out_len)
{
	int rc;
	struct run_geom g;
	if (!s || !d_iq || !d_out || !n_blocks || (block_len < 2 && !s->allow_empty) || (block_len & 1))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_fm_stream_run: bad arguments");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	if (!s->pending) {
		s->fixups = 0;
		s->h_prev_index = s->carry.prev_index;
		s->h_prev_lpr_index = s->carry.prev_lpr_index;
	}
	if ((rc = run_geometry(s, n_blocks, block_len, out_cap, &g)) != RXGPU_OK)
		return rc;
	if ((rc = enqueue_run(s, d_iq, n_blocks, block_len, d_out, &g)) != RXGPU_OK)
		return rc;
	if (out_len)
		*out_len = (size_t)g.J;
	if (block_out_len)
		block_lengths(s, &g, n_blocks, block_out_len);
	if (s->p.squelch_level || s->p.dc_block_raw)
		return finish_runs(s);               /* squelch_hits is counted on the host, the -E rdc buffer is reused: run by run */
	return RXGPU_OK;
}

int rxgpu_fm_stream_wait(rxgpu_fm_stream *s)
{
	if (!s)
		return rxgpu_fail(RXGPU_EINVAL, "null stream");
	return finish_runs(s);
}

int rxgpu_fm_stream_run(rxgpu_fm_stream *s, const int16_t *d_iq, size_t n_blocks, size_t block_len,
                        int16_t *d_out, size_t out_cap, size_t *out_len, int *block_out_len)
{
	int rc;
	if (s && s->pending && (rc = finish_runs(s)) != RXGPU_OK)
		return rc;
	if ((rc = rxgpu_fm_stream_run_async(s, d_iq, n_blocks, block_len, d_out, out_cap, out_len, block_out_len)) != RXGPU_OK)
		return rc;
	return finish_runs(s);
}

/* chunk of a host-fed call: whole blocks, about 64 MiB (large enough that the per-run launches vanish, small enough
 * that the first chunk's copy -- the only one nothing overlaps -- is a small part of the call) */
#define HOST_CHUNK_BYTES ((size_t)64 << 20)

static int run_host_chunked(rxgpu_fm_stream *s, const int16_t *h_iq, size_t n_blocks, size_t block_len,
                            int16_t *h_out, size_t out_cap, size_t *out_len, int *block_out_len)
{
	int rc;
	hipStream_t sa = rxgpu_hip_stream(), sb = rxgpu_hip_stream2(), sc = rxgpu_hip_stream3();
	size_t chunk_target = HOST_CHUNK_BYTES;
	if (s->k_host_chunk)                                 /* $RXGPU_HOST_CHUNK, bytes; tests use it to get many chunks out of a small capture */
		chunk_target = (size_t)s->k_host_chunk;
	size_t cb = chunk_target / (block_len * 2);
	if (cb < 1) cb = 1;
	if (cb > n_blocks) cb = n_blocks;
	if (cb > s->max_blocks) cb = s->max_blocks;
	/* squelch and -E rdc finish run by run and count on the host: no gain from chunking, keep them whole */
	if (s->p.squelch_level || s->p.dc_block_raw)
		cb = n_blocks;
	if (n_blocks > s->max_blocks && cb == n_blocks)
		return rxgpu_fail(RXGPU_ECAPACITY, "stream created for %zu blocks, run asks %zu", s->max_blocks, n_blocks);
	const size_t chunk_bytes = cb * block_len * 2;
	if (s->stage_in_cap < chunk_bytes) {
		for (int i = 0; i < 3; i++) {
			hipFree(s->stage_in[i]);
			s->stage_in[i] = NULL;
		}
		s->stage_in_cap = 0;
		for (int i = 0; i < 3; i++) {
			RX_HIP(hipMalloc((void **)&s->stage_in[i], chunk_bytes));
			if (!s->ev_h2d[i])
				RX_HIP(hipEventCreateWithFlags(&s->ev_h2d[i], hipEventDisableTiming));
		}
		s->stage_in_cap = chunk_bytes;
	}
	/* the whole call's output stays on the device until the end: raw_demod hands lowpassed[] through (2 int16 per
	 * decimated sample), every other mode at most one per decimated sample; each chunk may round up by one */
	const size_t n_chunks = (n_blocks + cb - 1) / cb;
	size_t per_block_M = s->p.downsample_passes ? ((block_len / 2) >> s->p.downsample_passes) + 1
	                                            : (block_len / 2) / (size_t)(s->p.downsample > 0 ? s->p.downsample : 1) + 2;
	const size_t out_need = (s->p.mode == RXGPU_MODE_RAW ? 2 : 1) * (per_block_M * n_blocks + 2 * n_chunks) + 16;
	if (s->stage_out_cap < out_need) {
		hipFree(s->stage_out);
		s->stage_out = NULL; s->stage_out_cap = 0;
		RX_HIP(hipMalloc((void **)&s->stage_out, out_need * 2));
		s->stage_out_cap = out_need;
	}
	size_t total = 0, done = 0;
	for (size_t c = 0; c < n_chunks; c++) {
		const size_t nb = n_blocks - done < cb ? n_blocks - done : cb;
		const int slot = (int)(c % 3);
		/* chunk c-3 used this buffer; its run was retired when run c-1 was enqueued.  The copy runs on its own stream
		 * while the runs of the chunks before it are on the compute streams. */
		RX_HIP(hipMemcpyAsync(s->stage_in[slot], h_iq + done * block_len, nb * block_len * 2, hipMemcpyHostToDevice, sc));
		RX_HIP(hipEventRecord(s->ev_h2d[slot], sc));
		/* every stream that reads the capture waits for it (decimator on A; discriminator seams, cascade and generic decimator on
		 * B; the -F seam histories on the fourth stream) */
		RX_HIP(hipStreamWaitEvent(rxgpu_hip_stream4(), s->ev_h2d[slot], 0));
		RX_HIP(hipStreamWaitEvent(sa, s->ev_h2d[slot], 0));
		RX_HIP(hipStreamWaitEvent(sb, s->ev_h2d[slot], 0));
		size_t got = 0;
		rc = rxgpu_fm_stream_run_async(s, s->stage_in[slot], nb, block_len, s->stage_out + total, s->stage_out_cap - total, &got,
		                               block_out_len ? block_out_len + done : NULL);
		if (rc != RXGPU_OK) {
			finish_runs(s);
			return rc;
		}
		total += got;
		done += nb;
	}
	if ((rc = finish_runs(s)) != RXGPU_OK)
		return rc;
	if (total > out_cap)
		return rxgpu_fail(RXGPU_ECAPACITY, "output needs %zu int16, capacity %zu", total, out_cap);
	if (total)
		RX_HIP(hipMemcpy(h_out, s->stage_out, total * 2, hipMemcpyDeviceToHost));
	if (out_len)
		*out_len = total;
	return RXGPU_OK;
}

int rxgpu_fm_stream_run_host(rxgpu_fm_stream *s, const int16_t *h_iq, size_t n_blocks, size_t block_len,
                             int16_t *h_out, size_t out_cap, size_t *out_len, int *block_out_len)
{
	int rc;
	if (!s || !h_iq || !h_out || !n_blocks || block_len < 2 || (block_len & 1))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_fm_stream_run_host: bad arguments");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	if (s->pending && (rc = finish_runs(s)) != RXGPU_OK)
		return rc;
	return run_host_chunked(s, h_iq, n_blocks, block_len, h_out, out_cap, out_len, block_out_len);
}

/* ------------------------------------------------------------------ parameter derivation (host only) */

/* demod_init (rtl_fm.c:1086-1115) and the -M switch (1320-1341) */
int rxgpu_fm_params_init(rxgpu_fm_params *p, const char *mode, int *rate_in)
{
	if (!p || !mode)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_fm_params_init: null argument");
	memset(p, 0, sizeof(*p));
	int rin = 24000;                         /* DEFAULT_SAMPLE_RATE, rtl_fm.c:73 */
	p->rate_out = 24000;
	p->rate_out2 = -1;                       /* "flag for disabled" */
	p->post_downsample = 1;
	p->adc_block_const = 9;
	p->rdc_block_const = 9;
	p->output_scale = 1;
	p->mode = RXGPU_MODE_FM;
	if (!strcmp(mode, "fm") || !strcmp(mode, "nbfm") || !strcmp(mode, "nfm")) {
		p->mode = RXGPU_MODE_FM;
	} else if (!strcmp(mode, "raw") || !strcmp(mode, "iq")) {
		p->mode = RXGPU_MODE_RAW;
	} else if (!strcmp(mode, "am")) {
		p->mode = RXGPU_MODE_AM;
	} else if (!strcmp(mode, "usb")) {
		p->mode = RXGPU_MODE_USB;
	} else if (!strcmp(mode, "lsb")) {
		p->mode = RXGPU_MODE_LSB;
	} else if (!strcmp(mode, "wbfm") || !strcmp(mode, "wfm")) {
		p->mode = RXGPU_MODE_FM;
		rin = 170000;
		p->rate_out = 170000;
		p->rate_out2 = 32000;
		p->custom_atan = 1;
		p->deemph = 1;
		p->squelch_level = 0;
	} else {
		return rxgpu_fail(RXGPU_EINVAL, "unknown -M mode \"%s\"", mode);
	}
	if (rate_in)
		*rate_in = rin;
	return RXGPU_OK;
}

/* `rate_in *= post_downsample` (rtl_fm.c:1371), optimal_settings (960-997), deemph_a (1410-1415) */
int rxgpu_fm_plan_settings(rxgpu_fm_params *p, int freq, int rate_in, int edge, int time_constant_us, rxgpu_fm_plan *plan)
{
	if (!p)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_fm_plan_settings: null argument");
	if (rate_in <= 0)
		return rxgpu_fail(RXGPU_EINVAL, "rate_in %d <= 0", rate_in);
	const int post = p->post_downsample > 0 ? p->post_downsample : 1;
	rate_in *= post;
	int downsample = (1000000 / rate_in) + 1;
	int passes = p->downsample_passes;
	if (passes) {
		passes = (int)log2(downsample) + 1;
		downsample = 1 << passes;
	}
	const int capture_rate = downsample * rate_in;
	int capture_freq = freq;
	if (!p->offset_tuning)
		capture_freq = freq + capture_rate / 4;
	capture_freq += edge * rate_in / 2;
	int output_scale = (1 << 15) / (128 * downsample);
	if (output_scale < 1)
		output_scale = 1;
	if (p->mode == RXGPU_MODE_FM)
		output_scale = 1;
	int deemph_a = p->deemph_a;
	if (p->deemph) {
		const double tc = (double)time_constant_us * 1e-6;
		deemph_a = (int)round(1.0 / ((1.0 - exp(-1.0 / (p->rate_out * tc)))));
	}
	p->downsample = downsample;
	p->downsample_passes = passes;
	p->output_scale = output_scale;
	p->deemph_a = deemph_a;
	if (plan) {
		plan->rate_in = rate_in;
		plan->downsample = downsample;
		plan->downsample_passes = passes;
		plan->output_scale = output_scale;
		plan->deemph_a = deemph_a;
		plan->capture_freq = (uint32_t)capture_freq;
		plan->capture_rate = (uint32_t)capture_rate;
	}
	return RXGPU_OK;
}

/* ------------------------------------------------------------------ drop-in entry points */

static void *g_fn_fm, *g_fn_am, *g_fn_usb, *g_fn_lsb, *g_fn_raw;

void rxgpu_set_demod_functions(void *fm, void *am, void *usb, void *lsb, void *raw)
{
	g_fn_fm = fm; g_fn_am = am; g_fn_usb = usb; g_fn_lsb = lsb; g_fn_raw = raw;
}

#define SIDECARS 16
/* side-car of a demod_state: deemph_filter's static accumulator, the stream object, the callback's device buffers for THIS
 * demod_state (raw block, the pre-staged block written / published in turn, the -E rdc scratch) and where the callback left
 * the block it handed over last (slot of cb_pre, its length, still valid?).  Everything per demod_state: two dongle threads
 * feeding two demod_states never touch the same buffer, and a published block can only be consumed by its own full_demod. */
static struct {
	const struct demod_state *d;
	int avg;
	rxgpu_fm_stream *s;
	rxgpu_fm_params p;
	int dev_slot, dev_len, dev_valid;
	int last_sr, last_sr_valid;          /* the squelch rms of the block the last rxgpu_full_demod took (rxgpu_dropin_block_rms) */
	int16_t *cb_in, *cb_pre[2];
	int *cb_rdc;                         /* dc_avgI/Q, the block averages, the int64 sums */
	int16_t *fd_in;                      /* full_demod's own upload buffer (the demod thread's; the callback's are the dongle thread's) */
	unsigned char *fb_dev, *fb_host;     /* the single-block path (dropin_fast_block): device rows + header, their pinned host mirror */
	unsigned char *gb_host;              /* the general path's page-locked mirror of result[] and lowpassed[] (rxgpu_full_demod) */
	const void *zc_in, *zc_out;          /* the callback's zero-copy form: the host buffers whose device addresses are cached below ... */
	void *zc_in_dev, *zc_out_dev;
	unsigned zc_gen;                     /* ... as of this rxgpu_pin_generation(); zc_in_dev == NULL: looked up, not page-locked */
} g_side[SIDECARS];
/* one callback at a time per side-car slot.  The locks live OUTSIDE the records: rxgpu_dropin_release zeroes a record while threads may be
 * queued on its lock, and a mutex must never be copied or cleared under its waiters. */
static pthread_mutex_t g_side_cb_lock[SIDECARS];
static int g_side_cb_lock_ready[SIDECARS];
static pthread_mutex_t g_side_lock = PTHREAD_MUTEX_INITIALIZER;

static int side_slot(const struct demod_state *d)
{
	int free_slot = -1, found = -1;
	pthread_mutex_lock(&g_side_lock);
	for (int i = 0; i < SIDECARS && found < 0; i++) {
		if (g_side[i].d == d)
			found = i;
		else if (!g_side[i].d && free_slot < 0)
			free_slot = i;
	}
	if (found < 0 && free_slot >= 0) {
		g_side[free_slot].d = d;
		if (!g_side_cb_lock_ready[free_slot]) {
			pthread_mutex_init(&g_side_cb_lock[free_slot], NULL);
			g_side_cb_lock_ready[free_slot] = 1;
		}
		found = free_slot;
	}
	pthread_mutex_unlock(&g_side_lock);
	return found;
}

int *rxgpu_deemph_state(const struct demod_state *d)
{
	int i = side_slot(d);
	return i < 0 ? NULL : &g_side[i].avg;
}

int rxgpu_dropin_block_rms(const struct demod_state *d, int *sr)
{
	int i = side_slot(d);
	if (i < 0 || !sr)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_dropin_block_rms: unknown demod_state");
	if (!g_side[i].last_sr_valid)
		return rxgpu_fail(RXGPU_EUNSUPPORTED, "no squelch on the last block: rms(d->lowpassed, d->lp_len, 1) is the value (lowpassed[] is intact)");
	*sr = g_side[i].last_sr;
	return RXGPU_OK;
}

/* Forget a demod_state: its stream object, device buffers, de-emphasis accumulator and side-car slot (SIDECARS of them exist).
 * For callers that create and destroy demod_state objects; the reference's own is a global that lives as long as the process. */
int rxgpu_dropin_release(const struct demod_state *d)
{
	int found = -1;
	pthread_mutex_lock(&g_side_lock);
	for (int i = 0; i < SIDECARS; i++)
		if (g_side[i].d == d && d)
			found = i;
	if (found >= 0) {
		const int i = found;
		/* a callback of this demod_state that is INSIDE its critical section finishes first; one still queued on the lock finds
		 * the slot no longer its own when it gets in (rxgpu_callback re-checks g_side[side].d under the lock) and resolves again */
		pthread_mutex_lock(&g_side_cb_lock[i]);
		if (g_side[i].s)
			rxgpu_fm_stream_destroy(g_side[i].s);
		hipFree(g_side[i].cb_in); hipFree(g_side[i].cb_rdc); hipFree(g_side[i].cb_pre[0]); hipFree(g_side[i].cb_pre[1]);
		hipFree(g_side[i].fd_in);
		hipFree(g_side[i].fb_dev);
		if (g_side[i].fb_host) hipHostFree(g_side[i].fb_host);
		if (g_side[i].gb_host) hipHostFree(g_side[i].gb_host);
		memset(&g_side[i], 0, sizeof(g_side[i]));
		pthread_mutex_unlock(&g_side_cb_lock[i]);
	}
	pthread_mutex_unlock(&g_side_lock);
	return found >= 0 ? RXGPU_OK : rxgpu_fail(RXGPU_EINVAL, "rxgpu_dropin_release: no side-car for this demod_state");
}

void rxgpu_dropin_invalidate(const struct demod_state *d)
{
	int i = side_slot(d);
	if (i >= 0)
		g_side[i].dev_valid = 0;
}

/* print once on stderr, release the device, _exit(1): rxgpu_fatal (rxgpu_rt.c) */
static void die(const char *what)
{
	rxgpu_fatal(what);
}

/* Page-lock the struct members the drop-in DMAs from/to (SURVEY.md section 8b "Ownership"): lowpassed[] .. result[] of
 * the demod_state as one range, buf16[] of the dongle_state.  Explicit and optional -- the structs must outlive the
 * registration (the reference's are globals, rtl_fm.c:190-191); without it the copies go through the runtime's
 * own bounce buffers.  Either pointer may be NULL. */
static int pin_range(const void *ptr, size_t bytes, int on)
{
	/* exactly the member's bytes, NOT rounded out to pages: the runtime resolves a host pointer by the registered
	 * ranges, and a foreign buffer that merely shares a boundary page with the struct must not resolve to this one
	 * (its copies would then fail with hipErrorInvalidValue as soon as they run past the registration's end) */
	rxgpu_pin_changed();
	if (on)
		RX_HIP(hipHostRegister((void *)ptr, bytes, hipHostRegisterDefault));
	else
		RX_HIP(hipHostUnregister((void *)ptr));
	return RXGPU_OK;
}

int rxgpu_dropin_pin(struct demod_state *d, struct dongle_state *s)
{
	int rc;
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	if (d && (rc = pin_range(d->lowpassed, (size_t)((char *)(d->result + RXGPU_MAXIMUM_BUF_LENGTH) - (char *)d->lowpassed), 1)) != RXGPU_OK)
		return rc;
	if (s && (rc = pin_range(s->buf16, sizeof(s->buf16), 1)) != RXGPU_OK)
		return rc;
	return RXGPU_OK;
}

int rxgpu_dropin_unpin(struct demod_state *d, struct dongle_state *s)
{
	int rc = RXGPU_OK, r2;
	if (d)
		rc = pin_range(d->lowpassed, 0, 0);
	if (s && (r2 = pin_range(s->buf16, 0, 0)) != RXGPU_OK)
		rc = r2;
	return rc;
}

/* Where a drop-in call pair spends its time (host clock, $RXGPU_DROPIN_TIMING=1; rxgpu_dropin_timing reads and clears):
 *   0 callback: H2D of the raw block + pre-stage kernel + D2H into buf16 (enqueue .. stream sync)
 *   1 callback: the hand-off (d->rw, memcpy into lowpassed[], publication, cond_signal)
 *   2 full_demod: parameters, side-car, carries in
 *   3 full_demod: the run (upload if the block is not the pre-staged one, every kernel, carries back)
 *   4 full_demod: D2H of result[] and lowpassed[], carries into the struct
 *   5 calls of rxgpu_callback, 6 calls of rxgpu_full_demod */
static double g_dt[7];
static int g_dt_on = -1;
static double now_us(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
static int dt_on(void)
{
	if (g_dt_on < 0) {
		const char *e = rxgpu_knob("RXGPU_DROPIN_TIMING");
		g_dt_on = e && atoi(e) > 0;
	}
	return g_dt_on;
}
int rxgpu_dropin_timing(double *us, int n)
{
	for (int i = 0; i < n && i < 7; i++) {
		us[i] = g_dt[i];
		g_dt[i] = 0;
	}
	return n < 7 ? n : 7;
}

/* the callback's device buffers of one demod_state, allocated on first use */
static int side_buffers(int slot)
{
	if (g_side[slot].cb_in)
		return RXGPU_OK;
	if (hipMalloc((void **)&g_side[slot].cb_in, RXGPU_MAXIMUM_BUF_LENGTH * 2 + 16) != hipSuccess ||
	    hipMalloc((void **)&g_side[slot].cb_rdc, 64) != hipSuccess ||
	    hipMalloc((void **)&g_side[slot].cb_pre[0], RXGPU_MAXIMUM_BUF_LENGTH * 2 + 16) != hipSuccess ||
	    hipMalloc((void **)&g_side[slot].cb_pre[1], RXGPU_MAXIMUM_BUF_LENGTH * 2 + 16) != hipSuccess)
		return rxgpu_fail(RXGPU_ENOMEM, "hipMalloc failed");
	return RXGPU_OK;
}

/* rxgpu_shutdown: everything the drop-ins keep alive between calls belongs to the device that is going away */
void rxgpu_fm_dropin_release(void)
{
	pthread_mutex_lock(&g_side_lock);
	for (int i = 0; i < SIDECARS; i++) {
		if (g_side[i].s)
			rxgpu_fm_stream_destroy(g_side[i].s);
		g_side[i].s = NULL;
		hipFree(g_side[i].cb_in); hipFree(g_side[i].cb_rdc); hipFree(g_side[i].cb_pre[0]); hipFree(g_side[i].cb_pre[1]);
		hipFree(g_side[i].fd_in);
		hipFree(g_side[i].fb_dev);
		if (g_side[i].fb_host) hipHostFree(g_side[i].fb_host);
		if (g_side[i].gb_host) hipHostFree(g_side[i].gb_host);
		g_side[i].fb_dev = g_side[i].fb_host = g_side[i].gb_host = NULL;
		g_side[i].zc_gen = 0;
		g_side[i].cb_in = g_side[i].cb_pre[0] = g_side[i].cb_pre[1] = g_side[i].fd_in = NULL;
		g_side[i].cb_rdc = NULL;
		g_side[i].dev_valid = 0;
	}
	pthread_mutex_unlock(&g_side_lock);
}

/* ---- the drop-in's single blocks of the plain FM chain: k_fm_block_dd + k_fm_row_audio, carries as kernel arguments, ONE copy back.
 * (rxgpu_fm_stream_run on one block: a dozen launches over four streams, carries up and down -- 84 us of 177 per 1 MiB block pair in round 3.) */
#define FB_HDR 2048                      /* rxk_blk_out, then audio_in[3], audio_out[3] at FB_AUDIO */
#define FB_AUDIO 1600
#define FB_MCAP (RXGPU_MAXIMUM_BUF_LENGTH / 2 / 8 + 64)
#define FB_BYTES (FB_HDR + 10 * (size_t)FB_MCAP + 256)

static int dropin_fast_ok(const rxgpu_fm_params *p, const struct demod_state *d)
{
	const char *k = rxgpu_knob("RXGPU_DROPIN_FAST");                 /* "0": always the general path (A/B, tests) */
	if (k && k[0] == '0')
		return 0;
	if (p->mode != RXGPU_MODE_FM || p->downsample_passes || p->squelch_level || p->dc_block_audio || p->post_downsample > 1 || p->custom_atan == 2)
		return 0;
	if (p->downsample < 8 || d->prev_index < 0 || d->prev_index >= p->downsample || d->lp_len < 2)
		return 0;
	if (((long)d->prev_index + d->lp_len / 2) / p->downsample < 1)   /* no decimated sample: fm_demod's corner cases stay where they are handled */
		return 0;
	if (p->deemph && p->deemph_a < 1)
		return 0;
	if (p->rate_out2 > 0 && (p->rate_out < p->rate_out2 || p->rate_out <= 0 || d->prev_lpr_index < 0 || d->prev_lpr_index >= p->rate_out))
		return 0;
	return 1;
}

static int dropin_fast_block(int slot, struct demod_state *d, const rxgpu_fm_params *p, const int16_t *d_block, int timing, double *t_a)
{
	hipStream_t st = rxgpu_hip_stream2();
	if (!g_side[slot].fb_dev) {
		if (hipMalloc((void **)&g_side[slot].fb_dev, FB_BYTES) != hipSuccess || hipHostMalloc((void **)&g_side[slot].fb_host, FB_BYTES, 0) != hipSuccess ||
		    hipMemset(g_side[slot].fb_dev, 0, FB_HDR) != hipSuccess)
			return rxgpu_fail(RXGPU_ENOMEM, "single-block workspace allocation failed");
	}
	unsigned char *dev = g_side[slot].fb_dev, *host = g_side[slot].fb_host;
	const unsigned n = (unsigned)d->lp_len / 2;
	const int ds = p->downsample;
	const unsigned long long M = ((unsigned long long)d->prev_index + n) / (unsigned long long)ds;
	const int resample = p->rate_out2 > 0;
	const unsigned long long J = resample ? ((unsigned long long)d->prev_lpr_index + M * (unsigned long long)p->rate_out2) / (unsigned long long)p->rate_out : M;
	const size_t row_b = (2 * (size_t)M + 15) & ~(size_t)15;
	rxk_blk_out *hdr = (rxk_blk_out *)dev;
	int *audio = (int *)(dev + FB_AUDIO);
	int16_t *pcm = (int16_t *)(dev + FB_HDR);                    /* the demodulated row, then (in place) the result */
	uint32_t *lp = (uint32_t *)(dev + FB_HDR + row_b);
	int16_t *keep = (int16_t *)(dev + FB_HDR + row_b + 4 * (size_t)M + 16);   /* the demodulated row once more: what a host fix-up starts again from */
	const int avg = g_side[slot].avg;
	const int serial = p->deemph && (p->deemph_a < 2 || p->deemph_a > 64 || avg < -32768 || avg > 32767);
	const int warm = p->deemph && !serial ? rxgpu_deemph_warm64(p->deemph_a) : 8;
	/* the page-locked mirror, same layout: the kernels write what the caller gets straight into it (hipHostMalloc'd memory has one address on both sides) */
	uint32_t *lp_h = (uint32_t *)(host + FB_HDR + row_b);
	int16_t *row_h = (int16_t *)(host + FB_HDR);
	int *audio_h = (int *)(host + FB_AUDIO) + 3;
	RX_K(rxk_fm_block_dd(st, d_block, n, ds, d->prev_index, d->now_r, d->now_j, d->pre_r, d->pre_j, p->custom_atan, lp, lp_h, pcm, keep, hdr, audio,
	                     avg, d->now_lpr, d->prev_lpr_index));
	for (int attempt = 0; ; attempt++) {
		RX_K(rxk_fm_row_audio(st, pcm, pcm, (unsigned)M, p->deemph, p->deemph_a, warm, serial, p->rate_out, resample ? p->rate_out2 : 0, (unsigned)J, audio,
		                      audio + 3, row_h, audio_h, hdr, host, (unsigned)(sizeof(rxk_blk_out) / 4)));
		RX_HIP(hipStreamSynchronize(st));
		const rxk_blk_out *h = (const rxk_blk_out *)host;
		if (!h->flag_cnt || attempt)
			break;
		/* libm samples the device could not decide: the host's libm, like fixup_from; then the audio stages again from the row as demodulated */
		const int cnt = h->flag_cnt;
		RX_HIP(hipMemsetAsync(&hdr->flag_cnt, 0, sizeof(int), st));
		if (cnt > RXK_BLK_FLAGS) {
			RX_HIP(hipStreamSynchronize(st));
			return RXGPU_EUNSUPPORTED;
		}
		RX_HIP(hipMemcpyAsync(pcm, keep, 2 * (size_t)M, hipMemcpyDeviceToDevice, st));
		for (int i = 0; i < cnt; i++) {
			const rxk_flag_rec *f = &h->rec[i];
			const int16_t v = (int16_t)polar_discriminant_host(f->ar, f->aj, f->br, f->bj);
			RX_HIP(hipMemcpyAsync(pcm + f->m, &v, 2, hipMemcpyHostToDevice, st));
			RX_HIP(hipStreamSynchronize(st));                    /* v lives on this stack frame */
		}
		if (g_side[slot].s)
			g_side[slot].s->fixups += cnt;
	}
	if (timing) { const double t_b = now_us(); g_dt[3] += t_b - *t_a; *t_a = t_b; }
	const rxk_blk_out *h = (const rxk_blk_out *)host;
	const int *audio_out = (const int *)(host + FB_AUDIO) + 3;
	memcpy(d->result, host + FB_HDR, 2 * (size_t)J);
	memcpy(d->lowpassed, host + FB_HDR + row_b, 4 * (size_t)M);
	d->lp_len = (int)(2 * M);
	d->result_len = (int)J;
	d->now_r = h->now_r; d->now_j = h->now_j; d->prev_index = h->prev_index;
	d->pre_r = h->pre_r; d->pre_j = h->pre_j;
	g_side[slot].avg = audio_out[0];
	d->now_lpr = audio_out[1]; d->prev_lpr_index = audio_out[2];
	g_side[slot].dev_valid = 0;
	g_side[slot].last_sr_valid = 0;
	if (timing) { g_dt[4] += now_us() - *t_a; g_dt[6] += 1; }
	return RXGPU_OK;
}

void rxgpu_full_demod(struct demod_state *d)
{
	const int timing = dt_on();
	double t_a = timing ? now_us() : 0, t_b;
	int slot = side_slot(d);
	if (slot < 0) {
		rxgpu_fail(RXGPU_ECAPACITY, "more than %d demod_state objects", SIDECARS);
		die("rxgpu_full_demod");
	}
	if (rxgpu_ensure_init() != RXGPU_OK)
		die("rxgpu_full_demod");
	int mode = RXGPU_MODE_FM;
	if (g_fn_fm) {
		void *fn = (void *)d->mode_demod;
		if (fn == g_fn_fm) mode = RXGPU_MODE_FM;
		else if (fn == g_fn_am) mode = RXGPU_MODE_AM;
		else if (fn == g_fn_usb) mode = RXGPU_MODE_USB;
		else if (fn == g_fn_lsb) mode = RXGPU_MODE_LSB;
		else if (fn == g_fn_raw) mode = RXGPU_MODE_RAW;
		else {
			rxgpu_fail(RXGPU_EUNSUPPORTED, "mode_demod %p is none of the registered demodulators", fn);
			die("rxgpu_full_demod");
		}
	}
	/* readStream may return any element count (rtl_fm.c:894-899): lp_len is whatever the callback got, times two */
	if (d->lp_len < 0 || d->lp_len > RXGPU_MAXIMUM_BUF_LENGTH || (d->lp_len & 1)) {
		rxgpu_fail(RXGPU_EINVAL, "lp_len %d (the callback stores twice readStream's element count, rtl_fm.c:899: even, at most %d)", d->lp_len, RXGPU_MAXIMUM_BUF_LENGTH);
		die("rxgpu_full_demod");
	}
	rxgpu_fm_params p;
	memset(&p, 0, sizeof(p));
	p.downsample = d->downsample;
	p.downsample_passes = d->downsample_passes;
	p.comp_fir_size = d->comp_fir_size;
	p.custom_atan = d->custom_atan;
	p.deemph = d->deemph;
	p.deemph_a = d->deemph_a;
	p.rate_out = d->rate_out;
	p.rate_out2 = d->rate_out2;
	p.prescaled = 1;                       /* lowpassed[] is already scaled + rotated by the callback */
	p.mode = mode;
	p.output_scale = d->output_scale;
	p.squelch_level = d->squelch_level;
	p.dc_block_audio = d->dc_block_audio;
	p.adc_block_const = d->adc_block_const;
	p.post_downsample = d->post_downsample;             /* -o; -E rdc already happened in the callback */
	if (!g_side[slot].s || memcmp(&p, &g_side[slot].p, sizeof(p))) {
		if (g_side[slot].s)
			rxgpu_fm_stream_destroy(g_side[slot].s);
		g_side[slot].s = NULL;
		if (rxgpu_fm_stream_create(&g_side[slot].s, &p, 1, RXGPU_MAXIMUM_BUF_LENGTH) != RXGPU_OK)
			die("rxgpu_full_demod");
		g_side[slot].s->allow_empty = 1;                /* a block that completes no window is legal here, see below */
		g_side[slot].s->one_stream = 1;                 /* one block per run: nothing to overlap, no event hops between streams */
		g_side[slot].p = p;
	}
	rxgpu_fm_stream *s = g_side[slot].s;
	hipStream_t sb = rxgpu_hip_stream2();
	const int16_t *d_block;
	if (g_side[slot].dev_valid && g_side[slot].dev_len == d->lp_len && d->lp_len > 0 && g_side[slot].cb_pre[g_side[slot].dev_slot]) {
		/* the block is the one rxgpu_callback pre-staged: it is still in HBM, no second trip over PCIe.  (The caller
		 * holds d->rw like the reference's demod thread, rtl_fm.c:922-924, so the callback cannot publish meanwhile.) */
		d_block = g_side[slot].cb_pre[g_side[slot].dev_slot];
	} else {
		/* any other lowpassed[]: up it goes.  At least the first two int16: an empty -F block still runs fifth_order on lowpassed[0]
		 * and lowpassed[1] (rtl_fm.c:419-423 reads data[0] whatever the length) */
		if (!g_side[slot].fd_in && hipMalloc((void **)&g_side[slot].fd_in, RXGPU_MAXIMUM_BUF_LENGTH * 2 + 16) != hipSuccess) {
			rxgpu_fail(RXGPU_ENOMEM, "hipMalloc failed");
			die("rxgpu_full_demod");
		}
		const size_t up = (size_t)(d->lp_len > 2 ? d->lp_len : 2) * 2;
		if (hipMemcpyAsync(g_side[slot].fd_in, d->lowpassed, up, hipMemcpyHostToDevice, sb) != hipSuccess) {
			rxgpu_fail(RXGPU_ENODEV, "upload of lowpassed[] failed");
			die("rxgpu_full_demod");
		}
		if (hipStreamSynchronize(sb) != hipSuccess) {       /* the run below starts on the other streams */
			rxgpu_fail(RXGPU_ENODEV, "upload of lowpassed[] failed");
			die("rxgpu_full_demod");
		}
		d_block = g_side[slot].fd_in;
	}
	/* one block of the plain chain (low_pass -> fm_demod -> deemph_filter -> low_pass_real): two launches and one copy back */
	if (dropin_fast_ok(&p, d)) {
		const int rc = dropin_fast_block(slot, d, &p, d_block, timing, &t_a);
		if (rc == RXGPU_OK)
			return;
		if (rc != RXGPU_EUNSUPPORTED)
			die("rxgpu_full_demod");
		/* more undecided libm samples than the block header holds: the general path, nothing has been written to *d yet */
	}
	rxgpu_fm_carry c;
	memset(&c, 0, sizeof(c));
	c.now_r = d->now_r; c.now_j = d->now_j; c.prev_index = d->prev_index;
	c.pre_r = d->pre_r; c.pre_j = d->pre_j;
	memcpy(c.lp_i_hist, d->lp_i_hist, sizeof(c.lp_i_hist));
	memcpy(c.lp_q_hist, d->lp_q_hist, sizeof(c.lp_q_hist));
	memcpy(c.droop_i_hist, d->droop_i_hist, sizeof(c.droop_i_hist));
	memcpy(c.droop_q_hist, d->droop_q_hist, sizeof(c.droop_q_hist));
	c.deemph_avg = g_side[slot].avg;
	c.now_lpr = d->now_lpr; c.prev_lpr_index = d->prev_lpr_index;
	c.squelch_hits = d->squelch_hits; c.dc_avg = d->dc_avg;
	rxgpu_fm_stream_set_carry(s, &c);
	const int pre_r_in = d->pre_r, pre_j_in = d->pre_j;
	size_t got = 0;
	const size_t cap = (size_t)RXGPU_MAXIMUM_BUF_LENGTH + 16;
	/* result[] and lowpassed[] come home through a page-locked mirror the run's last kernels write themselves (hipHostMalloc'd memory has one address on
	 * both sides): no copy call behind the run -- two of them, each staged and waited for by the runtime, were 21 us per block (round 6) */
	if (!g_side[slot].gb_host && hipHostMalloc((void **)&g_side[slot].gb_host, (size_t)RXGPU_MAXIMUM_BUF_LENGTH * 4 + 128, 0) != hipSuccess) {
		rxgpu_fail(RXGPU_ENOMEM, "hipHostMalloc failed");
		die("rxgpu_full_demod");
	}
	int16_t *const m_res = (int16_t *)g_side[slot].gb_host, *const m_lp = m_res + cap + 16;
	s->lp_mirror = m_lp;
	if (timing) { t_b = now_us(); g_dt[2] += t_b - t_a; t_a = t_b; }
	if (rxgpu_fm_stream_run(s, d_block, 1, (size_t)d->lp_len, m_res, cap, &got, NULL) != RXGPU_OK)
		die("rxgpu_full_demod");
	g_side[slot].dev_valid = 0;
	rxgpu_fm_stream_get_carry(s, &c);
	if (timing) { t_b = now_us(); g_dt[3] += t_b - t_a; t_a = t_b; }
	if (got > RXGPU_MAXIMUM_BUF_LENGTH) {
		rxgpu_fail(RXGPU_ECAPACITY, "result needs %zu int16", got);
		die("rxgpu_full_demod");
	}
	const struct run_geom *g = &s->last;
	const int lp_len_out = g->passes ? (g->literal ? g->lf : (int)(2 * g->K)) : (int)(2 * g->M);
	{
		const size_t back = g->passes ? (size_t)(lp_len_out > 2 ? lp_len_out : 2) : (size_t)lp_len_out;
		if (got)
			memcpy(d->result, m_res, got * 2);
		if (back)
			memcpy(d->lowpassed, m_lp, back * 2);
		d->lp_len = lp_len_out;
	}
	d->result_len = (int)got;
	d->now_r = c.now_r; d->now_j = c.now_j; d->prev_index = c.prev_index;
	d->pre_r = c.pre_r; d->pre_j = c.pre_j;
	memcpy(d->lp_i_hist, c.lp_i_hist, sizeof(c.lp_i_hist));
	memcpy(d->lp_q_hist, c.lp_q_hist, sizeof(c.lp_q_hist));
	memcpy(d->droop_i_hist, c.droop_i_hist, sizeof(c.droop_i_hist));
	memcpy(d->droop_q_hist, c.droop_q_hist, sizeof(c.droop_q_hist));
	g_side[slot].avg = c.deemph_avg;
	d->now_lpr = c.now_lpr; d->prev_lpr_index = c.prev_lpr_index;
	d->squelch_hits = c.squelch_hits; d->dc_avg = c.dc_avg;
	g_side[slot].last_sr_valid = p.squelch_level != 0;
	if (p.squelch_level) {
		const int sr = s->below_host[s->max_blocks + 1];
		g_side[slot].last_sr = sr < 0 ? INT_MIN : sr;       /* an empty block: (int)NaN, rtl_fm.c:756 on x86-64 */
	}
	if (lp_len_out < 2 && mode == RXGPU_MODE_FM) {
		/* Fewer than one decimated sample (a read shorter than the decimation).  fm_demod (rtl_fm.c:584-615) still writes result[0]
		 * from lp[0], lp[1] against the old pre_r/pre_j, and takes the new pre_r/pre_j from lp[lp_len-2], lp[lp_len-1] -- in FRONT of
		 * lowpassed[] (the tail of d->thread) for lp_len 0 or 1.  The device never sees that memory; the drop-in has the real struct
		 * and finishes the block the way the C does.  (lowpassed[0], [1]: what low_pass left untouched, or what the last fifth_order
		 * pass wrote -- copied back above.) */
		const int16_t *lp = d->lowpassed;
		const int16_t *end = (const int16_t *)((const char *)d + offsetof(struct demod_state, lowpassed)) + lp_len_out;
		d->result[0] = (int16_t)polar_discriminant_host(lp[0], lp[1], pre_r_in, pre_j_in);
		d->pre_r = end[-2];
		d->pre_j = end[-1];
	}
	if (timing) { g_dt[4] += now_us() - t_a; g_dt[6] += 1; }
}

void rxgpu_callback(int16_t *buf, uint32_t len, void *ctx)
{
	struct dongle_state *s = ctx;
	struct demod_state *d;
	if (!s)
		return;
	d = s->demod_target;
	if (rxgpu_ensure_init() != RXGPU_OK)
		die("rxgpu_callback");
	if (len > RXGPU_MAXIMUM_BUF_LENGTH || (len & 1)) {
		/* the reference passes r * 2 for r <= MAXIMUM_BUF_LENGTH / 2 elements (rtl_fm.c:872, 894-899): even, bounded */
		rxgpu_fail(RXGPU_EINVAL, "callback length %u", len);
		die("rxgpu_callback");
	}
	int side = side_slot(d);
	if (side < 0) {
		rxgpu_fail(RXGPU_ECAPACITY, "more than %d demod_state objects", SIDECARS);
		die("rxgpu_callback");
	}
	if (s->mute) {                                   /* rtl_fm.c:839-843 */
		for (int i = 0; i < s->mute && i < (int)len; i++)
			buf[i] = 0;
		s->mute = 0;
	}
	if (!len && d->dc_block_raw) {
		rxgpu_fail(RXGPU_EUNSUPPORTED, "-E rdc on an empty read divides by zero in the reference (rtl_fm.c:711)");
		die("rxgpu_callback");
	}
	const int timing = dt_on();
	double t_a = timing ? now_us() : 0, t_b;
	/* the buffers below belong to this demod_state; one callback at a time on them.  The slot was resolved before the lock: if
	 * rxgpu_dropin_release ran while this thread was queued, the record is zeroed (or already someone else's) -- resolve again */
	for (;;) {
		pthread_mutex_lock(&g_side_cb_lock[side]);
		if (__atomic_load_n(&g_side[side].d, __ATOMIC_ACQUIRE) == d)
			break;
		pthread_mutex_unlock(&g_side_cb_lock[side]);
		side = side_slot(d);
		if (side < 0) {
			rxgpu_fail(RXGPU_ECAPACITY, "more than %d demod_state objects", SIDECARS);
			die("rxgpu_callback");
		}
	}
	if (side_buffers(side) != RXGPU_OK) {
		pthread_mutex_unlock(&g_side_cb_lock[side]);
		die("rxgpu_callback");
	}
	int16_t *const cb_in = g_side[side].cb_in;
	int *const cb_rdc = g_side[side].cb_rdc;
	/* write the slot that is NOT published: full_demod may be reading the published one right now (it runs under
	 * d->rw; the publication below happens under d->rw too) */
	const int w = g_side[side].dev_valid ? g_side[side].dev_slot ^ 1 : 0;
	int16_t *pre = g_side[side].cb_pre[w];
	hipStream_t st = rxgpu_hip_stream3();            /* its own stream: the demod thread's runs use the other two */
	/* Both the read buffer and buf16[] page-locked (rxgpu_pin / rxgpu_dropin_pin): the block crosses PCIe inside ONE launch.
	 * The device addresses are looked up once per buffer pair and registration state. */
	int zc = 0;
	if (len && !d->dc_block_raw && !(rxgpu_knob("RXGPU_DROPIN_ZC") && rxgpu_knob("RXGPU_DROPIN_ZC")[0] == '0')) {
		const unsigned gen = rxgpu_pin_generation();
		if (g_side[side].zc_gen != gen || g_side[side].zc_in != (const void *)buf || g_side[side].zc_out != (const void *)s->buf16) {
			void *a = NULL, *b = NULL;
			void *a_end = NULL;
			/* the whole block, not just its first byte (a caller may have page-locked less than it reads), and 8-byte pieces */
			if (hipHostGetDevicePointer(&a, buf, 0) != hipSuccess || hipHostGetDevicePointer(&a_end, (char *)buf + RXGPU_MAXIMUM_BUF_LENGTH * 2 - 1, 0) != hipSuccess ||
			    hipHostGetDevicePointer(&b, s->buf16, 0) != hipSuccess || (((size_t)a | (size_t)b) & 7u)) {
				(void)hipGetLastError();                 /* not page-locked: the copies below */
				a = b = NULL;
			}
			g_side[side].zc_in = buf; g_side[side].zc_out = s->buf16;
			g_side[side].zc_in_dev = a; g_side[side].zc_out_dev = b;
			g_side[side].zc_gen = gen;
		}
		zc = g_side[side].zc_in_dev != NULL && g_side[side].zc_out_dev != NULL;
	}
	int ok = 1;
	if (zc) {
		ok = !RX_FAULT() &&
		     rxk_fm_prestage_zc(st, (const int16_t *)g_side[side].zc_in_dev, len / 2, !s->offset_tuning, pre, (int16_t *)g_side[side].zc_out_dev) == 0 &&
		     hipStreamSynchronize(st) == hipSuccess;
	} else if (len) {
		ok = hipMemcpyAsync(cb_in, buf, (size_t)len * 2, hipMemcpyHostToDevice, st) == hipSuccess;
	}
	if (zc) {
		/* done */
	} else if (ok && len && d->dc_block_raw) {
		/* rtl_fm.c:850-852: scale, dc_block_raw_filter, rotate; cb_rdc = state[2] | avg[2] | sums[2] */
		int state[2] = { d->dc_avgI, d->dc_avgQ };
		ok = hipMemcpyAsync(cb_rdc, state, 8, hipMemcpyHostToDevice, st) == hipSuccess &&
		     rxk_fm_rdc(st, cb_in, 1, len / 2, 0, !s->offset_tuning, d->rdc_block_const, cb_rdc, (long long *)(cb_rdc + 4),
		                cb_rdc + 2, pre) == 0 &&
		     hipMemcpyAsync(state, cb_rdc, 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
		     hipMemcpyAsync(s->buf16, pre, (size_t)len * 2, hipMemcpyDeviceToHost, st) == hipSuccess &&
		     hipStreamSynchronize(st) == hipSuccess;
		d->dc_avgI = state[0];
		d->dc_avgQ = state[1];
	} else if (ok && len) {
		ok = !RX_FAULT() && rxk_fm_prestage(st, cb_in, len / 2, !s->offset_tuning, pre) == 0 &&
		     hipMemcpyAsync(s->buf16, pre, (size_t)len * 2, hipMemcpyDeviceToHost, st) == hipSuccess &&
		     hipStreamSynchronize(st) == hipSuccess;
	}
	if (!ok) {
		pthread_mutex_unlock(&g_side_cb_lock[side]);
		rxgpu_fail(RXGPU_ENODEV, "device pre-stage failed: %s", hipGetErrorString(hipGetLastError()));
		die("rxgpu_callback");
	}
	if (timing) { t_b = now_us(); g_dt[0] += t_b - t_a; t_a = t_b; }
	pthread_rwlock_wrlock(&d->rw);                   /* rtl_fm.c:858-862 */
	memcpy(d->lowpassed, s->buf16, 2 * (size_t)len);
	d->lp_len = (int)len;
	g_side[side].dev_slot = w;
	g_side[side].dev_len = (int)len;
	g_side[side].dev_valid = len > 0;
	pthread_rwlock_unlock(&d->rw);
	pthread_mutex_unlock(&g_side_cb_lock[side]);
	pthread_mutex_lock(&d->ready_m);
	pthread_cond_signal(&d->ready);
	pthread_mutex_unlock(&d->ready_m);
	if (timing) { g_dt[1] += now_us() - t_a; g_dt[5] += 1; }
}
