/* rxgpu.h -- C ABI of librxgpu.so: the rx_tools sample-stream DSP path on MI355X (gfx950).
 *
 * Plain C, plain pointers and sizes.  The reference (rxseger/rx_tools v1.0.3) has no
 * plugin/FFI layer; its boundary for this path is four C call sites and the structs they
 * mutate.  Each entry point below names the reference interface it replaces (file:line
 * under /root/reference/src).  INTEGRATION.md shows the two-line patches that bind them.
 *
 * Error convention: the reference's path functions are void and report to stderr
 * (SURVEY.md section 8b).  The drop-in entry points keep that; everything else returns 0 on
 * success or a negative RXGPU_E* code, with text from rxgpu_last_error().  Nothing here
 * ever writes to stdout (fd 1 carries the audio/CSV stream in the reference), and there
 * is NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef RXGPU_H
#define RXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct demod_state;     /* rtl_fm.c:124-159    (layout: rxgpu_ref_structs.h) */
struct dongle_state;    /* rtl_fm.c:104-122 */
struct tuning_state;    /* rtl_power.c:89-108 */

enum {
	RXGPU_OK = 0,
	RXGPU_ENODEV = -1,        /* no HIP device / runtime error (text in rxgpu_last_error) */
	RXGPU_EINVAL = -2,        /* bad argument */
	RXGPU_EUNSUPPORTED = -3,  /* geometry or mode outside what the device path implements */
	RXGPU_ENOMEM = -4,
	RXGPU_ECAPACITY = -5      /* caller-provided buffer too small */
};

/* ------------------------------------------------------------------ runtime */

/* Bind the calling process to one HIP device and create the library's streams.
 * device < 0: use $RXGPU_DEVICE, else $LOCAL_RANK, else 0 (drop-in binaries gain no new
 * flag; SURVEY.md section 5 "Config").  Idempotent for the same device. */
int rxgpu_init(int device);
void rxgpu_shutdown(void);
int rxgpu_device_count(void);
const char *rxgpu_last_error(void);
/* the hipStream_t (as void*) all kernels of this library are launched on */
void *rxgpu_stream(void);
int rxgpu_sync(void);
/* The $RXGPU_* tuning knobs (last table) are read from the environment at rxgpu_init and whenever a stream,
 * channeliser or scan object is created -- never on a per-block path -- and an object keeps the plan it was created with.
 * rxgpu_knobs_reload re-reads them now (A/B tools that flip a kernel-variant knob between runs of one object). */
void rxgpu_knobs_reload(void);
/* Page-lock / release a host buffer in place (hipHostRegister), so that the host-fed entry points
 * (rxgpu_fm_stream_run_host, the drop-ins) DMA it without a bounce.  Optional: pageable memory works too. */
int rxgpu_pin(void *ptr, size_t bytes);
int rxgpu_unpin(void *ptr);

/* Diagnostics: GB/s (read + written bytes) this box's HBM gives a plain stream in the access shapes of the HBM-bound kernels, arithmetic
 * taken out -- mode 0: 16 B read per unit, nothing written; 1: 16 B read, 16 B written; 2: 8 B read, 16 B written (CS16->CF32); 3: 16 B
 * read, 8 B written (CS16->CS8); 4: 16 B read through the round-3 grid-stride loop (what the span order replaced).  bench.py prints every HBM-bound leg beside the ceiling of the box it ran on. */
int rxgpu_diag_stream_rate(int mode, size_t units, int reps, double *gbs);

/* Per-kernel device timing with hipEvents on the launch stream.  level 1 brackets only the
 * kernels that dominate each path ("fm_decimate", "fm_fifth", "pw_fft"), level 2 every
 * stage ("fm_disc", "fm_deemph", "fm_resample", "fm_droop", "pw_downsample", "pw_rms", ...);
 * 0 switches it off.  Totals are read back with rxgpu_prof_get. */
void rxgpu_prof_enable(int level);
void rxgpu_prof_reset(void);
int rxgpu_prof_get(const char *name, double *total_ms, long *launches);

/* ------------------------------------------------------------- rx_fm: drop-in */

/* Replaces full_demod(d) at rtl_fm.c:923 (definition rtl_fm.c:759-824).
 * In: d->lowpassed[0..d->lp_len) already scaled and rotated by the callback, all
 * parameters and carries in *d.  Out: d->result[0..d->result_len), d->lowpassed[0..lp_len')
 * (decimated IQ), and every carry (lp_len, now_r/now_j/prev_index, pre_r/pre_j, lp_*_hist,
 * droop_*_hist, now_lpr/prev_lpr_index) exactly as the CPU leaves them.  deemph_filter's
 * function-static `avg` (rtl_fm.c:669) has no field in the struct: it lives in a side-car
 * keyed by the demod_state address (rxgpu_deemph_state).  Covered: -M fm|am|usb|lsb|raw,
 * -A std|fast|lut|ale, -l squelch, -F 0|9, -E deemp|adc|rdc (rdc runs in rxgpu_callback), -o, -r.
 * Any block length the reference takes: readStream may return any element count (rtl_fm.c:894-899), so lp_len is any even number from 0 to
 * MAXIMUM_BUF_LENGTH -- -F blocks that are not a multiple of 2^passes follow the C's own int16 indexing (lp_len >> i turns odd, the final
 * lp_len may be odd), and a block that completes no decimated sample gets what the C does on the struct's memory (fm_demod's result[0], and
 * pre_r/pre_j from lp[lp_len-2], lp[lp_len-1] in front of lowpassed[]).
 * -L level printing (rtl_fm.c:792-807) sits inside full_demod on file-static counters of rtl_fm.c: the library cannot see them, so the
 * block stays with the caller -- behind this call, fed `sr` from rxgpu_dropin_block_rms (dropin/rx_fm_unit.c does exactly that, and the
 * rx_fm built by dropin/Makefile prints the reference's level lines).
 * -o with a block whose demodulated length is no multiple of the step: low_pass_simple's last loop turn then sums past `len` (rtl_fm.c:373-387) but
 * stores that sum BEHIND the len / step values it returns -- the block hands on its complete groups, the remainder is dropped; reproduced.
 * Not on the device path, by decision: the two shapes the reference itself dies on (-E adc with no demodulated sample: division by
 * result_len == 0, rtl_fm.c:693; -E rdc on an empty read, rtl_fm.c:711): those print to stderr and exit -- there is no CPU fallback.
 * If the block in d->lowpassed is the one rxgpu_callback handed over last (same demod_state, same lp_len, nobody
 * called rxgpu_dropin_invalidate), the copy it left in HBM is used and the block does not cross PCIe a second time.
 * A block of the plain FM chain (low_pass with downsample >= 8, fm_demod -A std | fast | ale, deemph_filter, low_pass_real; no -F, squelch,
 * dc block, -o) takes two launches and no copy operation (~40 us per 1 MiB); every other shape the general stream path (~105 us).
 * Same results either way ($RXGPU_DROPIN_FAST=0 takes the general path always). */
void rxgpu_full_demod(struct demod_state *d);
/* What the `void` drop-in entry points do on a device error, for callers (the files under dropin/) that want the same end: one line on stderr ("rxgpu: <what>:
 * <rxgpu_last_error()>"; never stdout -- that is the audio / CSV stream), the device drained of what is in flight (at most five seconds: a
 * watchdog ends the process if the device no longer answers; nothing is freed under the application's other threads), then _exit(1) -- NOT exit(): no atexit handlers or static destructors of a SoapySDR
 * driver run while the application's other threads are still inside it or hold d->rw.  A second thread that fails meanwhile waits for the first. */
void rxgpu_fatal(const char *what);
/* a caller that edits d->lowpassed between rxgpu_callback and rxgpu_full_demod says so here */
void rxgpu_dropin_invalidate(const struct demod_state *d);
/* full_demod's `sr` (rtl_fm.c:781: rms(d->lowpassed, d->lp_len, 1) of the decimated block, BEFORE a quiet block is zeroed) for the
 * block the last rxgpu_full_demod(d) took, as the squelch kernel computed it; INT_MIN for an empty block ((int)NaN on x86-64).
 * RXGPU_EUNSUPPORTED when that block ran without squelch (squelch_level == 0): lowpassed[] is then intact and the reference's own
 * rms(d->lowpassed, d->lp_len, 1) is the value.  What -L needs (rtl_fm.c:792-807). */
int rxgpu_dropin_block_rms(const struct demod_state *d, int *sr);
/* Forget a demod_state: frees its stream object, device buffers, de-emphasis accumulator and side-car slot (16 exist; the 17th
 * distinct live demod_state makes the drop-ins fail).  For callers that create and destroy demod_states; the reference's own is a
 * global.  RXGPU_EINVAL if `d` has no side-car. */
int rxgpu_dropin_release(const struct demod_state *d);
/* Optional, once at start-up: page-lock lowpassed[]..result[] of *d and buf16[] of *s so that the drop-in's copies are
 * DMA'd in place (SURVEY.md section 8b "Ownership").  The structs must outlive the registration -- the reference's are
 * globals (rtl_fm.c:190-191); rxgpu_dropin_unpin before freeing heap-allocated ones.  Either pointer may be NULL.
 * Exactly the members' bytes are registered (no rounding out to pages), so neighbouring host buffers keep resolving
 * as pageable memory. */
int rxgpu_dropin_pin(struct demod_state *d, struct dongle_state *s);
int rxgpu_dropin_unpin(struct demod_state *d, struct dongle_state *s);

/* Diagnostics: with $RXGPU_DROPIN_TIMING=1 the two drop-ins accumulate host-clock microseconds per phase -- us[0] callback copies + pre-stage
 * kernel, [1] callback hand-off, [2] full_demod set-up, [3] full_demod run (kernels + carries), [4] full_demod result / lowpassed[] copies,
 * [5] callbacks, [6] full_demod calls.  Copies up to n (<= 7) values, clears the table, returns how many. */
int rxgpu_dropin_timing(double *us, int n);

/* Replaces rtlsdr_callback(buf, len, ctx) at rtl_fm.c:899 (definition 828-863):
 * mute-zero, CS16 -> 8-bit-range scale, rotate16_90 unless offset tuning, hand-off into
 * s->demod_target->lowpassed under d->rw, signal d->ready.  len = int16 count.
 * With buf16[] page-locked (rxgpu_dropin_pin) AND the whole read buffer `buf` (MAXIMUM_BUF_LENGTH int16, rtl_fm.c:871-873) page-locked
 * with rxgpu_pin, the block crosses PCIe inside one launch that reads `buf` and writes buf16[] (30 instead of 47-55 us per 1 MiB);
 * otherwise an H2D copy, the kernel and a D2H copy.  Same bytes either way ($RXGPU_DROPIN_ZC=0 keeps the copies). */
void rxgpu_callback(int16_t *buf, uint32_t len, void *ctx);

/* full_demod dispatches on d->mode_demod, a pointer into the reference's own code (rtl_fm.c:154, 808-809).
 * The drop-in cannot know those addresses: tell it once, e.g.
 *   rxgpu_set_demod_functions(&fm_demod, &am_demod, &usb_demod, &lsb_demod, &raw_demod);
 * Without this call every demod_state is taken to be in fm mode. */
void rxgpu_set_demod_functions(void *fm, void *am, void *usb, void *lsb, void *raw);

/* side-car for the de-emphasis accumulator of a given demod_state */
int *rxgpu_deemph_state(const struct demod_state *d);

/* -------------------------------------------------------- rx_fm: batched stream */

/* Parameters of the chain: the demod_state fields full_demod reads (rtl_fm.c:136-151)
 * plus dongle_state.offset_tuning (rtl_fm.c:854). */
typedef struct rxgpu_fm_params {
	int downsample;          /* low_pass boxcar length when downsample_passes == 0 */
	int downsample_passes;   /* > 0: fifth_order cascade (the -F path) */
	int comp_fir_size;       /* 9: cic_9_tables droop compensation after the cascade */
	int custom_atan;         /* -A: 0 std (libm atan2), 1 fast (fast_atan2), 2 lut (polar_disc_lut), 3 ale (esbensen) */
	int deemph;              /* deemph_filter on/off */
	int deemph_a;
	int rate_out;            /* low_pass_real input rate */
	int rate_out2;           /* low_pass_real output rate; <= 0 disables it */
	int offset_tuning;       /* != 0: no rotate16_90 */
	int prescaled;           /* != 0: input is already lowpassed[] (skip scale + rotate) */
	int mode;                /* -M: RXGPU_MODE_FM/AM/USB/LSB/RAW = fm_demod/am_demod/usb_demod/lsb_demod/raw_demod (rtl_fm.c:584-665) */
	int output_scale;        /* demod_state.output_scale (rtl_fm.c:988-992); 0 is taken as 1 */
	int squelch_level;       /* -l: power squelch on the decimated block (rtl_fm.c:781-790); 0 = off */
	int dc_block_audio;      /* -E adc: dc_block_audio_filter (rtl_fm.c:684-697, 818) */
	int adc_block_const;     /* its averaging constant (default 9, rtl_fm.c:1106) */
	int post_downsample;     /* -o: low_pass_simple on the demodulated block (rtl_fm.c:373-387, 814-815); 0/1 = off.
	                          * Every block's demodulated length must be a multiple of it (the reference's own
	                          * precondition; it reads stale data otherwise) -- RXGPU_EUNSUPPORTED if not */
	int dc_block_raw;        /* -E rdc: dc_block_raw_filter on the scaled capture, in the callback (rtl_fm.c:699-721, 850-852) */
	int rdc_block_const;     /* its averaging constant (default 9, rtl_fm.c:1110) */
} rxgpu_fm_params;

enum { RXGPU_MODE_FM = 0, RXGPU_MODE_AM = 1, RXGPU_MODE_USB = 2, RXGPU_MODE_LSB = 3, RXGPU_MODE_RAW = 4 };

/* Parameter derivation, host only (no device needed): what rx_fm's main() does between getopt and the first block.
 *   rxgpu_fm_params_init   demod_init's defaults (rtl_fm.c:1086-1115) + the -M switch (1320-1341); mode is one of
 *                          "fm" "nbfm" "nfm" "wbfm" "wfm" "am" "usb" "lsb" "raw" "iq"; *rate_in receives demod.rate_in
 *                          (24000, or 170000 for wbfm); downsample_passes stays 0 (set it to 1, like -F does at 1306,
 *                          with comp_fir_size = the -F argument, before planning)
 *   rxgpu_fm_plan_settings `rate_in *= post_downsample` (1371), optimal_settings (960-997: downsample,
 *                          downsample_passes, capture_freq/rate, output_scale) and deemph_a (1410-1415);
 *                          reads p->downsample_passes (truthy = -F), offset_tuning, mode, deemph, rate_out,
 *                          post_downsample; writes p->downsample, downsample_passes, output_scale, deemph_a and *plan */
typedef struct rxgpu_fm_plan {
	int rate_in;                 /* after the post_downsample multiplication */
	int downsample, downsample_passes, output_scale, deemph_a;
	uint32_t capture_freq, capture_rate;   /* dongle.freq / dongle.rate */
} rxgpu_fm_plan;
int rxgpu_fm_params_init(rxgpu_fm_params *p, const char *mode, int *rate_in);
int rxgpu_fm_plan_settings(rxgpu_fm_params *p, int freq, int rate_in, int edge, int time_constant_us, rxgpu_fm_plan *plan);

/* Every value the chain carries from one call to the next (SURVEY.md section 8b contract) */
typedef struct rxgpu_fm_carry {
	int now_r, now_j, prev_index;           /* low_pass          rtl_fm.c:139,141 */
	int pre_r, pre_j;                        /* fm_demod          rtl_fm.c:140 */
	int16_t lp_i_hist[10][6], lp_q_hist[10][6];   /* fifth_order  rtl_fm.c:130-131 */
	int16_t droop_i_hist[9], droop_q_hist[9];     /* generic_fir  rtl_fm.c:133-134 */
	int deemph_avg;                          /* deemph_filter's static, rtl_fm.c:669 */
	int now_lpr, prev_lpr_index;             /* low_pass_real     rtl_fm.c:150-151 */
	int squelch_hits;                        /* power squelch     rtl_fm.c:145 */
	int dc_avg;                              /* dc_block_audio    rtl_fm.c:152 */
	int dc_avgI, dc_avgQ;                    /* dc_block_raw      rtl_fm.c:153 */
} rxgpu_fm_carry;

typedef struct rxgpu_fm_stream rxgpu_fm_stream;

/* Workspace for up to max_blocks blocks of block_len int16 (I,Q interleaved) per run. */
int rxgpu_fm_stream_create(rxgpu_fm_stream **out, const rxgpu_fm_params *params,
                           size_t max_blocks, size_t block_len);
void rxgpu_fm_stream_destroy(rxgpu_fm_stream *s);
int rxgpu_fm_stream_set_carry(rxgpu_fm_stream *s, const rxgpu_fm_carry *c);
int rxgpu_fm_stream_get_carry(rxgpu_fm_stream *s, rxgpu_fm_carry *c);

/* n_blocks consecutive callback blocks through rtlsdr_callback's pre-stage + full_demod,
 * with exactly the per-block semantics of the reference (rotation phase restart, libm
 * discriminator on each block's first sample, fifth_order seam rule, carries).
 * block_len: any even int16 count >= 2.  -F blocks that are not a multiple of 2^passes take the per-block kernels that index like the C
 * (slow path); RXGPU_EUNSUPPORTED only where the reference reads struct memory the stream does not have: -F blocks that leave fewer than
 * two int16 after the cascade, low_pass blocks shorter than downsample (the drop-in handles both on the real struct).
 * d_iq : DEVICE pointer, n_blocks * block_len int16, resident in HBM.
 * d_out: DEVICE pointer, capacity out_cap int16; receives the concatenated result[] of all
 *        blocks.  *out_len = int16 written.  block_out_len (HOST, optional, n_blocks ints)
 *        = result_len of each block.  Synchronous: returns after the device finished and
 *        the carries were read back. */
int rxgpu_fm_stream_run(rxgpu_fm_stream *s, const int16_t *d_iq, size_t n_blocks, size_t block_len,
                        int16_t *d_out, size_t out_cap, size_t *out_len, int *block_out_len);

/* Pipelined form: enqueue and return.  The HBM-bound decimator of run r+1 overlaps the
 * latency-bound audio stages of run r on a second stream; carries are chained on the device.
 * *out_len and block_out_len are filled immediately (they are closed-form in the geometry).
 * rxgpu_fm_stream_wait() blocks until every enqueued run has finished and brings the carries
 * back; get_carry/set_carry/run wait implicitly.  At most two runs are in flight: enqueueing run
 * r+2 first retires run r (normally long finished).  d_iq and d_out of a run must stay valid
 * until it is retired.  A libm-discriminator sample the device could not decide (see host_fixups:
 * about 2e-10 of the libm samples) is settled when its run is retired: the host re-evaluates it
 * with its own libm, patches the demodulated sample and redoes the audio stages of that run and
 * of the one behind it -- the sequence continues, nothing is rolled back. */
int rxgpu_fm_stream_run_async(rxgpu_fm_stream *s, const int16_t *d_iq, size_t n_blocks, size_t block_len,
                              int16_t *d_out, size_t out_cap, size_t *out_len, int *block_out_len);
int rxgpu_fm_stream_wait(rxgpu_fm_stream *s);

/* Same with HOST input/output buffers: the capture crosses PCIe in chunks of whole blocks into two device
 * staging buffers on a copy stream while the previous chunk is being demodulated (double-buffered; pin the
 * buffers with rxgpu_pin for DMA without a bounce), results come back the same way.  Synchronous. */
int rxgpu_fm_stream_run_host(rxgpu_fm_stream *s, const int16_t *h_iq, size_t n_blocks, size_t block_len,
                             int16_t *h_out, size_t out_cap, size_t *out_len, int *block_out_len);

/* Number of libm-discriminator samples re-evaluated on the host since the last wait()/run() began
 * (device fp64 atan2 result within 2^-33 of a truncation boundary, where a last-ulp difference to
 * glibc's atan2 could matter). */
long rxgpu_fm_stream_host_fixups(const rxgpu_fm_stream *s);

/* ------------------------------------------------- rx_fm: channeliser (extension)
 *
 * BASELINE configs[4] / SURVEY.md section 8(f) rank 2.  NOT a reference feature: rx_tools has a single
 * demod_state ("multiple of these, eventually", rtl_fm.c:189) and no mixer beyond rotate16_90.  It is
 * specified from reference primitives only, so that its oracle is made of functions already pinned against
 * the reference: every window of N = 2^bin_e capture samples goes through fix_fft (rtl_power.c:264-320) --
 * the bank of "mix by k*fs/N, boxcar-sum N samples" channels, i.e. low_pass (rtl_fm.c:351-371) at ds = N for
 * every offset at once, in the reference's fixed-point scaling -- and bin first_bin+c of successive windows
 * is channel c's lowpassed[] stream, demodulated by fm_demod (rtl_fm.c:584-615; each callback block's
 * first sample through libm atan2, the rest per custom_atan) with per-channel carried pre_r/pre_j.
 * Channel spacing = channel sample rate = fs/N (20 Msps, N=1024: 19.5 kHz NBFM channels).
 * Channel response: a rectangular N-sample window, critically sampled -- each channel is the boxcar low_pass of the reference, so its
 * selectivity is a sinc's (first side lobe -13 dB, nulls at the neighbouring channel centres): adjacent-channel energy away from those
 * centres leaks in, exactly as it does into rx_fm's own low_pass at ds = N.  There is no windowed / polyphase prototype filter here,
 * because the reference has none to pin one against. */
typedef struct rxgpu_chan_params {
	int bin_e;               /* window length 2^bin_e complex samples (1..15) */
	int first_bin;           /* channel c = FFT bin (first_bin + c) mod N */
	int n_channels;
	int custom_atan;         /* 0 std, 1 fast */
	/* per-channel audio stages, every channel with carried state of its own like a demod_state (NBFM defaults: both off,
	 * rtl_fm.c:1086-1099): deemph_filter (rtl_fm.c:667-682) and low_pass_real (389-409) on the channel's demodulated stream */
	int deemph, deemph_a;    /* -E deemp; deemph_a as rxgpu_fm_plan_settings derives it for the channel rate fs / N */
	int rate_out, rate_out2; /* low_pass_real: channel rate (fs / N) -> rate_out2; rate_out2 <= 0 disables it */
	/* 0: the bank of fix_fft bins above (default).  1: SURVEY 8(f)2's literal definition -- the capture through the callback's scale
	 * (rtl_fm.c:845-848, no rotation), per channel an integer NCO at k * fs / N (cos / sin from the reference's Sinewave table, each product
	 * rounded by FIX_MPY, rtl_power.c:256-262), then low_pass (rtl_fm.c:351-371) at downsample N; everything behind it as above.  The
	 * same channels in another fixed-point rounding, at ~50 times the arithmetic: windows of 2^3 .. 2^12 samples. */
	int nco;
} rxgpu_chan_params;

typedef struct rxgpu_chan rxgpu_chan;

/* sinewave: rxgpu_sine_table(bin_e).  block_len (int16) must hold a whole number of windows. */
int rxgpu_chan_create(rxgpu_chan **out, const rxgpu_chan_params *p, size_t max_blocks, size_t block_len,
                      const int16_t *sinewave);
void rxgpu_chan_destroy(rxgpu_chan *s);
/* pre: n_channels pairs (pre_r, pre_j) */
int rxgpu_chan_set_carry(rxgpu_chan *s, const int *pre);
int rxgpu_chan_get_carry(rxgpu_chan *s, int *pre);
/* audio: n_channels triples (deemph avg, now_lpr, prev_lpr_index) -- rtl_fm.c:669, 150-151 per channel */
int rxgpu_chan_set_audio_carry(rxgpu_chan *s, const int *audio);
int rxgpu_chan_get_audio_carry(rxgpu_chan *s, int *audio);
/* d_iq: DEVICE, n_blocks * block_len int16.  d_out: DEVICE, [n_channels][out_stride] int16; channel c's
 * output at d_out[c*out_stride .. + *windows_out): one demodulated sample per window, or -- with low_pass_real on -- the
 * resampled audio (*windows_out = samples per channel, the same for every channel).  Synchronous. */
int rxgpu_chan_run(rxgpu_chan *s, const int16_t *d_iq, size_t n_blocks, size_t block_len, int16_t *d_out,
                   size_t out_stride, size_t *windows_out);
/* The same without waiting: up to two runs in flight on the library's stream, the carries chained from run to run on the device (a channel's
 * carry is its last bin, which no host fix-up changes), so that the host's part of run r -- reading the flag count, settling an undecided libm
 * sample -- happens while run r + 1 computes.  rxgpu_chan_wait retires what is in flight (older run first), brings the carries home and
 * returns the windows per channel of the LAST run; rxgpu_chan_get_carry / _set_carry / _host_fixups refer to retired runs only.  d_iq and
 * d_out of a run must stay untouched until it is retired (the wait, or the second rxgpu_chan_run_async after it).  With the per-channel audio
 * stages on (deemph / rate_out2) a run is finished before the call returns, like rxgpu_chan_run. */
int rxgpu_chan_run_async(rxgpu_chan *s, const int16_t *d_iq, size_t n_blocks, size_t block_len, int16_t *d_out, size_t out_stride);
int rxgpu_chan_wait(rxgpu_chan *s, size_t *windows_out);
/* undecided libm samples the host settled in the runs the last rxgpu_chan_run / rxgpu_chan_wait retired */
long rxgpu_chan_host_fixups(const rxgpu_chan *s);

/* --------------------------------------------------------- rx_power: drop-in */

/* Replaces scanner(channel)'s per-tune compute at rtl_power.c:709-770 for tunes whose
 * buf16 the caller has already filled (the device I/O part of scanner(), 683-703, stays with
 * the caller): for every tune, ts->avg[] += / MAX= and ts->samples += exactly as the CPU.
 * Globals of the reference are passed explicitly: window_coefs (rtl_power.c:87,1034-1037),
 * Sinewave (82,240-254; 3/4 * 2^bin_e entries), boxcar/comp_fir_size/peak_hold (115-117).
 * By default the call ends with the merge: the structs are current when it returns and the library keeps no pointer of the
 * caller's past the call.
 *
 * rxgpu_scan_deferred(1) (or $RXGPU_SCAN_DEFERRED=1 at the first scan) -- what the INTEGRATION.md patch switches on, because the
 * reference reads avg[] only in csv_dbm, once per report interval (rtl_power.c:1045-1050): the sums then stay ON THE DEVICE between
 * calls -- a call uploads the tunes' buf16, adds the sweep to device-resident accumulators and returns without waiting -- and
 * ts->avg[] / ts->samples are brought up to date by rxgpu_scan_sync(tunes, n), which rxgpu_csv_dbm calls by itself for a
 * tuning_state of the sweep; call it explicitly in front of the reference's own csv_dbm or any other reader of the struct.
 * LIFETIME in deferred mode: `tunes` must stay allocated until rxgpu_scan_sync has returned -- sync BEFORE freeing or replacing the
 * array.  The library never writes through a pointer that was not handed to the running call: a pending interval that meets
 * another array, count or geometry makes rxgpu_scan fail (RXGPU_EINVAL, "sync first"), one still pending at rxgpu_shutdown is
 * dropped with a line on stderr. */
int rxgpu_scan(struct tuning_state *tunes, int tune_count, const int *window_coefs,
               const int16_t *sinewave, int boxcar, int comp_fir_size, int peak_hold);
/* deferred mode: merge what rxgpu_scan accumulated since the last sync into tunes[i].avg[] (+=, MAX with peak hold) and
 * tunes[i].samples.  `tunes`/`tune_count` must be the array of the pending sweep (RXGPU_EINVAL otherwise, nothing is written).
 * A no-op when nothing is pending (and always in the default mode). */
int rxgpu_scan_sync(struct tuning_state *tunes, int tune_count);
/* 1: leave the sums on the device between rxgpu_scan calls; 0 (default): merge at the end of every call.  Switching off with an
 * interval pending is RXGPU_EINVAL. */
int rxgpu_scan_deferred(int on);
long rxgpu_scan_syncs(void);    /* downloads made so far (diagnostics / tests) */
/* Diagnostics: with $RXGPU_DROPIN_TIMING=1 rxgpu_scan / rxgpu_scan_sync accumulate host-clock microseconds per phase -- us[0] scan: geometry check +
 * the table of page-locked rows, [1] scan: gather launch (or staging), [2] scan: enqueue of the transforms, [3] scan: wait until the caller's buffers
 * have been read, [4] sync: wait for the accumulators' D2H, [5] sync: merge into avg[] / samples, [6] scans, [7] syncs.  Copies up to n (<= 8)
 * values, clears the table, returns how many. */
int rxgpu_scan_timing(double *us, int n);
/* INPUT of rxgpu_scan: scanner() copies every tune's buf16 into fft_buf (rtl_power.c:715-720).  Here the tunes' buf16 -- malloc'd once by
 * frequency_range and never freed (rtl_power.c:518-531) -- are page-locked in place the first time a sweep geometry sees them (exactly the
 * buf_len int16 scanner() reads; buffers the caller page-locked itself with rxgpu_pin are used as they are) and ONE launch pulls all of them
 * across PCIe into the scan's contiguous input: no host memcpy, no staging copy.  The call returns when that launch has read the buffers (the
 * caller refills them at once), not when the transforms are done.  The registrations are released with the sweep geometry
 * (rxgpu_scan_release / rxgpu_shutdown / another geometry).  LIFETIME: a buf16 that has been handed to rxgpu_scan must stay allocated
 * until one of those -- freeing or reallocating it under its registration leaves pinned pages behind that a new allocation at the same
 * address would alias (the gather would read the OLD pages, no error).  A call on a sub-array of the registered sweep
 * (rxgpu_scan(&tunes[i], j - i): the drop-in's missed-read path) gathers through the rows the table already has; a shorter call on buffers
 * the table does not know is staged; neither disturbs the full sweep's registrations.  A buffer that cannot be page-locked, a buffer that cannot be page-locked, a row that is no multiple of 16 bytes,
 * or $RXGPU_SCAN_ZC=0 select the older path (gather into pinned staging by memcpy, one H2D).  1 if the last rxgpu_scan read zero-copy. */
int rxgpu_scan_zero_copy(void);
/* OUTPUT of rxgpu_scan_sync (and of every rxgpu_scan in the default mode): the tunes' avg[] -- malloc'd once by frequency_range like buf16, the same
 * LIFETIME rule -- are page-locked in place the first time they are merged into, and ONE launch adds (peak hold: maxes) the device's accumulators into
 * them across PCIe; the host adds the sample counts.  The same conditions select the older path (D2H of the accumulators, additions on the calling
 * thread).  1 if the last merge ran in place. */
int rxgpu_scan_sync_in_place(void);
/* csv_dbm ends by zeroing the row it printed (rtl_power.c:815-817).  rxgpu_csv_dbm does the same and remembers it: the next in-place merge into a sweep
 * whose rows are ALL known to hold zeros writes the accumulators over them instead of reading, adding and writing back -- 19.6 MB one way across
 * the link at the configs[2] geometry, not both ways.  A caller that keeps the reference's own csv_dbm says so with this call after its loop over the
 * tunes; it is a promise that avg[] of tunes[0 .. tune_count) hold zeros NOW and are written by nothing but this library until the
 * next merge.  Rows the library has not page-locked (the copying path) are ignored.  Returns the number of rows marked. */
int rxgpu_scan_rows_cleared(const struct tuning_state *tunes, int tune_count);
/* Forget the cached sweep geometry of rxgpu_scan: its scan object, device buffers and the page-lock registrations of the tunes' buf16.
 * For callers that free or replace their tune buffers (the reference never does); call it BEFORE freeing them.  A pending deferred
 * interval is dropped with a line on stderr (rxgpu_scan_sync first).  rxgpu_shutdown does the same. */
void rxgpu_scan_release(void);

/* csv_dbm(ts) (rtl_power.c:774-817) writing to `file`.  Host code, NOT a device function: our restatement of the
 * reference's text formatter (one index map per printed bin), tested byte-for-byte against the reference's output.
 * Optional -- fed the bit-exact avg[] of rxgpu_scan, the reference's own csv_dbm prints the same bytes. */
void rxgpu_csv_dbm(struct tuning_state *ts, void *file /* FILE* */);

/* Host-side planners/tables with the reference's exact arithmetic (needed to size shards):
 * frequency_range (rtl_power.c:431-543), sine_table (240-254), window tables (322-401). */
typedef struct rxgpu_power_plan {
	int tune_count, bin_e, buf_len, downsample, downsample_passes, rate;
	int64_t first_freq, bw_seen;
	double crop;
} rxgpu_power_plan;
int rxgpu_power_plan_range(const char *range, double crop, int boxcar, rxgpu_power_plan *plan);
int rxgpu_sine_table(int log2n, int16_t *sinewave /* 3n/4 entries */);
int rxgpu_window_coefs(const char *name, int length, int *coefs);

/* -------------------------------------------------------- rx_power: batched scan */

typedef struct rxgpu_power_params {
	int bin_e, buf_len, downsample, downsample_passes;
	int boxcar, comp_fir_size, peak_hold;
} rxgpu_power_params;

typedef struct rxgpu_power_scan rxgpu_power_scan;

int rxgpu_power_scan_create(rxgpu_power_scan **out, const rxgpu_power_params *p, int max_tunes,
                            const int *window_coefs, const int16_t *sinewave);
void rxgpu_power_scan_destroy(rxgpu_power_scan *s);

/* `passes` scanner() passes over `tunes` tunes.
 * d_in  : DEVICE, [passes][tunes][buf_len] int16.
 * d_avg : DEVICE, [tunes][1<<bin_e] int64, accumulated into (+= or MAX with peak_hold).
 * d_samples: DEVICE, [tunes] int32, accumulated into.
 * Asynchronous on rxgpu_stream(); call rxgpu_sync() (or a hipStreamSynchronize) to wait. */
int rxgpu_power_scan_run(rxgpu_power_scan *s, const int16_t *d_in, int passes, int tunes,
                         int64_t *d_avg, int32_t *d_samples);

/* ------------------------------------------------------------ rx_power: tunes sharded over the GPUs of one node
 *
 * scanner()'s tunes are independent (rtl_power.c:679-771); rows only meet when main() prints them in tune order
 * (rtl_power.c:1047-1050).  One process per GPU; rank r scans the contiguous range rxgpu_shard_tunes gives it and ONE
 * collective launch per report interval (RCCL over xGMI, enqueued on rxgpu_stream() behind the scan: the ncclGather of the
 * [per][N] int64 avg block and the ncclGather of the [per] int32 samples inside one ncclGroup) brings every rank's rows to
 * the root, which feeds csv_dbm.  librccl is bound at run time
 * ($RXGPU_RCCL_LIB, else an already loaded librccl, else the loader path, else /opt/rocm/lib). */
typedef struct rxgpu_comm rxgpu_comm;
/* 128 bytes (ncclUniqueId): made on one rank, handed to the others by whatever launched the processes (MPI, a file,
 * torch.distributed's store ...), then every rank calls rxgpu_comm_create -- collective, like ncclCommInitRank */
int rxgpu_comm_unique_id(void *id128);
int rxgpu_comm_create(rxgpu_comm **out, const void *id128, int rank, int world);
/* or wrap an ncclComm_t the application already has (not destroyed by rxgpu_comm_destroy).
 * Both check rank/world against what the communicator itself reports (ncclCommUserRank / ncclCommCount): RXGPU_EINVAL if
 * the caller's sharding would not match the communicator's. */
int rxgpu_comm_adopt(rxgpu_comm **out, void *nccl_comm, int rank, int world);
void rxgpu_comm_destroy(rxgpu_comm *c);
int rxgpu_comm_rank(const rxgpu_comm *c);
int rxgpu_comm_world(const rxgpu_comm *c);
long rxgpu_comm_gathers(const rxgpu_comm *c);   /* grouped gathers enqueued on this communicator so far */
const char *rxgpu_comm_library(void);     /* which librccl was bound (NULL: none found) */
/* rank's tunes: [*first, *first + *count), *per = ceil(total / world) = rows of every rank's padded block */
int rxgpu_shard_tunes(int rank, int world, int total, int *first, int *count, int *per);
/* d_avg_local [per][n_bins] int64, d_samples_local [per] int32 (DEVICE) -> on the root d_avg_all [world][per][n_bins],
 * d_samples_all [world][per] (ignored elsewhere).  Asynchronous on rxgpu_stream().  c == NULL: a single process.
 * per == 0 (a sweep of no tunes) is a no-op. */
int rxgpu_power_gather(rxgpu_comm *c, const int64_t *d_avg_local, const int32_t *d_samples_local, int per, int n_bins,
                       int64_t *d_avg_all, int32_t *d_samples_all, int root);
/* rxgpu_power_scan_run over this rank's tunes (d_in_local: [passes][count][buf_len]) followed by the gather; the padding
 * rows [count, per) of a short last rank are zeroed here, so the root never sees what the caller left in them.
 * Asynchronous: the scan on rxgpu_stream(), the gather behind it on the library's copy stream, so that the NEXT interval's scan
 * overlaps this interval's transfer -- alternate two sets of local buffers to use that (a call that finds its send buffers still
 * being gathered makes its scan wait).  The root reads d_avg_all / d_samples_all after rxgpu_sync(), which covers both streams. */
int rxgpu_power_scan_run_sharded(rxgpu_power_scan *s, rxgpu_comm *c, const int16_t *d_in_local, int passes, int total_tunes,
                                 int64_t *d_avg_local, int32_t *d_samples_local, int n_bins,
                                 int64_t *d_avg_all, int32_t *d_samples_all, int root);

/* ------------------------------------------------------------------ rx_sdr output formats
 *
 * rx_sdr reads CS16 (or CS12) from the device and writes the stream in the format asked with -F; the
 * conversions are inline loops in its main() (rtl_sdr.c:354-391).  They are replaced by one call per read:
 *
 *   RXGPU_SDR_CS16_TO_CU8    buf8[i] = (uint8)(x / 32767.0 * 128.0 + 127.4)     rtl_sdr.c:376-378
 *   RXGPU_SDR_CS16_TO_CS8    buf8[i] = (int8) (x / 32767.0 * 128.0 + 0.4)       rtl_sdr.c:368-370
 *                            (x >= 32665 gives 128, stored as -128 like the reference build)
 *   RXGPU_SDR_CS16_TO_CF32   fbuf[i] = x * 1.0f / SHRT_MAX                        rtl_sdr.c:384-386
 *   RXGPU_SDR_CS12_TO_CS16   3 packed bytes -> (b1<<12)|(b0<<4), (b2<<8)|(b1&0xf0) rtl_sdr.c:356-363
 *
 * n_elems counts complex elements (one I/Q pair), like readStream's return value. */
enum {
	RXGPU_SDR_CS16_TO_CU8 = 0,
	RXGPU_SDR_CS16_TO_CS8 = 1,
	RXGPU_SDR_CS16_TO_CF32 = 2,
	RXGPU_SDR_CS12_TO_CS16 = 3
};
/* bytes read / written for n_elems elements of a conversion (0 for an unknown conversion) */
size_t rxgpu_sdr_in_bytes(int conversion, size_t n_elems);
size_t rxgpu_sdr_out_bytes(int conversion, size_t n_elems);
/* device pointers (16-byte aligned), asynchronous on rxgpu_stream() */
int rxgpu_sdr_convert(int conversion, const void *d_in, size_t n_elems, void *d_out);
/* host pointers: what rx_sdr's read loop calls instead of its for-loops; synchronous.
 * Replaces rtl_sdr.c:354-391 (the fwrite that follows stays). */
int rxgpu_sdr_convert_host(int conversion, const void *in, size_t n_elems, void *out);

/* rx_fm -E wav: the 44 header bytes generate_header() writes (rtl_fm.c:1174-1206); raw_mode != 0 is the
 * two-channel header used with -M raw.  Host only. */
void rxgpu_wav_header(int rate, int raw_mode, unsigned char out[44]);

#ifdef __cplusplus
}
#endif
#endif
