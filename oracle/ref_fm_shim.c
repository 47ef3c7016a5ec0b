/* oracle/_ref/libref_fm.so -- TEST INFRASTRUCTURE ONLY.
 *
 * Wraps the reference's own rx_fm translation unit, compiled where it lies
 * (REF_RTL_FM_C = "/root/reference/src/rtl_fm.c", passed by oracle/Makefile),
 * UNMODIFIED, by #including it so that its file-static functions
 * (rtlsdr_callback rtl_fm.c:828, optimal_settings rtl_fm.c:960) and globals
 * (demod, dongle, output, controller rtl_fm.c:190-193) are reachable.
 * No reference source is copied into this repository; only the accessors below are
 * ours.  Everything else exported by the resulting .so (full_demod, low_pass,
 * fifth_order, fm_demod, polar_disc_fast, fast_atan2, deemph_filter, low_pass_real,
 * rotate16_90, generic_fir, ...) is the reference's code with external linkage.
 */
#define main rx_fm_main
#include REF_RTL_FM_C
#undef main
#include <stddef.h>

size_t ref_fm_sizeof_demod_state(void)  { return sizeof(struct demod_state); }
size_t ref_fm_sizeof_dongle_state(void) { return sizeof(struct dongle_state); }
size_t ref_fm_offsetof(int which)
{
	switch (which) {
	case 0: return offsetof(struct demod_state, lowpassed);
	case 1: return offsetof(struct demod_state, lp_len);
	case 2: return offsetof(struct demod_state, lp_i_hist);
	case 3: return offsetof(struct demod_state, result);
	case 4: return offsetof(struct demod_state, result_len);
	case 5: return offsetof(struct demod_state, rate_in);
	case 6: return offsetof(struct demod_state, now_r);
	case 7: return offsetof(struct demod_state, downsample);
	case 8: return offsetof(struct demod_state, deemph);
	case 9: return offsetof(struct demod_state, now_lpr);
	case 10: return offsetof(struct demod_state, mode_demod);
	case 11: return offsetof(struct demod_state, rw);
	case 12: return offsetof(struct demod_state, output_target);
	case 13: return offsetof(struct demod_state, droop_i_hist);
	case 14: return offsetof(struct demod_state, dc_block_audio);
	case 15: return offsetof(struct dongle_state, buf16);
	case 16: return offsetof(struct dongle_state, mute);
	case 17: return offsetof(struct dongle_state, demod_target);
	case 18: return offsetof(struct dongle_state, offset_tuning);
	}
	return (size_t)-1;
}

/* the reference's own global instances (rtl_fm.c:190-191) */
struct demod_state  *ref_fm_demod(void)  { return &demod; }
struct dongle_state *ref_fm_dongle(void) { return &dongle; }

/* same initialisation order as the reference's main() (rtl_fm.c:1218-1221) */
void ref_fm_init(void)
{
	dongle_init(&dongle);
	demod_init(&demod);
	output_init(&output);
	controller_init(&controller);
}

/* rtlsdr_callback is static (rtl_fm.c:828): expose it as-is */
void ref_fm_callback(int16_t *buf, uint32_t len, struct dongle_state *s) { rtlsdr_callback(buf, len, s); }

/* optimal_settings is static (rtl_fm.c:960) and works on the globals */
void ref_fm_optimal_settings(int freq, int rate) { optimal_settings(freq, rate); }

/* mode_demod targets, for pointer-identity set-up from ctypes */
void *ref_fm_fn(int which)
{
	switch (which) {
	case 0: return (void *)&fm_demod;
	case 1: return (void *)&raw_demod;
	case 2: return (void *)&am_demod;
	case 3: return (void *)&usb_demod;
	case 4: return (void *)&lsb_demod;
	}
	return NULL;
}

/* The de-emphasis state is a function-static int (rtl_fm.c:669) with no reset
 * entry.  Drive it to a requested value using only the reference's own
 * deemph_filter(): each call moves avg towards the fed sample by round(d/a). */
int ref_fm_deemph_force(int target)
{
	static struct demod_state tmp;   /* 1 MiB: keep off the stack */
	int avg, guard = 0;
	tmp.deemph_a = 2;
	tmp.result_len = 1;
	tmp.result[0] = 0;
	deemph_filter(&tmp);             /* learn the current value (exact while |avg| < 32768) */
	avg = tmp.result[0];
	while (avg != target && guard++ < 100000) {
		long want = (long)avg + 2L * ((long)target - avg);   /* a=2: avg += round(d/2) */
		if (want > 32767) want = 32767;
		if (want < -32768) want = -32768;
		tmp.result[0] = (int16_t)want;
		deemph_filter(&tmp);
		avg = tmp.result[0];
	}
	return avg;
}

/* single-threaded timing loop for the CPU baseline: callback + full_demod over
 * n_blocks blocks of block_len int16 taken round-robin from iq; returns the number of
 * int16 results produced (so the work cannot be optimised away). */
long ref_fm_run_blocks(const int16_t *iq, size_t n_blocks_in_buf, size_t block_len, size_t n_calls,
                       int16_t *scratch, int16_t *out, size_t out_cap)
{
	long produced = 0;
	size_t pos = 0;
	for (size_t c = 0; c < n_calls; c++) {
		const int16_t *src = iq + (c % n_blocks_in_buf) * block_len;
		memcpy(scratch, src, block_len * sizeof(int16_t));   /* callback may zero 'mute' samples in place */
		rtlsdr_callback(scratch, (uint32_t)block_len, &dongle);
		full_demod(&demod);
		if (out) {
			size_t n = (size_t)demod.result_len;
			if (pos + n > out_cap) n = out_cap - pos;
			memcpy(out + pos, demod.result, n * sizeof(int16_t));
			pos += n;
		}
		produced += demod.result_len;
	}
	return produced;
}

/* generate_header (rtl_fm.c:1174-1206) into a caller buffer; raw_mode selects mode_demod == &raw_demod */
int ref_fm_wav_header(int rate, int raw_mode, unsigned char *out, size_t cap)
{
	char *mem = NULL;
	size_t len = 0;
	void (*saved)(struct demod_state *) = demod.mode_demod;
	FILE *saved_file = output.file;
	int saved_rate = output.rate;
	output.file = open_memstream(&mem, &len);
	output.rate = rate;
	demod.mode_demod = raw_mode ? &raw_demod : &fm_demod;
	generate_header(&demod, &output);
	fclose(output.file);
	output.file = saved_file;
	output.rate = saved_rate;
	demod.mode_demod = saved;
	if (len > cap) len = cap;
	memcpy(out, mem, len);
	free(mem);
	return (int)len;
}

/* ---- channeliser checker (the channeliser is an extension specified from reference primitives: the decimated stream of a
 * channel is handed to the reference's OWN full_demod, one callback block at a time, on the reference's own global demod_state).
 * The de-emphasis state is one function-static int for the whole process (rtl_fm.c:669), so it is forced to the channel's carried
 * value in front of every call (ref_fm_deemph_force) and read back afterwards with the reference's own deemph_filter on a single
 * zero sample with a = 2^20: |0 - avg| + a/2 < a, the update term truncates to 0 and result[0] is avg, unchanged. */
static int ref_fm_deemph_peek(void)
{
	static struct demod_state tmp;
	tmp.deemph_a = 1 << 20;
	tmp.result_len = 1;
	tmp.result[0] = 0;
	deemph_filter(&tmp);
	return tmp.result[0];
}

/* lp: [n_ch][2 * wpb] int16, channel c's decimated IQ of this block (what low_pass at downsample = 1 passes through unchanged,
 * rtl_fm.c:351-371).  pre: 2 ints per channel (pre_r, pre_j); audio: 3 per channel (deemph avg, now_lpr, prev_lpr_index).
 * out: [n_ch][out_stride]; returns result_len of the last channel (the same for all: equal lengths and phases), or -1. */
static int ref_fm_chan_block_ds(const int16_t *lp, int n_ch, int wpb, int ds, int custom_atan, int deemph, int deemph_a, int rate_out, int rate_out2,
                                int *pre, int *audio, int16_t *out, size_t out_stride);
int ref_fm_chan_block(const int16_t *lp, int n_ch, int wpb, int custom_atan, int deemph, int deemph_a, int rate_out, int rate_out2,
                      int *pre, int *audio, int16_t *out, size_t out_stride)
{
	return ref_fm_chan_block_ds(lp, n_ch, wpb, 1, custom_atan, deemph, deemph_a, rate_out, rate_out2, pre, audio, out, out_stride);
}
/* the NCO mode: lp holds channel c's MIXED stream at the capture rate, [n_ch][2 * wpb * ds] int16; the reference's own low_pass at
 * downsample = ds (rtl_fm.c:351-371) decimates it inside full_demod */
int ref_fm_chan_block_mixed(const int16_t *lp, int n_ch, int wpb, int ds, int custom_atan, int *pre, int16_t *out, size_t out_stride)
{
	static int audio[3 * 4096];
	if (n_ch > 4096 || (size_t)2 * wpb * ds > (size_t)MAXIMUM_BUF_LENGTH)
		return -4;
	memset(audio, 0, sizeof(audio));
	return ref_fm_chan_block_ds(lp, n_ch, wpb, ds, custom_atan, 0, 0, 0, -1, pre, audio, out, out_stride);
}
/* the callback's scale (rtl_fm.c:845-848) without the rotation, through the reference's own rtlsdr_callback: buf -> out, len int16 */
void ref_fm_scale_block(const int16_t *buf, uint32_t len, int16_t *out)
{
	static int16_t tmp[MAXIMUM_BUF_LENGTH];
	const int saved = dongle.offset_tuning, saved_mute = dongle.mute, saved_dc = demod.dc_block_raw;
	memcpy(tmp, buf, (size_t)len * sizeof(int16_t));
	dongle.offset_tuning = 1;                                /* rotate16_90 off (rtl_fm.c:854) */
	dongle.mute = 0;
	demod.dc_block_raw = 0;
	dongle.demod_target = &demod;
	rtlsdr_callback(tmp, len, &dongle);
	memcpy(out, demod.lowpassed, (size_t)len * sizeof(int16_t));
	dongle.offset_tuning = saved;
	dongle.mute = saved_mute;
	demod.dc_block_raw = saved_dc;
}
static int ref_fm_chan_block_ds(const int16_t *lp, int n_ch, int wpb, int ds, int custom_atan, int deemph, int deemph_a, int rate_out, int rate_out2,
                                int *pre, int *audio, int16_t *out, size_t out_stride)
{
	int n_out = -1;
	demod_init(&demod);
	demod.downsample = ds;
	demod.downsample_passes = 0;
	demod.post_downsample = 1;
	demod.squelch_level = 0;
	demod.mode_demod = &fm_demod;
	demod.custom_atan = custom_atan;
	demod.deemph = deemph;
	demod.deemph_a = deemph_a;
	demod.rate_in = demod.rate_out = rate_out;
	demod.rate_out2 = rate_out2;
	demod.dc_block_audio = 0;
	for (int c = 0; c < n_ch; c++) {
		memcpy(demod.lowpassed, lp + (size_t)c * 2 * wpb * ds, (size_t)2 * wpb * ds * sizeof(int16_t));
		demod.lp_len = 2 * wpb * ds;
		demod.now_r = demod.now_j = 0;
		demod.prev_index = 0;
		demod.pre_r = pre[2 * c];
		demod.pre_j = pre[2 * c + 1];
		demod.now_lpr = audio[3 * c + 1];
		demod.prev_lpr_index = audio[3 * c + 2];
		if (deemph && ref_fm_deemph_force(audio[3 * c]) != audio[3 * c])
			return -2;
		full_demod(&demod);
		pre[2 * c] = demod.pre_r;
		pre[2 * c + 1] = demod.pre_j;
		if (deemph)
			audio[3 * c] = ref_fm_deemph_peek();
		audio[3 * c + 1] = demod.now_lpr;
		audio[3 * c + 2] = demod.prev_lpr_index;
		if ((size_t)demod.result_len > out_stride)
			return -3;
		memcpy(out + (size_t)c * out_stride, demod.result, (size_t)demod.result_len * sizeof(int16_t));
		n_out = demod.result_len;
	}
	return n_out;
}
