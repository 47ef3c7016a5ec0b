/* rx_oracle.h -- CPU restatement of the rx_tools sample-stream DSP path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (rx_tools_amd/, librxgpu.so) must not link, import or call anything in oracle/.
 *
 * Parity pinning: the reference ships no tests or golden vectors, so
 * this restatement is pinned against the reference's OWN code compiled unmodified
 * (oracle/_ref, see Makefile) -- directly in tests/test_oracle_vs_ref.py when
 * /root/reference is present, and through tests/golden/ fixtures generated from it by
 * oracle/gen_golden.py everywhere else.
 *
 * Every function cites the reference lines (relative to /root/reference/src) it follows.
 */
#ifndef RX_ORACLE_H
#define RX_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ rx_fm */

#define RXO_MAX_PASSES 10

typedef struct rxo_fm_state {
	/* parameters (demod_state rtl_fm.c:124-159, set by main 1331-1341,1410-1415) */
	int downsample;          /* low_pass decimation (used when downsample_passes == 0) */
	int downsample_passes;   /* >0: fifth_order cascade instead of low_pass */
	int comp_fir_size;       /* 9: droop compensation after the cascade */
	int custom_atan;         /* 0 libm atan2, 1 fast_atan2 */
	int deemph, deemph_a;
	int rate_out, rate_out2; /* low_pass_real ratio; rate_out2 <= 0 disables it */
	int offset_tuning;       /* !=0: skip rotate16_90 (rtl_fm.c:854) */
	int mute;                /* zero this many leading int16 of the next block (839-843) */
	int mode;                /* 0 fm_demod, 1 am_demod, 2 usb_demod, 3 lsb_demod, 4 raw_demod (rtl_fm.c:584-665) */
	int output_scale;        /* rtl_fm.c:988-992 */
	int squelch_level;       /* rtl_fm.c:781-790 */
	int dc_block_audio, adc_block_const;   /* rtl_fm.c:684-697, 818 */
	int post_downsample;     /* -o: low_pass_simple on the demodulated block, rtl_fm.c:814-815 (0/1 = off) */
	int dc_block_raw, rdc_block_const;     /* -E rdc: dc_block_raw_filter in the callback, rtl_fm.c:699-721, 850-852 */
	/* carries */
	int now_r, now_j, prev_index;
	int pre_r, pre_j;
	int16_t lp_i_hist[RXO_MAX_PASSES][6], lp_q_hist[RXO_MAX_PASSES][6];
	int16_t droop_i_hist[9], droop_q_hist[9];
	int deemph_avg;          /* the function-static `avg` of deemph_filter (669) */
	int now_lpr, prev_lpr_index;
	int squelch_hits;
	int dc_avg;
	int dc_avgI, dc_avgQ;    /* dc_block_raw_filter */
} rxo_fm_state;

/* rtl_fm.c:845-848: CS16 -> 8-bit-range int16 through fp64 */
int16_t rxo_scale_sample(int16_t x);
/* rtl_fm.c:309-327 */
void rxo_rotate_90(int16_t *buf, uint32_t len);
/* rtl_fm.c:351-371; returns the new lp_len */
int rxo_low_pass(int16_t *lp, int lp_len, int downsample, int *now_r, int *now_j, int *prev_index);
/* rtl_fm.c:411-440 (stateful, one interleaved half) */
void rxo_fifth_order_fm(int16_t *data, int length, int16_t hist[6]);
/* rtl_fm.c:442-465 */
void rxo_generic_fir_fm(int16_t *data, int length, const int *fir, int16_t hist[9]);
/* rtl_fm.c:288-300 */
const int *rxo_cic9_table(int passes);
/* rtl_fm.c:485-506, 508-513, 476-483 */
int rxo_fast_atan2(int y, int x);
int rxo_polar_disc_fast(int ar, int aj, int br, int bj);
int rxo_polar_discriminant(int ar, int aj, int br, int bj);
/* rtl_fm.c:515-564 (table from atan_lut_init with the same libm), 566-582 */
int rxo_polar_disc_lut(int ar, int aj, int br, int bj);
int rxo_esbensen(int ar, int aj, int br, int bj);
/* rtl_fm.c:739-757 */
int rxo_rms(const int16_t *samples, int len, int step);
/* rtl_fm.c:617-665: mode 1 am, 2 usb, 3 lsb, 4 raw; returns result_len */
int rxo_simple_demod(int mode, const int16_t *lp, int lp_len, int output_scale, int16_t *result);
/* rtl_fm.c:684-697 */
void rxo_dc_block_audio(int16_t *result, int n, int adc_block_const, int *dc_avg);
/* rtl_fm.c:373-387 (len % step == 0); returns the new length */
int rxo_low_pass_simple(int16_t *signal2, int len, int step);
/* rtl_fm.c:699-721: one callback block of len int16 */
void rxo_dc_block_raw(int16_t *buf, int len, int rdc_block_const, int *dc_avgI, int *dc_avgQ);
/* rtl_fm.c:584-615; returns result_len */
int rxo_fm_demod(const int16_t *lp, int lp_len, int custom_atan, int *pre_r, int *pre_j, int16_t *result);
/* rtl_fm.c:667-682 */
void rxo_deemph(int16_t *result, int n, int a, int *avg);
/* rtl_fm.c:389-409; returns the new result_len */
int rxo_low_pass_real(int16_t *result, int n, int rate_out, int rate_out2, int *now_lpr, int *prev_lpr_index);

/* rtl_fm.c:828-863 pre-stage into lp[] (mute, scale, rotate), then full_demod 759-824
 * on the wbfm/fm path.  in: len int16 (I,Q interleaved).  lp: scratch of >= len int16,
 * holds the decimated IQ on return (*lp_len_out int16).  out: >= len/2 int16.
 * Returns result_len. */
int rxo_fm_block(rxo_fm_state *st, const int16_t *in, int len, int16_t *lp, int *lp_len_out, int16_t *out);
/* full_demod alone on an already prepared lowpassed[] buffer (drop-in contract) */
int rxo_fm_full_demod(rxo_fm_state *st, int16_t *lp, int *lp_len, int16_t *out);
/* n_blocks consecutive blocks of block_len int16; returns total int16 written to out */
long rxo_fm_stream(rxo_fm_state *st, const int16_t *in, size_t n_blocks, int block_len,
                   int16_t *out, int *per_block_len);

/* ------------------------------------------------- rx_fm channeliser (extension)
 * BASELINE configs[4] / SURVEY section 8(f) rank 2.  NOT in the reference, which has a single demod_state
 * ("multiple of these, eventually", rtl_fm.c:189) and no mixer beyond rotate16_90.  Specified here
 * entirely from reference primitives: every window of N = 2^bin_e capture samples goes through the
 * reference's integer FFT fix_fft (rtl_power.c:264-320) -- mathematically the bank of "mix by k*fs/N,
 * boxcar-sum N samples" channels, i.e. rx_fm's low_pass at ds = N for every offset at once -- and bin
 * first_bin+c of successive windows is channel c's lowpassed[] stream, demodulated by fm_demod
 * (rtl_fm.c:584-615) with its own carried pre_r/pre_j, callback block after callback block. */
typedef struct rxo_chan_cfg {
	int bin_e;               /* window = 1<<bin_e complex samples; channel spacing fs/N, channel rate fs/N */
	int first_bin;           /* channel c is FFT bin (first_bin + c) mod N */
	int n_channels;
	int custom_atan;         /* 0 std, 1 fast (as -A) */
	const int16_t *sinewave; /* rxo_sine_table(bin_e) */
} rxo_chan_cfg;
/* one callback block of len int16 (len/2 % N == 0): out[c * out_stride + w] for its len/2/N windows;
 * pre[2*c], pre[2*c+1] = the channel's carried pre_r, pre_j; work: 2*N int16 + n_channels*2*windows int16 */
void rxo_chan_block(const rxo_chan_cfg *cfg, const int16_t *in, int len, int *pre, int16_t *out, size_t out_stride);
/* the same interface for SURVEY 8(f)2's literal definition: callback scale -> integer NCO (Sinewave table, FIX_MPY per product) ->
 * low_pass at downsample N -> fm_demod, per channel (rx_oracle.c) */
void rxo_chan_nco_block(const rxo_chan_cfg *cfg, const int16_t *in, int len, int *pre, int16_t *out, size_t out_stride);

/* --------------------------------------------------------------- rx_power */

typedef struct rxo_power_cfg {
	int bin_e;               /* log2 FFT length; 0 -> rms_power path */
	int buf_len;             /* int16 per tune */
	int downsample;          /* ds */
	int downsample_passes;   /* ds_p (fifth_order cascade when boxcar == 0) */
	int boxcar;
	int comp_fir_size;
	int peak_hold;
	const int *window_coefs; /* 1<<bin_e entries, (int)(256*w) */
	const int16_t *sinewave; /* 3/4 * (1<<bin_e) entries from rxo_sine_table */
} rxo_power_cfg;

/* rtl_power.c:240-254: fills sinewave[0 .. 3n/4) for n = 1<<log2n */
void rxo_sine_table(int log2n, int16_t *sinewave);
/* rtl_power.c:256-262 */
int16_t rxo_fix_mpy(int16_t a, int16_t b);
/* rtl_power.c:264-320 with N_WAVE == 1<<m (always the case, rtl_power.c:1028) */
int rxo_fix_fft(int16_t *iq, int m, const int16_t *sinewave);
/* rtl_power.c:609-624 */
void rxo_remove_dc(int16_t *data, int length);
/* rtl_power.c:582-607 (stateless ease-in variant) */
void rxo_fifth_order_power(int16_t *data, int length);
/* rtl_power.c:626-654 */
void rxo_generic_fir_power(int16_t *data, int length, const int *fir);
/* rtl_power.c:322-401; name as accepted by -w (881-897); returns (int)(256*w) table */
int rxo_window_coefs(const char *name, int length, int *coefs);
/* rtl_power.c:403-429 */
void rxo_rms_power(const int16_t *buf, int buf_len, int peak_hold, int64_t *avg0, int *samples);
/* one tune of scanner() (rtl_power.c:709-770); work: scratch of buf_len int16 */
void rxo_power_tune(const rxo_power_cfg *cfg, const int16_t *buf16, int16_t *work, int64_t *avg, int *samples);
/* csv_dbm (rtl_power.c:774-817) without the timestamp columns; writes a NUL-terminated
 * row (with trailing newline) and resets avg/samples like the reference.  Returns length. */
int rxo_csv_row(char *dst, size_t cap, int64_t freq, int rate, int bin_e, int downsample, double crop,
                int64_t *avg, int *samples);

/* ------------------------------------------------- rx_sdr output formats (rtl_sdr.c), rx_fm WAV header */

/* rtl_sdr.c:368-370, 376-378, 384-386: n int16 values (2 per element) */
void rxo_sdr_cs16_to_cs8(const int16_t *in, size_t n, int8_t *out);
void rxo_sdr_cs16_to_cu8(const int16_t *in, size_t n, uint8_t *out);
void rxo_sdr_cs16_to_cf32(const int16_t *in, size_t n, float *out);
/* rtl_sdr.c:356-363: n_elems packed 3-byte elements -> n_elems (I,Q) int16 pairs */
void rxo_sdr_cs12_to_cs16(const uint8_t *in, size_t n_elems, int16_t *out);
/* rtl_fm.c:1174-1206: raw_mode = (mode_demod == &raw_demod) */
void rxo_wav_header(int rate, int raw_mode, uint8_t out[44]);

#ifdef __cplusplus
}
#endif
#endif
