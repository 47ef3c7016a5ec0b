/* rxgpu_ref_structs.h -- layouts of the reference structs the drop-in entry points take.
 *
 * These restate, field for field, the state structs of rxseger/rx_tools v1.0.3 so that
 * librxgpu can be handed the reference's own objects by pointer:
 *
 *   struct dongle_state   /root/reference/src/rtl_fm.c:104-122
 *   struct demod_state    /root/reference/src/rtl_fm.c:124-159   (1 049 160 bytes, x86-64 glibc)
 *   struct tuning_state   /root/reference/src/rtl_power.c:89-108
 *
 * A maintainer who patches the two call sites does NOT include this file:
 * the reference's own definitions are the same types.  It exists for librxgpu's own
 * translation units and for out-of-tree callers.  tests/test_abi.py checks sizeof/offsetof
 * against the values reported by the reference's own translation unit (oracle/_ref).
 */
#ifndef RXGPU_REF_STRUCTS_H
#define RXGPU_REF_STRUCTS_H

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>

#define RXGPU_MAXIMUM_OVERSAMPLE 16                       /* rtl_fm.c:81 */
#define RXGPU_DEFAULT_BUF_LENGTH (1 * 16384)              /* rtl_fm.c:80 */
#define RXGPU_MAXIMUM_BUF_LENGTH (RXGPU_MAXIMUM_OVERSAMPLE * RXGPU_DEFAULT_BUF_LENGTH) /* :82 */

struct output_state;
struct demod_state;

struct dongle_state {
	int exit_flag;
	pthread_t thread;
	void *dev;                       /* SoapySDRDevice* */
	void *stream;                    /* SoapySDRStream* */
	size_t channel;
	char *dev_query;
	uint32_t freq;
	uint32_t rate;
	uint32_t bandwidth;
	char *gain_str;
	int16_t buf16[RXGPU_MAXIMUM_BUF_LENGTH];
	int ppm_error;
	int offset_tuning;
	int direct_sampling;
	int mute;
	struct demod_state *demod_target;
};

struct demod_state {
	int exit_flag;
	pthread_t thread;
	int16_t lowpassed[RXGPU_MAXIMUM_BUF_LENGTH];
	int lp_len;
	int16_t lp_i_hist[10][6];
	int16_t lp_q_hist[10][6];
	int16_t result[RXGPU_MAXIMUM_BUF_LENGTH];
	int16_t droop_i_hist[9];
	int16_t droop_q_hist[9];
	int result_len;
	int rate_in;
	int rate_out;
	int rate_out2;
	int now_r, now_j;
	int pre_r, pre_j;
	int prev_index;
	int downsample;                  /* min 1, max 256 */
	int post_downsample;
	int output_scale;
	int squelch_level, conseq_squelch, squelch_hits, terminate_on_squelch, squelch_zero;
	int downsample_passes;
	int comp_fir_size;
	int custom_atan;
	int deemph, deemph_a;
	int now_lpr;
	int prev_lpr_index;
	int dc_block_audio, dc_avg, adc_block_const;
	int dc_block_raw, dc_avgI, dc_avgQ, rdc_block_const;
	void (*mode_demod)(struct demod_state *);
	pthread_rwlock_t rw;
	pthread_cond_t ready;
	pthread_mutex_t ready_m;
	struct output_state *output_target;
};

struct tuning_state {
	int64_t freq;
	int rate;
	int bin_e;
	int64_t *avg;                    /* length == 2^bin_e */
	int samples;
	int downsample;
	int downsample_passes;
	double crop;
	int16_t *buf16;
	int buf_len;
};

#endif
