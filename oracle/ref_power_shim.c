/* oracle/_ref/libref_power.so -- TEST INFRASTRUCTURE ONLY.
 *
 * Wraps the reference's own rx_power translation unit (REF_RTL_POWER_C =
 * "/root/reference/src/rtl_power.c"), compiled where it lies and UNMODIFIED, by
 * #including it (its device handles `dev`/`stream`, `dump`, `tuner_sleep_usec` are
 * file-static, rtl_power.c:78-79,545-547).  rtl_power.c and rtl_fm.c both define
 * fifth_order/generic_fir/cic_9_tables/frequency_range/usage with different
 * signatures, hence two separate shared objects.
 */
#define main rx_power_main
#include REF_RTL_POWER_C
#undef main
#include <stddef.h>

void soapy_fake_set_source(const int16_t *iq, size_t n_complex, size_t max_chunk);
void soapy_fake_set_discard_buffer(const void *p);
void soapy_fake_set_freq_script(const double *f, size_t n);

size_t ref_power_sizeof_tuning_state(void) { return sizeof(struct tuning_state); }
struct tuning_state *ref_power_tunes(void) { return tunes; }
int ref_power_tune_count(void) { return tune_count; }

/* flag effects of main() (rtl_power.c:908-920) */
void ref_power_set_flags(int boxcar_, int comp_fir_size_, int peak_hold_)
{
	boxcar = boxcar_; comp_fir_size = comp_fir_size_; peak_hold = peak_hold_;
}

/* what main() does between option parsing and the scan loop
 * (rtl_power.c:944, 960, 1027-1037), with window_fn chosen by name like -w */
int ref_power_setup(const char *range, double crop, const char *window)
{
	char *arg = strdup(range);
	double (*window_fn)(int, int) = rectangle;
	int i, length;
	tune_count = 0;
	frequency_range(arg, crop);
	free(arg);
	if (strcmp(window, "hamming") == 0) window_fn = hamming;
	if (strcmp(window, "blackman") == 0) window_fn = blackman;
	if (strcmp(window, "blackman-harris") == 0) window_fn = blackman_harris;
	if (strcmp(window, "hann-poisson") == 0) window_fn = hann_poisson;
	if (strcmp(window, "youssef") == 0) window_fn = youssef;
	if (strcmp(window, "kaiser") == 0) window_fn = kaiser;
	if (strcmp(window, "bartlett") == 0) window_fn = bartlett;
	dev = SoapySDRDevice_makeStrArgs("");
	stream = NULL;
	tuner_sleep_usec = 0;
	soapy_fake_set_discard_buffer(dump);
	sine_table(tunes[0].bin_e);
	free(fft_buf);
	fft_buf = malloc(tunes[0].buf_len * sizeof(int16_t) * 2);
	length = 1 << tunes[0].bin_e;
	free(window_coefs);
	window_coefs = malloc(length * sizeof(int));
	for (i = 0; i < length; i++)
		window_coefs[i] = (int)(256 * window_fn(i, length));
	return tune_count;
}

const int *ref_power_window_coefs(void) { return window_coefs; }
const int16_t *ref_power_sinewave(void) { return Sinewave; }
int ref_power_n_wave(void) { return N_WAVE; }

/* One or more scanner() passes (rtl_power.c:670) over caller-provided samples laid
 * out [pass][tune][buf_len int16].  The fake readStream hands scanner() exactly
 * buf_len int16 (= buf_len/2 complex) per tune. */
void ref_power_scan(const int16_t *in, int passes)
{
	size_t buf_len = (size_t)tunes[0].buf_len;
	soapy_fake_set_source(in, (size_t)passes * (size_t)tune_count * buf_len / 2, buf_len / 2);
	for (int p = 0; p < passes; p++)
		scanner(0);
}

/* The same with the tuner already "on frequency" for every tune, so that scanner() skips retune()
 * (its 5 ms settle sleep and flush read are device I/O, not the compute chain being timed). */
void ref_power_scan_tuned(const int16_t *in, int passes)
{
	static double *freqs;
	free(freqs);
	freqs = malloc(sizeof(double) * (size_t)tune_count);
	for (int i = 0; i < tune_count; i++)
		freqs[i] = (double)tunes[i].freq;
	soapy_fake_set_freq_script(freqs, (size_t)tune_count);
	ref_power_scan(in, passes);
	soapy_fake_set_freq_script(NULL, 0);
}

/* csv_dbm() (rtl_power.c:774) for every tune into a caller-named file, without the
 * two strftime columns main() prepends (rtl_power.c:1046-1048) */
int ref_power_csv(const char *path)
{
	int i;
	file = fopen(path, "wb");
	if (!file) return -1;
	for (i = 0; i < tune_count; i++)
		csv_dbm(&tunes[i]);
	fclose(file);
	return 0;
}

/* ---- channeliser checker: every window of 2^bin_e complex samples through the reference's OWN fix_fft (rtl_power.c:264-320,
 * with the table its own sine_table built), bins first_bin .. first_bin + n_ch - 1 (mod N) of window w written as channel c's
 * w-th decimated IQ sample: lp[c][2 * w], lp[c][2 * w + 1].  Returns fix_fft's worst return value. */
int ref_power_chan_windows(const int16_t *in, int windows, int bin_e, int first_bin, int n_ch, int16_t *lp)
{
	static int table_for = -1;
	const int n = 1 << bin_e;
	int rc = 0;
	if (table_for != bin_e) {
		sine_table(bin_e);
		table_for = bin_e;
	}
	int16_t *win = malloc((size_t)2 * n * sizeof(int16_t));
	for (int w = 0; w < windows; w++) {
		memcpy(win, in + (size_t)w * 2 * n, (size_t)2 * n * sizeof(int16_t));
		int r = fix_fft(win, bin_e);
		if (r < rc)
			rc = r;
		for (int c = 0; c < n_ch; c++) {
			const int bin = (first_bin + c) & (n - 1);
			lp[((size_t)c * windows + w) * 2] = win[2 * bin];
			lp[((size_t)c * windows + w) * 2 + 1] = win[2 * bin + 1];
		}
	}
	free(win);
	return rc;
}
