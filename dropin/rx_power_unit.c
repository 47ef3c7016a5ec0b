/* dropin/rx_power_unit.c -- rx_power with scanner()'s per-tune compute on the MI355X, built from the USER'S OWN rtl_power.c,
 * unmodified (RXGPU_REF_RTL_POWER_C = "<rx_tools>/src/rtl_power.c", set by dropin/Makefile; nothing of it is copied).
 *
 * Two declarations in front of the file make its scanner (rtl_power.c:670) and csv_dbm (774) WEAK definitions; the strong ones in
 * rx_power_hooks.c win at link time, so main()'s calls (rtl_power.c:1040, 1049) land there.  What follows the #include is ours and
 * sees the file's statics (dev, stream, do_exit).
 *
 * scanner() interleaves device I/O (retune + readStream per tune, 679-703) with the per-tune compute (709-770).  The tunes are
 * independent, so the two are separated: first every tune's buf16 is read exactly as the reference reads it, then all tunes go to the
 * device in one rxgpu_scan.  The sums stay on the device between sweeps (rxgpu_scan_deferred: tunes[] is a static global, it outlives
 * every pending interval) and come home once per report, in front of the file's own csv_dbm.
 */
#include <rxgpu.h>

struct tuning_state;
void scanner(size_t channel) __attribute__((weak));                 /* rtl_power.c:670: overridable */
void csv_dbm(struct tuning_state *ts) __attribute__((weak));        /* rtl_power.c:774 */

#include RXGPU_REF_RTL_POWER_C

/* ------------------------------------------------------------------ behind the reference's code */

/* the file's own csv_dbm under a second name (the first one is overridden) */
void rxgpu_dropin_ref_csv_dbm(struct tuning_state *ts) __attribute__((alias("csv_dbm")));

/* the capture-replay SoapySDR stand-in, when that is what is linked: retune()'s flush reads (into the file-static `dump`) must
 * not eat the capture */
void soapy_fake_set_discard_buffer(const void *p) __attribute__((weak));

static void rxgpu_dropin_power_setup(void)
{
	if (rxgpu_init(-1) != RXGPU_OK) {
		fprintf(stderr, "rx_power (rxgpu): %s\n", rxgpu_last_error());
		exit(1);
	}
	rxgpu_scan_deferred(1);
	if (soapy_fake_set_discard_buffer)
		soapy_fake_set_discard_buffer(dump);
}

static void rxgpu_dropin_die(const char *what)
{
	/* the reference's exit(1) would write out what `file` (the CSV, rtl_power.c:1007-1016) still buffers; rxgpu_fatal leaves with _exit.
	 * rx_power is single-threaded: nothing else can be inside stdio here */
	if (file)
		fflush(file);
	rxgpu_fatal(what);                                             /* one line on stderr, device drained, _exit(1) */
}

void rxgpu_dropin_scanner(size_t channel)
{
	static int ready;
	static unsigned char got[MAX_TUNES];
	int i, missed = 0, n_read;
	if (!ready) {
		rxgpu_dropin_power_setup();
		ready = 1;
	}
	/* the I/O half of the loop body: on-frequency check, retune, one read per tune into its own buf16 */
	for (i = 0; i < tune_count; i++) {
		struct tuning_state *ts = &tunes[i];
		void *buffs[] = { ts->buf16 };
		int flags = 0, r;
		long long timeNs = 0;
		if (do_exit >= 2)
			break;                                                 /* the reference returns here with tunes [0, i) already integrated */
		if ((int64_t)SoapySDRDevice_getFrequency(dev, SOAPY_SDR_RX, channel) != ts->freq)
			retune(dev, stream, ts->freq, channel);
		r = SoapySDRDevice_readStream(dev, stream, buffs, tunes[0].buf_len, &flags, &timeNs, 1000000);
		got[i] = r >= 0;
		if (r < 0) {
			fprintf(stderr, "Error: reading stream %d\n", r);      /* the reference skips the tune's compute (`continue`) */
			missed++;
		}
	}
	n_read = i;
	for (; i < tune_count; i++) {
		got[i] = 0;                                                    /* an interrupted sweep: the tunes never read count as missed */
		missed++;
	}
	/* the compute half, every tune that was read: one call when none was missed */
	if (!missed) {
		if (rxgpu_scan(tunes, tune_count, window_coefs, Sinewave, boxcar, comp_fir_size, peak_hold) != RXGPU_OK)
			rxgpu_dropin_die("rxgpu_scan");
		return;
	}
	if (missed == tune_count)
		return;
	/* some reads failed: the runs of tunes that were read go one by one, merged at once (another array start per call) */
	if (rxgpu_scan_sync(tunes, tune_count) != RXGPU_OK || rxgpu_scan_deferred(0) != RXGPU_OK)
		rxgpu_dropin_die("rxgpu_scan_sync");
	for (i = 0; i < n_read; ) {
		int j = i;
		while (j < n_read && got[j])
			j++;
		/* scanner() takes the geometry from tunes[0] for every tune (rtl_power.c:676-678); frequency_range gives them all the same */
		if (j > i && rxgpu_scan(&tunes[i], j - i, window_coefs, Sinewave, boxcar, comp_fir_size, peak_hold) != RXGPU_OK)
			rxgpu_dropin_die("rxgpu_scan");
		i = j + 1;
	}
	rxgpu_scan_deferred(1);
}

void rxgpu_dropin_csv_dbm(struct tuning_state *ts)
{
	/* the report (rtl_power.c:1045-1050): the interval's sums come home once, in front of the first row */
	if (rxgpu_scan_sync(tunes, tune_count) != RXGPU_OK)
		rxgpu_dropin_die("rxgpu_scan_sync");
	rxgpu_dropin_ref_csv_dbm(ts);
	/* the file's csv_dbm has zeroed the row (rtl_power.c:815-817): the next interval's merge need not read it across the link */
	rxgpu_scan_rows_cleared(ts, 1);
}
