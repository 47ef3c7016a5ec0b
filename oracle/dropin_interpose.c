/* oracle/_ref/libdropin.so -- TEST INFRASTRUCTURE ONLY (tests/test_dropin_e2e.py).
 *
 * Shows librxgpu dropping into the reference's OWN rx_fm main()/threads without touching
 * its source: libref_fm.so (rtl_fm.c compiled unmodified, -fPIC) calls full_demod() through
 * its PLT, so a definition that sits earlier in the global symbol scope replaces it.  This
 * file provides that definition -- the one-line patch of INTEGRATION.md done by the dynamic
 * linker instead of an editor -- plus the pacing the reference's lossy single-slot hand-off
 * (rtl_fm.c:858-862, 921-924) needs to make an end-to-end run reproducible.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <semaphore.h>
#include <signal.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>
#include "rxgpu.h"
#include "rxgpu_ref_structs.h"

static sem_t done;
static int sem_ready;
static long calls;

__attribute__((constructor)) static void setup(void) { sem_init(&done, 0, 0); sem_ready = 1; }

/* replaces the reference's full_demod (rtl_fm.c:759) for every caller in the process */
void full_demod(struct demod_state *d)
{
	rxgpu_full_demod(d);
	calls++;
	sem_post(&done);
}

long dropin_calls(void) { return calls; }

/* soapy_fake pace hook: hand out the next block only after the previous one went through
 * full_demod (and give the output thread a moment to write it) */
void dropin_pace(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_REALTIME, &ts);
	ts.tv_sec += 5;
	if (sem_ready)
		sem_timedwait(&done, &ts);
	usleep(3000);
}

/* soapy_fake end-of-stream hook: what a user's ^C does (rtl_fm.c:274-278) */
void dropin_eos(void)
{
	usleep(50000);
	raise(SIGINT);
}

/* ------------------------------------------------------------------ rx_power
 * replaces the reference's scanner() (rtl_power.c:670) for every caller in the process: the device
 * I/O half of the original loop (readStream per tune, 688-703) followed by rxgpu_scan for the
 * compute half (709-770), exactly the patch of INTEGRATION.md section 2.  The reference's globals are
 * looked up at run time so that this object also loads next to libref_fm.so. */
typedef int (*read_stream_fn)(void *device, void *stream, void *const *buffs, const size_t numElems, int *flags,
                              long long *timeNs, const long timeoutUs);
static long scans;

void scanner(size_t channel)
{
	struct tuning_state *tunes = dlsym(RTLD_DEFAULT, "tunes");
	int tune_count = *(int *)dlsym(RTLD_DEFAULT, "tune_count");
	int *window_coefs = *(int **)dlsym(RTLD_DEFAULT, "window_coefs");
	int16_t *sinewave = *(int16_t **)dlsym(RTLD_DEFAULT, "Sinewave");
	int boxcar = *(int *)dlsym(RTLD_DEFAULT, "boxcar");
	int comp_fir_size = *(int *)dlsym(RTLD_DEFAULT, "comp_fir_size");
	int peak_hold = *(int *)dlsym(RTLD_DEFAULT, "peak_hold");
	read_stream_fn SoapySDRDevice_readStream = (read_stream_fn)dlsym(RTLD_DEFAULT, "SoapySDRDevice_readStream");
	(void)channel;
	static int16_t flush[4 * 16384];
	static int deferred;
	if (!deferred) {                   /* tunes[] is a static global of the reference: it outlives every pending interval */
		rxgpu_scan_deferred(1);
		deferred = 1;
	}
	for (int i = 0; i < tune_count; i++) {
		void *buffs[] = { tunes[i].buf16 };
		void *fbuffs[] = { flush };
		int flags = 0;
		long long timeNs = 0;
		/* retune()'s settle-and-flush read (rtl_power.c:560-576): part of the device I/O, kept */
		(void)SoapySDRDevice_readStream(NULL, NULL, fbuffs, 16384, &flags, &timeNs, 1000000);
		if (SoapySDRDevice_readStream(NULL, NULL, buffs, (size_t)tunes[0].buf_len, &flags, &timeNs, 1000000) < 0) {
			usleep(1000);
			return;                    /* capture exhausted: nothing to add until main() reports */
		}
	}
	if (rxgpu_scan(tunes, tune_count, window_coefs, sinewave, boxcar, comp_fir_size, peak_hold) != RXGPU_OK) {
		fprintf(stderr, "rxgpu_scan: %s\n", rxgpu_last_error());
		_exit(1);
	}
	scans++;
}

long dropin_scans(void) { return scans; }

/* main()'s report loop (rtl_power.c:1045-1050) reads tunes[i].avg through the reference's own csv_dbm: the one added line of
 * the INTEGRATION.md patch -- rxgpu_scan_sync in front of it -- brings the sums the sweeps left on the device home first
 * (a no-op from the second row of an interval on), then the reference's csv_dbm prints the row */
void csv_dbm(struct tuning_state *ts)
{
	static void (*ref_csv_dbm)(struct tuning_state *);
	if (!ref_csv_dbm) {
		/* the reference object was dlopen'ed after this one (RTLD_NEXT does not reach it): find it through a symbol only it
		 * defines, and take ITS csv_dbm */
		Dl_info info;
		void *h = NULL;
		if (dladdr(dlsym(RTLD_DEFAULT, "tunes"), &info) && info.dli_fname)
			h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
		if (h)
			ref_csv_dbm = (void (*)(struct tuning_state *))dlsym(h, "csv_dbm");
		if (!ref_csv_dbm || ref_csv_dbm == csv_dbm) {
			fprintf(stderr, "dropin: the reference's csv_dbm was not found\n");
			_exit(1);
		}
	}
	if (rxgpu_scan_sync((struct tuning_state *)dlsym(RTLD_DEFAULT, "tunes"), *(int *)dlsym(RTLD_DEFAULT, "tune_count")) != RXGPU_OK) {
		fprintf(stderr, "rxgpu_scan_sync: %s\n", rxgpu_last_error());
		_exit(1);
	}
	ref_csv_dbm(ts);
}
