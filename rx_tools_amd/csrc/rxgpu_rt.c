/* rxgpu_rt.c -- runtime of librxgpu: device binding, the launch stream, error text and
 * per-kernel event timing.  Plain C over the HIP C API. */
#include "rxgpu_internal.h"
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <string.h>

static int g_device = -1;
static hipStream_t g_stream, g_stream2, g_stream3, g_stream4;
/* The drop-in is entered from two threads of the reference (rtlsdr_callback on the dongle thread, full_demod on the
 * demod thread, rtl_fm.c:899/923): the error text and the "this thread is bound to the device" flag are per thread,
 * initialisation is serialised. */
static __thread char g_err[512];
static __thread int t_bound_device = -1;
static pthread_mutex_t g_init_lock = PTHREAD_MUTEX_INITIALIZER;

int rxgpu_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

const char *rxgpu_last_error(void) { return g_err; }

int rxgpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

static int init_locked(int device);

/* ------------------------------------------------------------------ tuning knobs
 * $RXGPU_* switches (A/B of kernel variants, test hooks; INTEGRATION.md lists them) are read from the environment in ONE place and
 * at known times -- rxgpu_init, the creation of a stream / channeliser / scan object, rxgpu_knobs_reload -- into a snapshot the
 * launch paths read: no getenv() on the per-block path (getenv is not safe against a concurrent setenv, and the drop-in runs on
 * two threads of the caller), and a knob cannot flip between two chained runs of one object.  A value that changes gets a fresh
 * copy and the old one is never freed (a reader may still hold it): a few bytes per changed knob. */
static const char *const g_knob_names[] = {
	/* behaviour a caller may choose */
	"RXGPU_SCAN_DEFERRED", "RXGPU_SCAN_ZC", "RXGPU_DROPIN_FAST", "RXGPU_DROPIN_ZC", "RXGPU_HOST_CHUNK", "RXGPU_DROPIN_TIMING",
	/* test hooks: regimes that only very long runs reach by themselves, forced wrong libm samples, short wave walks */
	"RXGPU_DEEMPH_TOPCAP", "RXGPU_DEEMPH_CHUNK", "RXGPU_SCAN_T", "RXGPU_FLAG_ALL", "RXGPU_DL_TW",
};
#define N_KNOBS ((int)(sizeof(g_knob_names) / sizeof(g_knob_names[0])))
static const char *volatile g_knob_val[sizeof(g_knob_names) / sizeof(g_knob_names[0])];
static pthread_mutex_t g_knob_lock = PTHREAD_MUTEX_INITIALIZER;

#ifdef RXGPU_FAULT_INJECT
static volatile long g_fail_after;      /* $RXGPU_FAIL_AFTER (test hook, rxgpu_fault_tick) */
#endif

void rxgpu_knobs_reload(void)
{
	pthread_mutex_lock(&g_knob_lock);
	for (int i = 0; i < N_KNOBS; i++) {
		const char *e = getenv(g_knob_names[i]);
		const char *old = g_knob_val[i];
		if (!e)
			g_knob_val[i] = NULL;
		else if (!old || strcmp(old, e))
			g_knob_val[i] = strdup(e);
	}
#ifdef RXGPU_FAULT_INJECT
	{
		const char *e = getenv("RXGPU_FAIL_AFTER");
		g_fail_after = e ? atol(e) : 0;
	}
#endif
	pthread_mutex_unlock(&g_knob_lock);
}

/* Test hook, TEST BUILD ONLY (librxgpu_fi.so: the host files compiled with -DRXGPU_FAULT_INJECT; the shipped library has neither this
 * function nor the call in RX_K): $RXGPU_FAIL_AFTER=n, read with the knobs -- the n-th kernel launch of the host code from now on reports
 * hipErrorLaunchFailure instead of being enqueued, the way tests/test_gpu_dropin.py makes a device error happen in the middle of a stream. */
#ifdef RXGPU_FAULT_INJECT
int rxgpu_fault_tick(void)
{
	if (g_fail_after <= 0)
		return 0;
	return __sync_sub_and_fetch(&g_fail_after, 1) == 0;
}
#endif

/* The library's failure convention where the replaced function is `void` (full_demod, rtlsdr_callback, scanner: SURVEY.md 8b -- the
 * reference prints to stderr and exits): say it ONCE on stderr -- never stdout, that is the audio / CSV stream --, let the device finish
 * what is in flight, and leave with _exit(1) (the driver reclaims the device memory with the process).  Not exit(): exit() runs the atexit handlers and static destructors of whatever
 * SoapySDR driver is loaded while the application's other thread (dongle / demod / output) is still running, possibly inside that driver
 * or holding d->rw -- a process that never ends.  A device that no longer answers must not hold the process either: a watchdog thread
 * ends it after five seconds whatever the synchronisation is waiting for. */
static void *fatal_watchdog(void *arg)
{
	(void)arg;
	struct timespec ts = { 5, 0 };
	nanosleep(&ts, NULL);
	_exit(1);
	return NULL;
}

void rxgpu_fatal(const char *what)
{
	static int once;
	if (__sync_lock_test_and_set(&once, 1))
		for (;;) pause();                                /* another thread is already taking the process down */
	fprintf(stderr, "rxgpu: %s: %s\n", what ? what : "fatal", rxgpu_last_error());
	fflush(stderr);
	pthread_t w;
	if (pthread_create(&w, NULL, fatal_watchdog, NULL) == 0)
		pthread_detach(w);
	else {
		/* no watchdog thread: SIGALRM's default action ends the process just as surely if the device never answers */
		signal(SIGALRM, SIG_DFL);
		alarm(5);
	}
	/* NOT rxgpu_shutdown(): that frees the drop-in's buffers and stream objects, which the application's other thread may be using this
	 * very moment -- a use-after-free on the way out.  What is in flight is drained; the device memory goes back with the process. */
	if (g_device >= 0)
		(void)hipDeviceSynchronize();
	_exit(1);
}

const char *rxgpu_knob(const char *name)
{
	for (int i = 0; i < N_KNOBS; i++)
		if (!strcmp(g_knob_names[i], name))
			return g_knob_val[i];
	fprintf(stderr, "librxgpu: knob %s is not in the snapshot table (rxgpu_rt.c)\n", name);
	abort();
}

int rxgpu_init(int device)
{
	pthread_mutex_lock(&g_init_lock);
	int rc = init_locked(device);
	pthread_mutex_unlock(&g_init_lock);
	return rc;
}

static int init_locked(int device)
{
	int n = 0;
	rxgpu_knobs_reload();
	if (device < 0) {
		const char *e = getenv("RXGPU_DEVICE");
		if (!e || !*e)
			e = getenv("LOCAL_RANK");
		device = (e && *e) ? atoi(e) : 0;
	}
	if (g_device == device) {
		/* hipSetDevice is per thread: bind the caller too */
		if (t_bound_device != device) {
			RX_HIP(hipSetDevice(device));
			t_bound_device = device;
		}
		return RXGPU_OK;
	}
	if (g_device >= 0)
		rxgpu_shutdown();
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return rxgpu_fail(RXGPU_ENODEV, "no HIP device visible (librxgpu has no CPU fallback)");
	if (device >= n)
		return rxgpu_fail(RXGPU_ENODEV, "device %d requested but only %d visible", device, n);
	RX_HIP(hipSetDevice(device));
	{
		/* the kernels' wave-private LDS exchanges and DPP scans assume 64 lanes per wave */
		int ws = 0;
		RX_HIP(hipDeviceGetAttribute(&ws, hipDeviceAttributeWarpSize, device));
		if (ws != 64)
			return rxgpu_fail(RXGPU_ENODEV, "device %d has %d-wide wavefronts; librxgpu is written for gfx950 (64)", device, ws);
	}
	t_bound_device = device;
	RX_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
	RX_HIP(hipStreamCreateWithFlags(&g_stream3, hipStreamNonBlocking));
	RX_HIP(hipStreamCreateWithFlags(&g_stream4, hipStreamNonBlocking));
	{
		/* the tail stream carries many small kernels behind a saturating one: give it the high priority (measured: normal or
		 * low priority changes nothing -- what those kernels wait for is wave slots, see rxk_fm_decimate_small / rxk_fm_fifth_fused) */
		int lo_p = 0, hi_p = 0;
		if (hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) != hipSuccess)
			lo_p = hi_p = 0;
		RX_HIP(hipStreamCreateWithPriority(&g_stream2, hipStreamNonBlocking, hi_p));
	}
	g_device = device;
	return RXGPU_OK;
}

void rxgpu_shutdown(void)
{
	if (g_device < 0)
		return;
	hipStreamSynchronize(g_stream);
	hipStreamSynchronize(g_stream2);
	hipStreamSynchronize(g_stream3);
	hipStreamSynchronize(g_stream4);
	/* what the drop-ins keep between calls (stream objects, the callback's device buffers, the scan object and its staging) lives
	 * on this device: none of it may survive into a re-initialisation on another one */
	rxgpu_fm_dropin_release();
	rxgpu_power_dropin_release();
	rxgpu_prof_reset();
	hipStreamDestroy(g_stream);
	hipStreamDestroy(g_stream2);
	hipStreamDestroy(g_stream3);
	hipStreamDestroy(g_stream4);
	g_stream = g_stream2 = g_stream3 = g_stream4 = NULL;
	g_device = -1;
}

int rxgpu_ensure_init(void)
{
	/* fast path: initialised, and this thread already talks to the right device */
	if (g_device >= 0 && t_bound_device == g_device)
		return RXGPU_OK;
	return rxgpu_init(g_device >= 0 ? g_device : -1);
}

hipStream_t rxgpu_hip_stream(void) { return g_stream; }
hipStream_t rxgpu_hip_stream2(void) { return g_stream2; }
hipStream_t rxgpu_hip_stream3(void) { return g_stream3; }
hipStream_t rxgpu_hip_stream4(void) { return g_stream4; }

/* Host buffers the caller wants DMA'd without a bounce (SURVEY.md section 8b "Ownership"): page-lock them in place. */
/* page-lock registrations come and go: whoever caches a device address of host memory keys it with this */
static volatile unsigned g_pin_gen = 1;
unsigned rxgpu_pin_generation(void) { return g_pin_gen; }
void rxgpu_pin_changed(void) { __sync_fetch_and_add(&g_pin_gen, 1); }

int rxgpu_pin(void *ptr, size_t bytes)
{
	int rc;
	if (!ptr || !bytes)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_pin: bad arguments");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	RX_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
	rxgpu_pin_changed();
	return RXGPU_OK;
}

int rxgpu_unpin(void *ptr)
{
	if (!ptr)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_unpin: null pointer");
	rxgpu_pin_changed();
	RX_HIP(hipHostUnregister(ptr));
	return RXGPU_OK;
}
void *rxgpu_stream(void) { return (void *)g_stream; }

int rxgpu_sync(void)
{
	if (g_device < 0)
		return rxgpu_fail(RXGPU_ENODEV, "rxgpu_sync before rxgpu_init");
	RX_HIP(hipStreamSynchronize(g_stream));
	RX_HIP(hipStreamSynchronize(g_stream2));
	RX_HIP(hipStreamSynchronize(g_stream3));
	RX_HIP(hipStreamSynchronize(g_stream4));
	rxgpu_prof_collect();
	return RXGPU_OK;
}

/* ------------------------------------------------------------------ event timing */

#define PROF_NAMES 24
#define PROF_PENDING 8192

struct prof_total { char name[24]; double ms; long n; };
struct prof_pair { int slot; hipEvent_t a, b; };

static int g_prof_on;
static pthread_mutex_t g_prof_lock = PTHREAD_MUTEX_INITIALIZER;   /* begin/end pairs come from one thread at a time per stream object; the tables are shared */
static struct prof_total g_tot[PROF_NAMES];
static int g_ntot;
static struct prof_pair g_pend[PROF_PENDING];
static int g_npend;
static hipEvent_t g_free[2 * PROF_PENDING];
static int g_nfree;

static int prof_slot(const char *name)
{
	for (int i = 0; i < g_ntot; i++)
		if (!strcmp(g_tot[i].name, name))
			return i;
	if (g_ntot == PROF_NAMES)
		return -1;
	snprintf(g_tot[g_ntot].name, sizeof(g_tot[g_ntot].name), "%s", name);
	g_tot[g_ntot].ms = 0;
	g_tot[g_ntot].n = 0;
	return g_ntot++;
}

static hipEvent_t prof_event(void)
{
	hipEvent_t e = NULL;
	if (g_nfree)
		return g_free[--g_nfree];
	hipEventCreate(&e);
	return e;
}

void rxgpu_prof_enable(int on) { g_prof_on = on; }

/* level 1 times only the kernels that dominate each path, level 2 everything */
static int prof_wanted(const char *name)
{
	if (g_prof_on >= 2)
		return 1;
	return !strcmp(name, "fm_decimate") || !strcmp(name, "pw_fft") || !strcmp(name, "fm_fifth") || !strcmp(name, "ch_fft") ||
	       !strcmp(name, "pw_gather");
}

/* the begin of a pair waits in a per-thread slot; the shared tables are only touched under the lock */
static __thread struct prof_pair t_cur = { -1, NULL, NULL };

void rxgpu_prof_begin_on(const char *name, hipStream_t st)
{
	if (!g_prof_on || !prof_wanted(name))
		return;
	pthread_mutex_lock(&g_prof_lock);
	t_cur.slot = g_npend < PROF_PENDING ? prof_slot(name) : -1;
	t_cur.a = t_cur.slot >= 0 ? prof_event() : NULL;
	pthread_mutex_unlock(&g_prof_lock);
	if (t_cur.a)
		hipEventRecord(t_cur.a, st);
}

void rxgpu_prof_end_on(const char *name, hipStream_t st)
{
	if (!g_prof_on || !prof_wanted(name) || t_cur.slot < 0 || !t_cur.a)
		return;
	pthread_mutex_lock(&g_prof_lock);
	t_cur.b = prof_event();
	pthread_mutex_unlock(&g_prof_lock);
	hipEventRecord(t_cur.b, st);
	pthread_mutex_lock(&g_prof_lock);
	if (g_npend < PROF_PENDING) {
		g_pend[g_npend++] = t_cur;
	} else {
		g_free[g_nfree++] = t_cur.a;
		g_free[g_nfree++] = t_cur.b;
	}
	pthread_mutex_unlock(&g_prof_lock);
	t_cur.slot = -1;
	t_cur.a = NULL;
}

void rxgpu_prof_abort(void)
{
	if (t_cur.slot < 0 || !t_cur.a)
		return;
	pthread_mutex_lock(&g_prof_lock);
	g_free[g_nfree++] = t_cur.a;
	pthread_mutex_unlock(&g_prof_lock);
	t_cur.slot = -1;
	t_cur.a = NULL;
}

static void prof_collect_locked(void);

void rxgpu_prof_collect(void)
{
	if (!g_npend)
		return;
	pthread_mutex_lock(&g_prof_lock);
	prof_collect_locked();
	pthread_mutex_unlock(&g_prof_lock);
}

static void prof_collect_locked(void)
{
	for (int i = 0; i < g_npend; i++) {
		float ms = 0;
		struct prof_pair *p = &g_pend[i];
		if (hipEventElapsedTime(&ms, p->a, p->b) == hipSuccess) {
			g_tot[p->slot].ms += ms;
			g_tot[p->slot].n += 1;
		}
		g_free[g_nfree++] = p->a;
		g_free[g_nfree++] = p->b;
	}
	g_npend = 0;
}

void rxgpu_prof_reset(void)
{
	if (g_device >= 0 && g_npend) {
		hipStreamSynchronize(g_stream);
		hipStreamSynchronize(g_stream2);
		hipStreamSynchronize(g_stream3);
		hipStreamSynchronize(g_stream4);
		rxgpu_prof_collect();
	}
	g_ntot = 0;
}

int rxgpu_prof_get(const char *name, double *total_ms, long *launches)
{
	if (g_device >= 0 && g_npend) {
		hipStreamSynchronize(g_stream);
		hipStreamSynchronize(g_stream2);
		hipStreamSynchronize(g_stream3);
		hipStreamSynchronize(g_stream4);
		rxgpu_prof_collect();
	}
	for (int i = 0; i < g_ntot; i++)
		if (!strcmp(g_tot[i].name, name)) {
			if (total_ms) *total_ms = g_tot[i].ms;
			if (launches) *launches = g_tot[i].n;
			return RXGPU_OK;
		}
	if (total_ms) *total_ms = 0;
	if (launches) *launches = 0;
	return RXGPU_EINVAL;
}
