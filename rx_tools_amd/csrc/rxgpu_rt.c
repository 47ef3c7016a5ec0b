/* rxgpu_rt.c -- runtime of librxgpu: device binding, the launch stream, error text and
 * per-kernel event timing.  Plain C over the HIP C API. */
#include "rxgpu_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_device = -1;
static hipStream_t g_stream, g_stream2;
static char g_err[512];

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

int rxgpu_init(int device)
{
	int n = 0;
	if (device < 0) {
		const char *e = getenv("RXGPU_DEVICE");
		if (!e || !*e)
			e = getenv("LOCAL_RANK");
		device = (e && *e) ? atoi(e) : 0;
	}
	if (g_device == device)
		return RXGPU_OK;
	if (g_device >= 0)
		rxgpu_shutdown();
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return rxgpu_fail(RXGPU_ENODEV, "no HIP device visible (librxgpu has no CPU fallback)");
	if (device >= n)
		return rxgpu_fail(RXGPU_ENODEV, "device %d requested but only %d visible", device, n);
	RX_HIP(hipSetDevice(device));
	RX_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
	{
		/* the tail stream carries many small kernels behind a saturating one: give it the high priority */
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
	rxgpu_prof_reset();
	hipStreamDestroy(g_stream);
	hipStreamDestroy(g_stream2);
	g_stream = g_stream2 = NULL;
	g_device = -1;
}

int rxgpu_ensure_init(void)
{
	if (g_device >= 0)
		return RXGPU_OK;
	return rxgpu_init(-1);
}

hipStream_t rxgpu_hip_stream(void) { return g_stream; }
hipStream_t rxgpu_hip_stream2(void) { return g_stream2; }
void *rxgpu_stream(void) { return (void *)g_stream; }

int rxgpu_sync(void)
{
	if (g_device < 0)
		return rxgpu_fail(RXGPU_ENODEV, "rxgpu_sync before rxgpu_init");
	RX_HIP(hipStreamSynchronize(g_stream));
	RX_HIP(hipStreamSynchronize(g_stream2));
	rxgpu_prof_collect();
	return RXGPU_OK;
}

/* ------------------------------------------------------------------ event timing */

#define PROF_NAMES 24
#define PROF_PENDING 8192

struct prof_total { char name[24]; double ms; long n; };
struct prof_pair { int slot; hipEvent_t a, b; };

static int g_prof_on;
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
	return !strcmp(name, "fm_decimate") || !strcmp(name, "pw_fft") || !strcmp(name, "fm_fifth") || !strcmp(name, "ch_fft");
}

void rxgpu_prof_begin_on(const char *name, hipStream_t st)
{
	if (!g_prof_on || g_npend == PROF_PENDING || !prof_wanted(name))
		return;
	struct prof_pair *p = &g_pend[g_npend];
	p->slot = prof_slot(name);
	if (p->slot < 0)
		return;
	p->a = prof_event();
	p->b = NULL;
	hipEventRecord(p->a, st);
}

void rxgpu_prof_end_on(const char *name, hipStream_t st)
{
	if (!g_prof_on || g_npend == PROF_PENDING || !prof_wanted(name))
		return;
	struct prof_pair *p = &g_pend[g_npend];
	if (p->slot < 0 || !p->a)
		return;
	p->b = prof_event();
	hipEventRecord(p->b, st);
	g_npend++;
	if (g_npend < PROF_PENDING) {
		g_pend[g_npend].a = NULL;
		g_pend[g_npend].slot = -1;
	}
}

void rxgpu_prof_collect(void)
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
	g_pend[0].a = NULL;
	g_pend[0].slot = -1;
}

void rxgpu_prof_reset(void)
{
	if (g_device >= 0 && g_npend) {
		hipStreamSynchronize(g_stream);
		hipStreamSynchronize(g_stream2);
		rxgpu_prof_collect();
	}
	g_ntot = 0;
}

int rxgpu_prof_get(const char *name, double *total_ms, long *launches)
{
	if (g_device >= 0 && g_npend) {
		hipStreamSynchronize(g_stream);
		hipStreamSynchronize(g_stream2);
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
