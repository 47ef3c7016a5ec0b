/* rxgpu_internal.h -- shared by the C host files of librxgpu (not public). */
#ifndef RXGPU_INTERNAL_H
#define RXGPU_INTERNAL_H

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include "rxgpu.h"
#include "kernels.h"

/* record an error message (printf-style) and return `code` */
int rxgpu_fail(int code, const char *fmt, ...);
/* make sure rxgpu_init ran (auto-initialises with device -1) */
int rxgpu_ensure_init(void);
unsigned rxgpu_pin_generation(void);   /* rxgpu_rt.c: changes whenever host memory is page-locked or released */
void rxgpu_pin_changed(void);
int rxgpu_deemph_warm64(int a);          /* rxgpu_chan.c: samples that bring any two int16 de-emphasis states within 64 of each other */
hipStream_t rxgpu_hip_stream(void);
hipStream_t rxgpu_hip_stream2(void);   /* second stream: the latency-bound tail of a pipelined rx_fm run */
hipStream_t rxgpu_hip_stream4(void);   /* fourth stream: small kernels that prepare the NEXT run's stream-A launch while this run's is still going */
hipStream_t rxgpu_hip_stream3(void);   /* third stream: host <-> device copies of the host-fed entry points */

#define RX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
	return rxgpu_fail(RXGPU_ENODEV, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#ifdef RXGPU_FAULT_INJECT              /* the test build (librxgpu_fi.so): $RXGPU_FAIL_AFTER makes the n-th launch fail, rxgpu_rt.c */
int rxgpu_fault_tick(void);
#define RX_FAULT() rxgpu_fault_tick()
#else
#define RX_FAULT() 0
#endif
#define RX_K(call) do { int e_ = RX_FAULT() ? (int)hipErrorLaunchFailure : (call); if (e_ != 0) \
	return rxgpu_fail(RXGPU_ENODEV, "%s launch failed: %s (%s:%d)", #call, hipGetErrorString((hipError_t)e_), __FILE__, __LINE__); } while (0)

/* rxgpu_shutdown: free what the drop-in entry points cache between calls (rxgpu_fm.c, rxgpu_power.c) */
void rxgpu_fm_dropin_release(void);
void rxgpu_power_dropin_release(void);

/* fix_fft twiddles for the device: n/2 plain + n/2 doubled entries (rxgpu_power.c); tw holds n + 2 */
void rxgpu_twiddle_table(const int16_t *sinewave, int n, uint32_t *tw);

/* kernel timing: bracket launches with events when profiling is on */
void rxgpu_prof_begin_on(const char *name, hipStream_t st);
void rxgpu_prof_end_on(const char *name, hipStream_t st);
#define rxgpu_prof_begin(name) rxgpu_prof_begin_on((name), rxgpu_hip_stream())
#define rxgpu_prof_end(name) rxgpu_prof_end_on((name), rxgpu_hip_stream())
/* drop a begun pair whose kernels were never enqueued (error paths) */
void rxgpu_prof_abort(void);
/* fold finished event pairs into the totals (call after a stream sync) */
void rxgpu_prof_collect(void);

#endif
