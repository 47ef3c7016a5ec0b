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
hipStream_t rxgpu_hip_stream(void);

#define RX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
	return rxgpu_fail(RXGPU_ENODEV, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define RX_K(call) do { int e_ = (call); if (e_ != 0) \
	return rxgpu_fail(RXGPU_ENODEV, "%s launch failed: %s (%s:%d)", #call, hipGetErrorString((hipError_t)e_), __FILE__, __LINE__); } while (0)

/* kernel timing: bracket launches with events when profiling is on */
void rxgpu_prof_begin(const char *name);
void rxgpu_prof_end(const char *name);
/* fold finished event pairs into the totals (call after a stream sync) */
void rxgpu_prof_collect(void);

#endif
