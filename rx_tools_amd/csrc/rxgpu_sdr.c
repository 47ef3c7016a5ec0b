/* rxgpu_sdr.c -- rx_sdr's -F output conversions and rx_fm's WAV header behind the C ABI (include/rxgpu.h). */
#include <stdlib.h>
#include <string.h>
#include "rxgpu_internal.h"

size_t rxgpu_sdr_in_bytes(int conversion, size_t n_elems)
{
	switch (conversion) {
	case RXGPU_SDR_CS16_TO_CU8: case RXGPU_SDR_CS16_TO_CS8: case RXGPU_SDR_CS16_TO_CF32: return n_elems * 4;
	case RXGPU_SDR_CS12_TO_CS16: return n_elems * 3;
	}
	return 0;
}

size_t rxgpu_sdr_out_bytes(int conversion, size_t n_elems)
{
	switch (conversion) {
	case RXGPU_SDR_CS16_TO_CU8: case RXGPU_SDR_CS16_TO_CS8: return n_elems * 2;
	case RXGPU_SDR_CS16_TO_CF32: return n_elems * 8;
	case RXGPU_SDR_CS12_TO_CS16: return n_elems * 4;
	}
	return 0;
}

static int convert_on(hipStream_t st, int conversion, const void *d_in, size_t n_elems, void *d_out)
{
	switch (conversion) {
	case RXGPU_SDR_CS16_TO_CU8:
		RX_K(rxk_sdr_cs16_to_8(st, (const int16_t *)d_in, 2ull * n_elems, 1, (uint8_t *)d_out));
		break;
	case RXGPU_SDR_CS16_TO_CS8:
		RX_K(rxk_sdr_cs16_to_8(st, (const int16_t *)d_in, 2ull * n_elems, 0, (uint8_t *)d_out));
		break;
	case RXGPU_SDR_CS16_TO_CF32:
		RX_K(rxk_sdr_cs16_to_cf32(st, (const int16_t *)d_in, 2ull * n_elems, (float *)d_out));
		break;
	case RXGPU_SDR_CS12_TO_CS16:
		RX_K(rxk_sdr_cs12_to_cs16(st, (const uint8_t *)d_in, n_elems, (int16_t *)d_out));
		break;
	default:
		return rxgpu_fail(RXGPU_EINVAL, "unknown rx_sdr conversion %d", conversion);
	}
	return RXGPU_OK;
}

int rxgpu_sdr_convert(int conversion, const void *d_in, size_t n_elems, void *d_out)
{
	int rc = rxgpu_ensure_init();
	if (rc)
		return rc;
	if (!rxgpu_sdr_out_bytes(conversion, 1))
		return rxgpu_fail(RXGPU_EINVAL, "unknown rx_sdr conversion %d", conversion);
	if (!n_elems)
		return RXGPU_OK;
	if (!d_in || !d_out)
		return rxgpu_fail(RXGPU_EINVAL, "null buffer");
	if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
		return rxgpu_fail(RXGPU_EINVAL, "rx_sdr conversion buffers must be 16-byte aligned");
	rxgpu_prof_begin("sdr_convert");
	rc = convert_on(rxgpu_hip_stream(), conversion, d_in, n_elems, d_out);
	rxgpu_prof_end("sdr_convert");
	return rc;
}

/* staging for the host entry point: grown on demand, kept for the life of the process (rx_sdr reads the same
 * block size every time) */
static void *stage_in, *stage_out;
static size_t stage_in_cap, stage_out_cap;

static int stage_reserve(void **p, size_t *cap, size_t need)
{
	if (need <= *cap)
		return RXGPU_OK;
	if (*p)
		(void)hipFree(*p);
	*p = NULL;
	*cap = 0;
	if (hipMalloc(p, need) != hipSuccess)
		return rxgpu_fail(RXGPU_ENOMEM, "hipMalloc(%zu) failed", need);
	*cap = need;
	return RXGPU_OK;
}

int rxgpu_sdr_convert_host(int conversion, const void *in, size_t n_elems, void *out)
{
	int rc = rxgpu_ensure_init();
	if (rc)
		return rc;
	const size_t nin = rxgpu_sdr_in_bytes(conversion, n_elems), nout = rxgpu_sdr_out_bytes(conversion, n_elems);
	if (!rxgpu_sdr_out_bytes(conversion, 1))
		return rxgpu_fail(RXGPU_EINVAL, "unknown rx_sdr conversion %d", conversion);
	if (!n_elems)
		return RXGPU_OK;
	if (!in || !out)
		return rxgpu_fail(RXGPU_EINVAL, "null buffer");
	if ((rc = stage_reserve(&stage_in, &stage_in_cap, nin)) || (rc = stage_reserve(&stage_out, &stage_out_cap, nout)))
		return rc;
	hipStream_t st = rxgpu_hip_stream();
	RX_HIP(hipMemcpyAsync(stage_in, in, nin, hipMemcpyHostToDevice, st));
	if ((rc = convert_on(st, conversion, stage_in, n_elems, stage_out)))
		return rc;
	RX_HIP(hipMemcpyAsync(out, stage_out, nout, hipMemcpyDeviceToHost, st));
	RX_HIP(hipStreamSynchronize(st));
	return RXGPU_OK;
}

/* rtl_fm.c:1174-1206 */
void rxgpu_wav_header(int rate, int raw_mode, unsigned char out[44])
{
	int s_rate = rate, b_rate = rate * 2;
	if (raw_mode)
		b_rate *= 2;
	memcpy(out, "RIFF\xFF\xFF\xFF\xFFWAVEfmt \x10\0\0\0\1\0", 22);
	out[22] = raw_mode ? 2 : 1; out[23] = 0;                       /* channels */
	for (int i = 0; i < 4; i++) {
		out[24 + i] = (unsigned char)((s_rate >> (8 * i)) & 0xFF);
		out[28 + i] = (unsigned char)((b_rate >> (8 * i)) & 0xFF);
	}
	out[32] = raw_mode ? 4 : 2; out[33] = 0;                       /* block align */
	out[34] = 0x10; out[35] = 0;                                   /* bits per channel */
	memcpy(out + 36, "data\xFF\xFF\xFF\xFF", 8);
}

/* What this box's HBM gives a plain stream in the converters' access shapes (no arithmetic): GB/s of read + written bytes.  The pool's
 * boxes differ by several per cent on pure reads and by more on mixed traffic; bench.py reports every HBM-bound leg
 * beside the ceiling measured in the same process. */
int rxgpu_diag_stream_rate(int mode, size_t units, int reps, double *gbs)
{
	int rc;
	if (mode < 0 || mode > 4 || !units || reps < 1 || !gbs)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_diag_stream_rate: bad arguments");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	const size_t in_b = units * (mode == 2 ? 8 : 16), out_b = (mode == 0 || mode == 4) ? 64 : units * (mode == 3 ? 8 : 16);
	void *d_in = NULL, *d_out = NULL;
	hipEvent_t e0 = NULL, e1 = NULL;
	hipStream_t st = rxgpu_hip_stream();
	if (hipMalloc(&d_in, in_b) != hipSuccess || hipMalloc(&d_out, out_b) != hipSuccess ||
	    hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
		hipFree(d_in); hipFree(d_out);
		if (e0) hipEventDestroy(e0);
		if (e1) hipEventDestroy(e1);
		return rxgpu_fail(RXGPU_ENOMEM, "rxgpu_diag_stream_rate: allocation failed");
	}
	hipMemsetAsync(d_in, 0x5a, in_b, st);
	rc = rxk_diag_stream(st, mode, d_in, units, d_out);                 /* warm */
	hipEventRecord(e0, st);
	for (int i = 0; i < reps && !rc; i++)
		rc = rxk_diag_stream(st, mode, d_in, units, d_out);
	hipEventRecord(e1, st);
	float ms = 0;
	if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0)
		rc = rc ? rc : -1;
	hipFree(d_in); hipFree(d_out);
	hipEventDestroy(e0); hipEventDestroy(e1);
	if (rc)
		return rxgpu_fail(RXGPU_ENODEV, "rxgpu_diag_stream_rate: launch failed");
	*gbs = (double)(in_b + ((mode == 0 || mode == 4) ? 0 : out_b)) * reps / (ms * 1e-3) / 1e9;
	return RXGPU_OK;
}
