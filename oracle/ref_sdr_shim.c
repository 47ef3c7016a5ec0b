/* oracle/_ref/libref_sdr.so -- TEST INFRASTRUCTURE ONLY.
 *
 * Wraps the reference's own rx_sdr translation unit (REF_RTL_SDR_C =
 * "/root/reference/src/rtl_sdr.c"), compiled where it lies and UNMODIFIED, by #including it.
 * Its output converters (rtl_sdr.c:354-391) are inline in main()'s read loop, so the only way to run
 * them is to run main(): ref_sdr_run() feeds the fake SoapySDR device from memory and lets
 * `rx_sdr -I <in> -F <out> file` write the converted stream.
 */
#define main rx_sdr_main
#include REF_RTL_SDR_C
#undef main
#include <stddef.h>

void soapy_fake_set_source(const int16_t *iq, size_t n_elems, size_t max_chunk);
void soapy_fake_set_elem_size(size_t bytes);
void soapy_fake_set_eos_hook(void (*fn)(void));

static void ref_sdr_eos(void) { do_exit = 1; }

/* in: n_elems complex elements in `in_fmt` ("CS16" or "CS12"); the converted stream is written to `path`.
 * chunk: elements handed out per readStream (0 = as asked).  Returns main()'s exit status. */
int ref_sdr_run(const void *in, size_t n_elems, const char *in_fmt, const char *out_fmt, const char *path, size_t chunk)
{
	char *argv[] = { "rx_sdr", "-I", (char *)in_fmt, "-F", (char *)out_fmt, (char *)path, NULL };
	do_exit = 0;
	samples_to_read = 0;
	optind = 1;
	soapy_fake_set_elem_size(strcmp(in_fmt, "CS12") == 0 ? 3 : 4);
	soapy_fake_set_source((const int16_t *)in, n_elems, chunk);
	soapy_fake_set_eos_hook(ref_sdr_eos);
	int rc = rx_sdr_main(6, argv);
	soapy_fake_set_elem_size(4);
	return rc;
}
