/* dropin/rx_fm_unit.c -- rx_fm with its demodulator chain on the MI355X, built from the USER'S OWN rtl_fm.c, unmodified.
 *
 * The file is compiled as part of this translation unit (RXGPU_REF_RTL_FM_C = "<rx_tools>/src/rtl_fm.c", set by dropin/Makefile;
 * nothing of it is copied anywhere).  One declaration in front of it makes its full_demod (rtl_fm.c:759) a WEAK definition, so the
 * strong one in rx_fm_hooks.c wins at link time and demod_thread_fn's call (rtl_fm.c:923) lands there -- the INTEGRATION.md patch
 * done by the linker.  What follows the #include is ours and sees the file's statics (the -L counters, the global demod/dongle).
 *
 * rtlsdr_callback (rtl_fm.c:828) is `static` and called from inside the file: no linker can redirect it.  By default it stays the
 * reference's CPU code (a scale/rotate loop over one block, ~0.5 ms per MiB) and rxgpu_full_demod uploads lowpassed[];
 * `make PATCH=1` compiles a scratch copy in which that one call site (rtl_fm.c:899) is rewritten to rxgpu_dropin_callback below.
 */
#include <rxgpu.h>

struct demod_state;
void full_demod(struct demod_state *d) __attribute__((weak));       /* rtl_fm.c:759: overridable */
/* what `make PATCH=1` turns the call at rtl_fm.c:899 into (defined behind the #include: it needs the file's MAXIMUM_BUF_LENGTH) */
static void rxgpu_dropin_callback(int16_t *buf, uint32_t len, void *ctx);

#include RXGPU_REF_RTL_FM_C

/* ------------------------------------------------------------------ behind the reference's code */

static int rxgpu_dropin_fm_setup(void)
{
	int rc = rxgpu_init(-1);                            /* $RXGPU_DEVICE / $LOCAL_RANK / 0: no new flag */
	if (rc != RXGPU_OK)
		return rc;
	/* full_demod dispatches on d->mode_demod, a pointer into THIS file's functions */
	rxgpu_set_demod_functions((void *)&fm_demod, (void *)&am_demod, (void *)&usb_demod, (void *)&lsb_demod, (void *)&raw_demod);
	/* the structs are globals (rtl_fm.c:190-191): page-lock the members the drop-in DMAs, nothing to undo before exit */
	if (rxgpu_dropin_pin(&demod, &dongle) != RXGPU_OK)
		fprintf(stderr, "rx_fm (rxgpu): buffers stay pageable: %s\n", rxgpu_last_error());
	return RXGPU_OK;
}

/* PATCH=1: rxgpu_callback, with the dongle thread's read buffer (malloc'd at rtl_fm.c:873, MAXIMUM_BUF_LENGTH int16) page-locked the first
 * time it is seen -- together with buf16[] (rxgpu_dropin_pin above) that lets the block cross PCIe inside one launch (k_fm_prestage_zc).
 * The buffer lives as long as the thread; the registration goes with the process. */
static void rxgpu_dropin_callback(int16_t *buf, uint32_t len, void *ctx)
{
	static int16_t *seen;
	if (buf != seen) {
		seen = buf;
		if (rxgpu_pin(buf, (size_t)MAXIMUM_BUF_LENGTH * sizeof(int16_t)) != RXGPU_OK)
			fprintf(stderr, "rx_fm (rxgpu): the read buffer stays pageable: %s\n", rxgpu_last_error());
	}
	rxgpu_callback(buf, len, ctx);
}

/* -L (rtl_fm.c:792-807): the level line lives inside full_demod on the file-static counters above; this is that block behind the
 * device call.  `sr` is the squelch kernel's rms of the decimated block where squelch ran (a quiet block is zeroed afterwards, like
 * in the reference), the file's own rms() on the lowpassed[] the device handed back otherwise. */
static void rxgpu_dropin_fm_levels(struct demod_state *d)
{
	int sr = 0;
	if (!printLevels)
		return;
	if (rxgpu_dropin_block_rms(d, &sr) != RXGPU_OK)
		sr = rms(d->lowpassed, d->lp_len, 1);
	--printLevelNo;
	levelSum += sr;
	if (levelMax < sr)
		levelMax = sr;
	if (levelMaxMax < sr)
		levelMaxMax = sr;
	if (!printLevelNo) {
		printLevelNo = printLevels;
		fprintf(stderr, "%f, %d, %d, %d\n", (levelSum / printLevels), levelMax, levelMaxMax, d->squelch_level);
		levelMax = 0;
		levelSum = 0;
	}
}

static int rxgpu_dropin_ready;

/* before main(): bind the device, page-lock the structs and push one tiny block through the library, so that the HIP runtime and the
 * kernels' code object are loaded BEFORE the radio starts streaming -- the reference's hand-off between its dongle and demod threads
 * is a single lossy slot (rtl_fm.c:858-862, 921-924): a first full_demod that takes a second would cost real samples */
__attribute__((constructor)) static void rxgpu_dropin_fm_start(void)
{
	static int16_t warm_in[2 * 4096], warm_out[4096];
	rxgpu_fm_params p;
	rxgpu_fm_stream *s = NULL;
	size_t n = 0;
	int rate_in = 0;
	if (rxgpu_dropin_fm_setup() != RXGPU_OK)
		return;                                         /* no device: `rx_fm -h` still works; the first full_demod reports it */
	if (rxgpu_fm_params_init(&p, "wbfm", &rate_in) == RXGPU_OK && rxgpu_fm_stream_create(&s, &p, 1, 2 * 4096) == RXGPU_OK) {
		rxgpu_fm_stream_run_host(s, warm_in, 1, 2 * 4096, warm_out, 4096, &n, NULL);
		rxgpu_fm_stream_destroy(s);
	}
	rxgpu_dropin_ready = 1;
}

void rxgpu_dropin_full_demod(struct demod_state *d)
{
	if (!rxgpu_dropin_ready) {
		if (rxgpu_dropin_fm_setup() != RXGPU_OK) {
			fprintf(stderr, "rx_fm (rxgpu): %s\n", rxgpu_last_error());
			exit(1);
		}
		rxgpu_dropin_ready = 1;
	}
	rxgpu_full_demod(d);
	rxgpu_dropin_fm_levels(d);
}
