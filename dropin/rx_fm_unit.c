/* dropin/rx_fm_unit.c -- rx_fm with its demodulator chain on the MI355X, built from the USER'S OWN rtl_fm.c, unmodified.
 *
 * The file is compiled as part of this translation unit (RXGPU_REF_RTL_FM_C = "<rx_tools>/src/rtl_fm.c", set by dropin/Makefile;
 * nothing of it is copied anywhere).  One declaration in front of it makes its full_demod (rtl_fm.c:759) a WEAK definition, so the
 * strong one in rx_fm_hooks.c wins at link time and demod_thread_fn's call (rtl_fm.c:923) lands there -- the INTEGRATION.md patch
 * done by the linker.  What follows the #include is ours and sees the file's statics (the -L counters, the global demod/dongle).
 *
 * rtlsdr_callback (rtl_fm.c:828) is `static` and called from inside the file: no linker can redirect it.  By default it stays the
 * reference's CPU code (a scale/rotate loop over one block, ~0.5 ms per MiB) and rxgpu_full_demod uploads lowpassed[];
 * `make PATCH=1` compiles a scratch copy in which that one call site (rtl_fm.c:899) is rewritten to rxgpu_callback.
 */
#include <rxgpu.h>

struct demod_state;
void full_demod(struct demod_state *d) __attribute__((weak));       /* rtl_fm.c:759: overridable */

#include RXGPU_REF_RTL_FM_C

/* ------------------------------------------------------------------ behind the reference's code */

static void rxgpu_dropin_fm_setup(void)
{
	if (rxgpu_init(-1) != RXGPU_OK) {                   /* $RXGPU_DEVICE / $LOCAL_RANK / 0: no new flag */
		fprintf(stderr, "rx_fm (rxgpu): %s\n", rxgpu_last_error());
		exit(1);
	}
	/* full_demod dispatches on d->mode_demod, a pointer into THIS file's functions */
	rxgpu_set_demod_functions((void *)&fm_demod, (void *)&am_demod, (void *)&usb_demod, (void *)&lsb_demod, (void *)&raw_demod);
	/* the structs are globals (rtl_fm.c:190-191): page-lock the members the drop-in DMAs, nothing to undo before exit */
	if (rxgpu_dropin_pin(&demod, &dongle) != RXGPU_OK)
		fprintf(stderr, "rx_fm (rxgpu): buffers stay pageable: %s\n", rxgpu_last_error());
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

void rxgpu_dropin_full_demod(struct demod_state *d)
{
	static int ready;
	if (!ready) {
		rxgpu_dropin_fm_setup();
		ready = 1;
	}
	if (printLevels && !d->squelch_level && d->downsample_passes == 0 && d->lp_len == 0) {
		/* rms() of nothing: (int)NaN; keep the reference's own arithmetic for it */
	}
	rxgpu_full_demod(d);
	rxgpu_dropin_fm_levels(d);
}
