/* dropin/rx_fm_hooks.c -- the strong definition that takes the place of rtl_fm.c:759 in the linked rx_fm (see rx_fm_unit.c). */
struct demod_state;
void rxgpu_dropin_full_demod(struct demod_state *d);
void full_demod(struct demod_state *d) { rxgpu_dropin_full_demod(d); }
