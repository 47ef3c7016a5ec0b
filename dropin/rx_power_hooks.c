/* dropin/rx_power_hooks.c -- the strong definitions that take the place of rtl_power.c:670 and :774 in the linked rx_power
 * (see rx_power_unit.c). */
#include <stddef.h>
struct tuning_state;
void rxgpu_dropin_scanner(size_t channel);
void rxgpu_dropin_csv_dbm(struct tuning_state *ts);
void scanner(size_t channel) { rxgpu_dropin_scanner(channel); }
void csv_dbm(struct tuning_state *ts) { rxgpu_dropin_csv_dbm(ts); }
