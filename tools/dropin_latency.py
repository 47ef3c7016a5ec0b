"""tools/dropin_latency.py -- rxgpu_callback + rxgpu_full_demod on the reference's own structs, 1 MiB block after block (what bench.py reports as
host_fed.dropin_block_us): the single-block path against the general one ($RXGPU_DROPIN_FAST=0), alternating in one process"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rx_tools_amd as R
from rx_tools_amd.structs import DemodState, DongleState
L = R.lib(); R.check(L.rxgpu_init(0))
os.environ["RXGPU_DROPIN_TIMING"] = "1"
block_len = 2 * 131072
blk0 = R.synth.sig_fm(block_len // 2)
names = ["callback: H2D + pre-stage + D2H", "callback: hand-off", "full_demod: set-up", "full_demod: run", "full_demod: copy back + struct"]
import itertools
for rep in range(2):
    for fast, pinbuf, zc in (("1", 0, "1"), ("1", 1, "1"), ("1", 1, "0"), ("0", 0, "1"), ("0", 1, "0")):
        os.environ["RXGPU_DROPIN_FAST"] = fast
        os.environ["RXGPU_DROPIN_ZC"] = zc
        L.rxgpu_knobs_reload()
        d = DemodState()
        d.rate_in = d.rate_out = 170000
        d.rate_out2, d.custom_atan, d.deemph, d.deemph_a, d.downsample = 32000, 1, 1, 13, 118
        d.post_downsample, d.output_scale, d.squelch_hits, d.adc_block_const, d.rdc_block_const = 1, 1, 11, 9, 9
        libc = C.CDLL(None)
        libc.pthread_rwlock_init(C.byref(d, DemodState.rw.offset), None)
        libc.pthread_cond_init(C.byref(d, DemodState.ready.offset), None)
        libc.pthread_mutex_init(C.byref(d, DemodState.ready_m.offset), None)
        g = DongleState(); g.demod_target = C.pointer(d)
        blk = blk0.copy()
        R.check(L.rxgpu_dropin_pin(C.addressof(d), C.addressof(g)))
        if pinbuf: R.check(L.rxgpu_pin(blk.ctypes.data, blk.nbytes))      # the dongle thread's read buffer (rtl_fm.c:873), one more line of the patch
        for _ in range(10):
            L.rxgpu_callback(blk.ctypes.data, block_len, C.addressof(g)); L.rxgpu_full_demod(C.addressof(d))
        ph = (C.c_double * 7)(); L.rxgpu_dropin_timing(ph, 7)
        nb = 300
        t0 = time.perf_counter()
        for _ in range(nb):
            L.rxgpu_callback(blk.ctypes.data, block_len, C.addressof(g)); L.rxgpu_full_demod(C.addressof(d))
        t = (time.perf_counter() - t0) / nb
        L.rxgpu_dropin_timing(ph, 7)
        p = list(ph)
        print("fast=%s read buffer pinned=%d zero-copy=%s  pair %.1f us  %s" % (fast, pinbuf, zc if pinbuf else "-", t * 1e6, {n: round(v / (p[5] if i < 2 else p[6]), 1) for i, (n, v) in enumerate(zip(names, p[:5]))}), flush=True)
        R.check(L.rxgpu_dropin_unpin(C.addressof(d), C.addressof(g)))
        if pinbuf: L.rxgpu_unpin(blk.ctypes.data)
        L.rxgpu_dropin_release(C.addressof(d))
