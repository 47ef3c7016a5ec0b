"""-m gpu: the drop-in entry points on the reference's own structs (rxgpu_full_demod,
rxgpu_callback, rxgpu_scan, rxgpu_csv_dbm) against the oracle."""
import ctypes as C

import numpy as np
import pytest

import rx_tools_amd as R
from rx_tools_amd.structs import DemodState, DongleState, TuningState
from support import (oracle, oracle_fm_state, sig_fm, sig_noise, ptr16, ptr32, ptr64, PowerCfg)

pytestmark = pytest.mark.gpu


def fresh_demod(**kw):
    d = DemodState()
    d.rate_in = d.rate_out = kw.get("rate_out", 170000)
    d.rate_out2 = kw.get("rate_out2", 32000)
    d.custom_atan = kw.get("custom_atan", 1)
    d.deemph = kw.get("deemph", 1)
    d.deemph_a = kw.get("deemph_a", 13)
    d.downsample = kw.get("downsample", 6)
    d.downsample_passes = kw.get("downsample_passes", 0)
    d.comp_fir_size = kw.get("comp_fir_size", 0)
    d.post_downsample = kw.get("post_downsample", 1)
    d.dc_block_raw = kw.get("dc_block_raw", 0)
    d.rdc_block_const = kw.get("rdc_block_const", 9)
    d.output_scale = kw.get("output_scale", 1)
    d.squelch_level = kw.get("squelch_level", 0)
    d.squelch_hits = 11
    d.dc_block_audio = kw.get("dc_block_audio", 0)
    d.adc_block_const = 9
    libc = C.CDLL(None)
    libc.pthread_rwlock_init(C.byref(d, DemodState.rw.offset), None)
    libc.pthread_cond_init(C.byref(d, DemodState.ready.offset), None)
    libc.pthread_mutex_init(C.byref(d, DemodState.ready_m.offset), None)
    return d


@pytest.mark.parametrize("kw", [dict(downsample=6), dict(downsample=118), dict(downsample_passes=3, comp_fir_size=9),
                                dict(downsample=9, custom_atan=0, deemph=0, rate_out2=-1),
                                dict(downsample=6, squelch_level=30, dc_block_audio=1),
                                dict(downsample=8, post_downsample=4), dict(downsample=118, dc_block_raw=1),
                                dict(downsample_passes=3, dc_block_raw=1, rdc_block_const=2, post_downsample=2)])
def test_full_demod_and_callback_dropin(kw):
    """rxgpu_callback + rxgpu_full_demod, block after block on a struct demod_state, == the oracle's
    rtlsdr_callback pre-stage + full_demod, including lowpassed[], lp_len and every carry"""
    L, O = R.lib(), oracle()
    R.check(L.rxgpu_init(0))
    block_len, n_blocks = 16384, 5
    iq = sig_fm(n_blocks * block_len // 2, seed=42)
    d = fresh_demod(**kw)
    s = DongleState()
    s.demod_target = C.pointer(d)
    s.mute = 100
    st = oracle_fm_state(**kw)
    st.mute = 100
    # the side-car accumulator is keyed by the struct's address; a recycled address keeps its value
    L.rxgpu_deemph_state(C.addressof(d)).contents.value = 0
    lp = np.zeros(block_len, np.int16)
    want = np.zeros(block_len, np.int16)
    for b in range(n_blocks):
        blk = np.ascontiguousarray(iq[b * block_len:(b + 1) * block_len])
        lp_len = C.c_int(0)
        n_want = O.rxo_fm_block(C.byref(st), ptr16(blk.copy()), block_len, ptr16(lp), C.byref(lp_len), ptr16(want))
        buf = blk.copy()
        L.rxgpu_callback(buf.ctypes.data, block_len, C.addressof(s))
        assert d.lp_len == block_len and s.mute == 0
        L.rxgpu_full_demod(C.addressof(d))
        assert d.result_len == n_want
        assert np.array_equal(np.ctypeslib.as_array(d.result)[:n_want], want[:n_want])
        assert d.lp_len == lp_len.value
        assert np.array_equal(np.ctypeslib.as_array(d.lowpassed)[:d.lp_len], lp[:lp_len.value])
        assert (d.now_r, d.now_j, d.prev_index, d.pre_r, d.pre_j, d.now_lpr, d.prev_lpr_index) == \
            (st.now_r, st.now_j, st.prev_index, st.pre_r, st.pre_j, st.now_lpr, st.prev_lpr_index)
        assert bytes(d.lp_i_hist) == bytes(st.lp_i_hist) and bytes(d.lp_q_hist) == bytes(st.lp_q_hist)
        assert bytes(d.droop_i_hist) == bytes(st.droop_i_hist) and bytes(d.droop_q_hist) == bytes(st.droop_q_hist)
        assert L.rxgpu_deemph_state(C.addressof(d)).contents.value == st.deemph_avg
        assert (d.squelch_hits, d.dc_avg, d.dc_avgI, d.dc_avgQ) == (st.squelch_hits, st.dc_avg, st.dc_avgI, st.dc_avgQ)


@pytest.mark.parametrize("zc", [1, 0])
@pytest.mark.parametrize("rng,flags,window", [("24M:60M:1k", (1, 0, 0), "hamming"), ("100M:105M:1M", (1, 0, 1), "rectangle"),
                                              ("100M:100.1M:100", (0, 9, 0), "blackman")])
def test_scan_and_csv_dropin(rng, flags, window, zc, tmp_path, monkeypatch):
    """rxgpu_scan on an array of struct tuning_state (+ rxgpu_csv_dbm) == the oracle's scanner()/csv_dbm, two passes; deferred mode.
    zc = 1: the tunes' buf16 page-locked in place by the library, one launch reads them across PCIe (the default); 0: pinned staging + H2D"""
    L, O = R.lib(), oracle()
    monkeypatch.setenv("RXGPU_SCAN_ZC", str(zc))
    L.rxgpu_knobs_reload()
    L.rxgpu_scan_release()                                     # a sweep geometry cached by an earlier test decided its input path already
    R.check(L.rxgpu_scan_deferred(1))
    plan = R.plan_range(rng, 0.0, flags[0])
    tunes, n = min(plan.tune_count, 5), 1 << plan.bin_e
    wc, sw = R.window_coefs(window, n), R.sine_table(plan.bin_e)
    bufs = [np.zeros(plan.buf_len, np.int16) for _ in range(tunes)]
    avgs = [np.zeros(n, np.int64) for _ in range(tunes)]
    arr = (TuningState * tunes)()
    for t in range(tunes):
        arr[t] = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, ptr64(avgs[t]), 0, plan.downsample,
                             plan.downsample_passes, plan.crop, ptr16(bufs[t]), plan.buf_len)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, flags[0], flags[1], flags[2],
                   ptr32(wc), ptr16(sw))
    want_avg = np.zeros((tunes, n), np.int64)
    want_samples = np.zeros(tunes, np.int32)
    work = np.zeros(plan.buf_len, np.int16)
    for p in range(2):
        data = sig_noise(tunes * plan.buf_len, seed=50 + p, amp=2500).reshape(tunes, plan.buf_len)
        for t in range(tunes):
            bufs[t][:] = data[t]
            smp = C.c_int(int(want_samples[t]))
            O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[t])), ptr16(work), ptr64(want_avg[t]), C.byref(smp))
            want_samples[t] = smp.value
        R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, *flags))
        assert L.rxgpu_scan_zero_copy() == zc
        for t in range(tunes):
            bufs[t][:] = -1                                    # the call has read the buffers when it returns: the caller refills them at once
        if p == 0:
            # the sums stay on the device until somebody asks: the struct still shows what the caller left there
            assert not any(a.any() for a in avgs) and all(arr[t].samples == 0 for t in range(tunes))
            continue
        syncs = L.rxgpu_scan_syncs()
        R.check(L.rxgpu_scan_sync(arr, tunes))                # both sweeps in one download
        assert L.rxgpu_scan_syncs() == syncs + 1
        for t in range(tunes):
            assert np.array_equal(avgs[t], want_avg[t]) and arr[t].samples == want_samples[t]
    # a third sweep is brought home by rxgpu_csv_dbm itself (below); the caller's partial sums are added to, not replaced
    data = sig_noise(tunes * plan.buf_len, seed=52, amp=2500).reshape(tunes, plan.buf_len)
    for t in range(tunes):
        bufs[t][:] = data[t]
        smp = C.c_int(int(want_samples[t]))
        O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[t])), ptr16(work), ptr64(want_avg[t]), C.byref(smp))
        want_samples[t] = smp.value
    R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, *flags))
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fclose.argtypes = [C.c_void_p]
    f = libc.fopen(str(tmp_path / "o.csv").encode(), b"wb")
    rows = []
    buf = C.create_string_buffer(1 << 20)
    for t in range(tunes):
        smp = C.c_int(int(want_samples[t]))
        O.rxo_csv_row(buf, len(buf), arr[t].freq, plan.rate, plan.bin_e, plan.downsample, plan.crop, ptr64(want_avg[t]), C.byref(smp))
        rows.append(buf.value.decode())
        L.rxgpu_csv_dbm(C.byref(arr[t]), f)
    libc.fclose(f)
    assert open(str(tmp_path / "o.csv")).read() == "".join(rows)
    L.rxgpu_scan_release()                                     # bufs are page-locked in place: released before they die (rxgpu.h, LIFETIME)
    R.check(L.rxgpu_scan_deferred(0))


@pytest.mark.parametrize("peak", [0, 1])
def test_report_intervals_merge_into_rows_csv_dbm_cleared(peak, tmp_path):
    """rx_power's loop for several report intervals (rtl_power.c:1039-1050): sweeps, then csv_dbm for every tune -- which prints and ZEROES avg[]
    (rtl_power.c:815-817; rxgpu_csv_dbm does the same and remembers it).  The merge of the next interval finds rows it knows to be zero and writes
    the accumulators over them without reading them across the link; an interval in which only SOME rows were printed falls back to the reading
    merge, and the rows that were not printed keep adding up.  Every CSV row == the oracle's, every interval."""
    L, O = R.lib(), oracle()
    L.rxgpu_knobs_reload()
    L.rxgpu_scan_release()
    R.check(L.rxgpu_scan_deferred(1))
    plan = R.plan_range("24M:60M:1k", 0.0, 1)
    tunes, n = 5, 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, peak, ptr32(wc), ptr16(sw))
    work = np.zeros(plan.buf_len, np.int16)
    arr, bufs, avgs = _tuning_array(plan, tunes, n)
    want = np.zeros((tunes, n), np.int64)
    ws = np.zeros(tunes, np.int32)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fclose.argtypes = [C.c_void_p]
    buf = C.create_string_buffer(1 << 20)
    seed = 900
    for interval, printed in enumerate([range(tunes), range(tunes), range(3), range(tunes), range(tunes)]):
        for sweep in range(2):
            for t in range(tunes):
                bufs[t][:] = sig_noise(plan.buf_len, seed=seed, amp=2500)
                seed += 1
                smp = C.c_int(int(ws[t]))
                O.rxo_power_tune(C.byref(cfg), ptr16(bufs[t].copy()), ptr16(work), ptr64(want[t]), C.byref(smp))
                ws[t] = smp.value
            R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, peak))
        path = str(tmp_path / ("i%d.csv" % interval))
        f = libc.fopen(path.encode(), b"wb")
        rows = []
        for t in printed:
            smp = C.c_int(int(ws[t]))
            O.rxo_csv_row(buf, len(buf), arr[t].freq, plan.rate, plan.bin_e, plan.downsample, plan.crop, ptr64(want[t]), C.byref(smp))
            rows.append(buf.value.decode())
            L.rxgpu_csv_dbm(C.byref(arr[t]), f)
            want[t][:] = 0                                     # csv_dbm's own reset (rtl_power.c:815-817), the oracle's row function leaves it to the caller
            ws[t] = 0
        libc.fclose(f)
        assert open(path).read() == "".join(rows), "interval %d" % interval
        assert L.rxgpu_scan_sync_in_place() == 1
        for t in range(tunes):
            assert np.array_equal(avgs[t], want[t]) and arr[t].samples == ws[t], (interval, t)
    L.rxgpu_scan_release()
    R.check(L.rxgpu_scan_deferred(0))


def test_scan_on_a_sub_array_keeps_the_sweeps_registrations():
    """The drop-in's missed-read path calls rxgpu_scan(&tunes[i], j - i) between full sweeps (dropin/rx_power_unit.c): the sub-array is
    found in the table of page-locked buffers and read zero-copy through the rows it already has; a shorter call on buffers the table
    does not know is staged; the next full sweep is zero-copy again -- and every sum is the oracle's"""
    L, O = R.lib(), oracle()
    L.rxgpu_knobs_reload()
    L.rxgpu_scan_release()
    R.check(L.rxgpu_scan_deferred(0))
    plan = R.plan_range("24M:60M:1k", 0.0, 1)
    tunes, n = 6, 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0, ptr32(wc), ptr16(sw))
    work = np.zeros(plan.buf_len, np.int16)
    arr, bufs, avgs = _tuning_array(plan, tunes, n)
    other, obufs, oavgs = _tuning_array(plan, 2, n)
    want = np.zeros((tunes, n), np.int64)
    ws = np.zeros(tunes, np.int32)
    owant, ows = np.zeros((2, n), np.int64), np.zeros(2, np.int32)

    def feed(bs, idx, w_avg, w_smp, seed):
        for k, t in enumerate(idx):
            bs[t][:] = sig_noise(plan.buf_len, seed=seed + k, amp=2500)
            smp = C.c_int(int(w_smp[t]))
            O.rxo_power_tune(C.byref(cfg), ptr16(bs[t].copy()), ptr16(work), ptr64(w_avg[t]), C.byref(smp))
            w_smp[t] = smp.value

    feed(bufs, range(tunes), want, ws, 300)
    R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert L.rxgpu_scan_zero_copy() == 1
    feed(bufs, [2, 3, 4], want, ws, 320)                       # tunes 2..4 again, as after a missed read
    sub = C.cast(C.byref(arr, 2 * C.sizeof(TuningState)), C.POINTER(TuningState))
    R.check(L.rxgpu_scan(sub, 3, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert L.rxgpu_scan_zero_copy() == 1
    feed(obufs, range(2), owant, ows, 340)                     # two tunes through buffers of their own: staged, the table stays
    R.check(L.rxgpu_scan(other, 2, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert L.rxgpu_scan_zero_copy() == 0
    feed(bufs, range(tunes), want, ws, 360)
    R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert L.rxgpu_scan_zero_copy() == 1
    for t in range(tunes):
        assert np.array_equal(avgs[t], want[t]) and arr[t].samples == ws[t]
    for t in range(2):
        assert np.array_equal(oavgs[t], owant[t]) and other[t].samples == ows[t]
    L.rxgpu_scan_release()


def _tuning_array(plan, tunes, n):
    bufs = [np.zeros(plan.buf_len, np.int16) for _ in range(tunes)]
    avgs = [np.zeros(n, np.int64) for _ in range(tunes)]
    arr = (TuningState * tunes)()
    for t in range(tunes):
        arr[t] = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, ptr64(avgs[t]), 0, plan.downsample,
                             plan.downsample_passes, plan.crop, ptr16(bufs[t]), plan.buf_len)
    return arr, bufs, avgs


def test_scan_default_is_current_after_every_call_and_never_writes_a_stale_array(capfd):
    """Default mode: ts->avg[] / ts->samples are the CPU's after EVERY rxgpu_scan and no pointer is kept.  Deferred mode: a pending
    interval is never written through an array the running call was not handed -- another array fails the scan until the first is
    synced, and a release with sums pending drops them (stderr) instead of touching the old array."""
    L, O = R.lib(), oracle()
    plan = R.plan_range("24M:60M:1k", 0.0, 1)
    tunes, n = 4, 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0, ptr32(wc), ptr16(sw))
    work = np.zeros(plan.buf_len, np.int16)

    def feed(arr_bufs, want_avg, want_samples, seed):
        data = sig_noise(tunes * plan.buf_len, seed=seed, amp=2500).reshape(tunes, plan.buf_len)
        for t in range(tunes):
            arr_bufs[t][:] = data[t]
            smp = C.c_int(int(want_samples[t]))
            O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[t])), ptr16(work), ptr64(want_avg[t]), C.byref(smp))
            want_samples[t] = smp.value

    R.check(L.rxgpu_scan_deferred(0))
    arr, bufs, avgs = _tuning_array(plan, tunes, n)
    want_avg, want_samples = np.zeros((tunes, n), np.int64), np.zeros(tunes, np.int32)
    for p in range(3):
        feed(bufs, want_avg, want_samples, 70 + p)
        R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
        for t in range(tunes):                                 # current after each call, nothing to sync
            assert np.array_equal(avgs[t], want_avg[t]) and arr[t].samples == want_samples[t]
    R.check(L.rxgpu_scan_sync(arr, tunes))                     # a no-op in this mode
    # another array right away: fine, nothing was pending
    arr2, bufs2, avgs2 = _tuning_array(plan, tunes, n)
    want2, ws2 = np.zeros((tunes, n), np.int64), np.zeros(tunes, np.int32)
    feed(bufs2, want2, ws2, 90)
    R.check(L.rxgpu_scan(arr2, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert all(np.array_equal(avgs2[t], want2[t]) for t in range(tunes))

    # deferred: the interval of `arr` is pending; arr2 must be refused and `arr` left alone until its own sync
    R.check(L.rxgpu_scan_deferred(1))
    feed(bufs, want_avg, want_samples, 95)
    before = [a.copy() for a in avgs]
    R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    assert L.rxgpu_scan(arr2, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0) == -2
    assert b"rxgpu_scan_sync" in L.rxgpu_last_error()
    assert L.rxgpu_scan_sync(arr2, tunes) == -2 and L.rxgpu_scan_sync(None, 0) == -2
    assert L.rxgpu_scan_deferred(0) == -2
    assert all(np.array_equal(avgs[t], before[t]) for t in range(tunes))       # nothing written behind the caller's back
    R.check(L.rxgpu_scan_sync(arr, tunes))
    assert all(np.array_equal(avgs[t], want_avg[t]) and arr[t].samples == want_samples[t] for t in range(tunes))
    # pending at release: dropped with a note, the array is not touched
    feed(bufs, want_avg, want_samples, 96)
    before = [a.copy() for a in avgs]
    R.check(L.rxgpu_scan(arr, tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    capfd.readouterr()
    L.rxgpu_shutdown()                                         # releases the drop-in caches
    R.check(L.rxgpu_init(0))
    assert "never merged" in capfd.readouterr().err
    assert all(np.array_equal(avgs[t], before[t]) for t in range(tunes))
    R.check(L.rxgpu_scan_deferred(0))


def test_dropin_handoff_stays_on_the_device_unless_invalidated():
    """rxgpu_callback leaves the pre-staged block in HBM and rxgpu_full_demod uses that copy; a caller that edits
    d->lowpassed in between says so with rxgpu_dropin_invalidate and gets the edited block demodulated"""
    L, O = R.lib(), oracle()
    R.check(L.rxgpu_init(0))
    block_len = 16384
    iq = sig_fm(2 * block_len // 2, seed=43)
    kw = dict(downsample=6)
    for edit in (False, True):
        d = fresh_demod(**kw)
        s = DongleState()
        s.demod_target = C.pointer(d)
        st = oracle_fm_state(**kw)
        L.rxgpu_deemph_state(C.addressof(d)).contents.value = 0
        lp = np.zeros(block_len, np.int16)
        want = np.zeros(block_len, np.int16)
        if edit:
            R.check(L.rxgpu_dropin_pin(C.addressof(d), C.addressof(s)))       # the optional in-place DMA of the struct members
        for b in range(2):
            blk = np.ascontiguousarray(iq[b * block_len:(b + 1) * block_len])
            raw = blk.copy()                                                   # the callback may zero its head (mute): not the test's copy
            L.rxgpu_callback(raw.ctypes.data, block_len, C.addressof(s))
            pre = np.ctypeslib.as_array(d.lowpassed)[:block_len].copy()
            if edit:
                pre[100:200] = 0
                np.ctypeslib.as_array(d.lowpassed)[:block_len] = pre
                L.rxgpu_dropin_invalidate(C.addressof(d))
            lp[:block_len] = pre
            lp_len = C.c_int(block_len)
            n_want = O.rxo_fm_full_demod(C.byref(st), ptr16(lp), C.byref(lp_len), ptr16(want))
            L.rxgpu_full_demod(C.addressof(d))
            assert d.result_len == n_want
            assert np.array_equal(np.ctypeslib.as_array(d.result)[:n_want], want[:n_want])
            assert d.lp_len == lp_len.value and np.array_equal(np.ctypeslib.as_array(d.lowpassed)[:d.lp_len], lp[:lp_len.value])
        if edit:
            R.check(L.rxgpu_dropin_unpin(C.addressof(d), C.addressof(s)))


def test_dropin_release_frees_the_side_car_slot():
    """sixteen side-car slots exist (stream object, callback staging, de-emphasis accumulator per demod_state); a caller that creates and
    destroys demod_states hands each slot back with rxgpu_dropin_release -- forty of them in a row, every one demodulating like the oracle"""
    L, O = R.lib(), oracle()
    R.check(L.rxgpu_init(0))
    block_len = 4096
    iq = sig_fm(block_len // 2, seed=7)
    kw = dict(downsample=6)
    want = np.zeros(block_len, np.int16)
    for i in range(40):
        d = fresh_demod(**kw)
        s = DongleState()
        s.demod_target = C.pointer(d)
        st = oracle_fm_state(**kw)
        L.rxgpu_deemph_state(C.addressof(d)).contents.value = 0
        raw = iq.copy()
        L.rxgpu_callback(raw.ctypes.data, block_len, C.addressof(s))
        lp = np.ctypeslib.as_array(d.lowpassed)[:block_len].copy()
        lp_len = C.c_int(block_len)
        n_want = O.rxo_fm_full_demod(C.byref(st), ptr16(lp), C.byref(lp_len), ptr16(want))
        L.rxgpu_full_demod(C.addressof(d))
        assert d.result_len == n_want and np.array_equal(np.ctypeslib.as_array(d.result)[:n_want], want[:n_want]), i
        sr = C.c_int(0)
        assert L.rxgpu_dropin_block_rms(C.addressof(d), C.byref(sr)) == -3          # no squelch on that block: the caller's own rms() applies
        R.check(L.rxgpu_dropin_release(C.addressof(d)))
        assert L.rxgpu_dropin_release(C.addressof(d)) == -2                          # nothing left to release
        del s, d


def test_rdc_on_an_empty_read_exits_like_the_header_says():
    """-E rdc on an EMPTY read: the reference divides by len/2 == 0 (rtl_fm.c:710-711); rxgpu_callback says so on stderr and exit(1)s
    (the reference's failure convention) instead of publishing an empty block"""
    import os
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, rx_tools_amd as R\n"
            "from test_gpu_dropin import fresh_demod\n"
            "from rx_tools_amd.structs import DongleState\n"
            "L = R.lib(); R.check(L.rxgpu_init(0))\n"
            "d = fresh_demod(dc_block_raw=1); s = DongleState(); s.demod_target = C.pointer(d)\n"
            "buf = np.zeros(16, np.int16)\n"
            "L.rxgpu_callback(buf.ctypes.data, 0, C.addressof(s))\n"
            "print('survived')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 1 and b"survived" not in out.stdout and b"rdc" in out.stderr, (out.returncode, out.stderr[-600:])


def _same_state(d, r, L, check_result0):
    n = r.result_len
    assert d.result_len == n and d.lp_len == r.lp_len
    assert np.array_equal(np.ctypeslib.as_array(d.result)[:n], np.ctypeslib.as_array(r.result)[:n])
    assert np.array_equal(np.ctypeslib.as_array(d.lowpassed)[:d.lp_len], np.ctypeslib.as_array(r.lowpassed)[:r.lp_len])
    if check_result0:
        assert d.result[0] == r.result[0]
    for f in ("now_r", "now_j", "prev_index", "pre_r", "pre_j", "now_lpr", "prev_lpr_index", "squelch_hits", "dc_avg", "dc_avgI", "dc_avgQ"):
        assert getattr(d, f) == getattr(r, f), f
    assert bytes(d.lp_i_hist) == bytes(r.lp_i_hist) and bytes(d.lp_q_hist) == bytes(r.lp_q_hist)
    assert bytes(d.droop_i_hist) == bytes(r.droop_i_hist) and bytes(d.droop_q_hist) == bytes(r.droop_q_hist)


ANY_LENGTH_PARAMS = [
    dict(downsample=6), dict(downsample=118), dict(downsample=2000, custom_atan=0), dict(downsample=118, squelch_level=40),
    dict(downsample=9, mode=1, deemph=0), dict(downsample=300, mode=4),
    dict(downsample_passes=3), dict(downsample_passes=3, comp_fir_size=9), dict(downsample_passes=7), dict(downsample_passes=7, squelch_level=50),
    dict(downsample_passes=4, mode=1, output_scale=2, deemph=0), dict(downsample_passes=5, mode=4), dict(downsample_passes=10, comp_fir_size=9),
    dict(downsample_passes=2, dc_block_raw=1, rdc_block_const=3),
    # -o on demodulated lengths that are no multiple of the step: low_pass_simple (rtl_fm.c:373-387) hands on the complete groups only
    dict(downsample=6, post_downsample=4), dict(downsample=118, post_downsample=3, custom_atan=1), dict(downsample_passes=3, post_downsample=2),
    dict(downsample=9, mode=1, deemph=0, post_downsample=5), dict(downsample_passes=2, comp_fir_size=9, post_downsample=16, rate_out2=-1),
]


FAST_BLOCK_PARAMS = [
    dict(downsample=118), dict(downsample=118, custom_atan=1), dict(downsample=118, custom_atan=3), dict(downsample=8, custom_atan=1, deemph_a=2),
    dict(downsample=30, deemph_a=64), dict(downsample=30, deemph_a=65), dict(downsample=50, deemph_a=1), dict(downsample=118, deemph_a=200, custom_atan=1),
    dict(downsample=118, deemph=0), dict(downsample=40, rate_out2=-1), dict(downsample=40, deemph=0, rate_out2=-1, custom_atan=1),
    dict(downsample=25, rate_out=96000, rate_out2=48000, deemph_a=7), dict(downsample=1000, custom_atan=1), dict(downsample=2000, custom_atan=0),
    dict(downsample=20, deemph_a=5), dict(downsample=20, deemph_a=6, custom_atan=1), dict(downsample=12, deemph_a=8),     # the 24-bit division's smallest a
]


@pytest.mark.parametrize("env", [{}, {"RXGPU_FLAG_ALL": "2"}, {"RXGPU_DROPIN_FAST": "0"}])
@pytest.mark.parametrize("kw", FAST_BLOCK_PARAMS)
def test_dropin_single_block_path(kw, env, monkeypatch):
    """the drop-in's single blocks of the plain FM chain (k_fm_block_dd + k_ch_audio: two launches, carries as kernel arguments, one copy back)
    against the reference itself, length after length; with every libm sample handed to the host WRONG ($RXGPU_FLAG_ALL=2: the block's first
    sample is patched and the audio stages run again from the row as demodulated; -A std blocks have more such samples than the block header
    holds and take the general path); and the same calls through the general path ($RXGPU_DROPIN_FAST=0)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    R.lib().rxgpu_knobs_reload()
    try:
        test_dropin_takes_every_block_length_the_reference_takes(kw)
    finally:
        monkeypatch.undo()
        R.lib().rxgpu_knobs_reload()


@pytest.mark.parametrize("fast", ["1", "0"])
@pytest.mark.parametrize("signal", ["fm", "noise", "steps"])
@pytest.mark.parametrize("kw", [dict(downsample=12, deemph_a=9), dict(downsample=12, deemph_a=63, custom_atan=1), dict(downsample=16, deemph_a=33, rate_out2=-1),
                                dict(downsample=30, deemph_a=15), dict(downsample=118, deemph_a=13), dict(downsample=11, deemph_a=21, rate_out=96000, rate_out2=8000)])
def test_dropin_short_rows_on_hostile_signals(kw, signal, fast, monkeypatch):
    """k_fm_row_audio (a one-block run's de-emphasis + low_pass_real: chunked warm-up from the ends of the int16 range, candidate tracking, the chunk
    tables walked by one thread, replay) against the reference itself, block after block of every length, for a = 9 ... 63 on an FM signal, on full-scale
    noise and on a carrier that slews the filter from rail to rail -- through the two-launch path and through the general one"""
    monkeypatch.setenv("RXGPU_DROPIN_FAST", fast)
    R.lib().rxgpu_knobs_reload()
    try:
        test_dropin_takes_every_block_length_the_reference_takes(kw, signal=signal)
    finally:
        monkeypatch.undo()
        R.lib().rxgpu_knobs_reload()


@pytest.mark.parametrize("kw", [dict(downsample=118), dict(downsample=6), dict(downsample_passes=3, comp_fir_size=9), dict(downsample=118, offset_tuning=1)])
def test_dropin_callback_zero_copy(kw):
    """buf16[] and the read buffer page-locked (rxgpu_dropin_pin + rxgpu_pin, what INTEGRATION.md's patch does): rxgpu_callback then reads the raw
    block from and writes the scaled one to host memory inside one launch (k_fm_prestage_zc); every length against the reference, the device
    addresses looked up again whenever a registration changed"""
    test_dropin_takes_every_block_length_the_reference_takes(kw, pin=True)


@pytest.mark.parametrize("kw", ANY_LENGTH_PARAMS)
def test_dropin_takes_every_block_length_the_reference_takes(kw, pin=False, signal="fm"):
    """readStream may return ANY element count (rtl_fm.c:894-899): the drop-in, call after call with a different length -- primes,
    two samples, reads shorter than the decimation, an empty read -- against the reference ITSELF (oracle/_ref: its own
    rtlsdr_callback + full_demod on its own struct), including the shapes where the C reads pre_r/pre_j from in front of
    lowpassed[] (the tail of d->thread: both structs carry the same value there) and where a -F block leaves an odd lp_len."""
    from support import have_ref, ref_fm, ref_fm_reset
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    L = R.lib()
    R.check(L.rxgpu_init(0))
    F = ref_fm()
    r, rs = ref_fm_reset(F, **kw)
    d = fresh_demod(**kw)
    d.mode_demod = r.mode_demod
    L.rxgpu_set_demod_functions(F.ref_fm_fn(0), F.ref_fm_fn(2), F.ref_fm_fn(3), F.ref_fm_fn(4), F.ref_fm_fn(1))
    d.thread = r.thread = 0x00007F3A5C1E9700
    s = DongleState()
    s.demod_target = C.pointer(d)
    s.offset_tuning = kw.get("offset_tuning", 0)
    L.rxgpu_deemph_state(C.addressof(d)).contents.value = 0
    rng = np.random.default_rng(abs(hash(str(sorted(kw.items())))) % (1 << 31))
    lens = [2 * 4099, 2 * 97, 2, 2 * 33, 0, 2 * 7, 2 * 65537, 2 * 131071, 2 * 1, 2 * 129, 2 * 100000, 4, 6, 2 * 1009, 2 * 131072, 2 * 12]
    lens += [2 * int(v) for v in rng.integers(1, 3000, 6)]
    if kw.get("dc_block_raw"):
        lens = [v for v in lens if v]                     # the reference divides by zero on an empty read with -E rdc (rtl_fm.c:711)
    iq = sig_fm(sum(lens) // 2 + 8, seed=77)
    if signal == "noise":                                 # full-scale noise: the discriminator's output jumps over the whole int16 range from sample to sample
        iq = sig_noise(sum(lens) + 16, seed=78, amp=32768)
    elif signal == "steps":                               # a carrier whose phase steps by +-90 degrees in long runs: the de-emphasis state slews from rail to rail
        ph = np.repeat(np.random.default_rng(5).integers(0, 2, sum(lens) // 2 // 37 + 2) * 2 - 1, 37)[:sum(lens) // 2 + 8]
        ang = np.cumsum(ph * (np.pi / 2))
        iq = np.empty(2 * len(ang), np.int16)
        iq[0::2] = np.round(30000 * np.cos(ang)).astype(np.int16)
        iq[1::2] = np.round(30000 * np.sin(ang)).astype(np.int16)
    pos = 0
    if pin:
        R.check(L.rxgpu_dropin_pin(C.addressof(d), C.addressof(s)))
        rbuf = np.zeros(262144, np.int16)                 # the dongle thread's read buffer: MAXIMUM_BUF_LENGTH int16, page-locked as a whole
        R.check(L.rxgpu_pin(rbuf.ctypes.data, rbuf.nbytes))
    try:
        for ln in lens:
            blk = np.ascontiguousarray(iq[pos:pos + max(ln, 2)])
            pos += ln
            a, b = blk.copy(), blk.copy()
            if pin:
                rbuf[:len(blk)] = blk
                b = rbuf
            F.ref_fm_callback(ptr16(a), C.c_uint32(ln), C.byref(rs))
            L.rxgpu_callback(b.ctypes.data, ln, C.addressof(s))
            assert d.lp_len == r.lp_len == ln
            assert np.array_equal(np.ctypeslib.as_array(d.lowpassed)[:ln], np.ctypeslib.as_array(r.lowpassed)[:ln])
            F.full_demod(C.byref(r))
            L.rxgpu_full_demod(C.addressof(d))
            _same_state(d, r, L, kw.get("mode", 0) == 0 and r.lp_len < 2)
    finally:
        L.rxgpu_set_demod_functions(None, None, None, None, None)
        if pin:
            L.rxgpu_unpin(rbuf.ctypes.data)
            L.rxgpu_dropin_unpin(C.addressof(d), C.addressof(s))


@pytest.mark.parametrize("fail_after,kw", [(7, dict(downsample=118)), (12, dict(downsample_passes=3, comp_fir_size=9)), (3, dict(downsample=6, squelch_level=30))])
def test_device_error_mid_stream_ends_the_process_without_deadlock(fail_after, kw):
    """A launch that fails in the middle of a stream ($RXGPU_FAIL_AFTER: the n-th launch of the host code reports hipErrorLaunchFailure) while
    the application's other thread -- the dongle thread, calling rxgpu_callback and taking d->rw for its hand-off -- keeps running: the
    process must end (rxgpu_fatal: one line on stderr, device released, _exit(1)) within seconds, with NOTHING on stdout (the audio stream)
    and an atexit handler that would block forever never run.  The hook exists in the TEST build of the host files only (librxgpu_fi.so,
    $RXGPU_LIB_FLAVOUR=fi: same kernels, -DRXGPU_FAULT_INJECT); the shipped library has no such switch."""
    import os
    import subprocess
    import sys
    code = (
        "import atexit, ctypes as C, os, sys, threading, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, rx_tools_amd as R\n"
        "from test_gpu_dropin import fresh_demod\n"
        "from rx_tools_amd.structs import DongleState\n"
        "from support import sig_fm\n"
        "atexit.register(lambda: time.sleep(3600))          # what a driver's exit handler waiting for its own thread would be\n"
        "L = R.lib(); R.check(L.rxgpu_init(0))\n"
        "kw = %r\n"
        "d = fresh_demod(**kw); s = DongleState(); s.demod_target = C.pointer(d)\n"
        "d2 = fresh_demod(**kw); s2 = DongleState(); s2.demod_target = C.pointer(d2)\n"
        "blk = sig_fm(131072, seed=3)\n"
        "def dongle():\n"
        "    b = blk.copy()\n"
        "    while True:\n"
        "        L.rxgpu_callback(b.ctypes.data, b.size, C.addressof(s2))\n"
        "t = threading.Thread(target=dongle, daemon=True); t.start()\n"
        "for i in range(3):\n"
        "    b = blk.copy(); L.rxgpu_callback(b.ctypes.data, b.size, C.addressof(s)); L.rxgpu_full_demod(C.addressof(d))\n"
        "sys.stderr.write('warm\\n'); sys.stderr.flush()\n"
        "os.environ['RXGPU_FAIL_AFTER'] = %r\n"
        "L.rxgpu_knobs_reload()\n"
        "for i in range(200):\n"
        "    b = blk.copy(); L.rxgpu_callback(b.ctypes.data, b.size, C.addressof(s)); L.rxgpu_full_demod(C.addressof(d))\n"
        "print('survived')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), kw, str(fail_after))
    t0 = __import__("time").time()
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                         env=dict(os.environ, RXGPU_LIB_FLAVOUR="fi"))
    took = __import__("time").time() - t0
    err = out.stderr.decode()
    assert out.returncode == 1, (out.returncode, err[-800:])
    assert out.stdout == b"", out.stdout[-200:]
    assert "warm" in err and err.count("rxgpu: ") == 1 and "failed" in err.split("rxgpu: ")[1], err[-800:]       # "... launch failed" (full_demod) or "device pre-stage failed" (callback)
    assert took < 60


def test_dropin_release_while_the_callback_runs():
    """rxgpu_dropin_release against a callback of the same demod_state on another thread (round 4's advisory: the slot index was taken
    before the lock, the locked mutex was copied and cleared): the dongle thread hammers rxgpu_callback, the main thread releases the
    side-car again and again -- every block that comes out afterwards is still the oracle's, and nothing crashes or deadlocks"""
    import threading
    L, O = R.lib(), oracle()
    R.check(L.rxgpu_init(0))
    block_len = 8192
    iq = sig_fm(block_len // 2, seed=9)
    kw = dict(downsample=6)
    d = fresh_demod(**kw)
    s = DongleState()
    s.demod_target = C.pointer(d)
    stop = threading.Event()
    calls = [0]

    def dongle():
        b = iq.copy()
        while not stop.is_set():
            L.rxgpu_callback(b.ctypes.data, block_len, C.addressof(s))
            calls[0] += 1
    t = threading.Thread(target=dongle)
    t.start()
    try:
        for _ in range(300):
            L.rxgpu_dropin_release(C.addressof(d))              # OK or "no side-car": both happen
    finally:
        stop.set()
        t.join(60)
    assert not t.is_alive() and calls[0] > 0
    # the state the callback publishes is still right: one more block, demodulated like the oracle from a fresh state
    L.rxgpu_dropin_release(C.addressof(d))
    d2 = fresh_demod(**kw)
    s2 = DongleState()
    s2.demod_target = C.pointer(d2)
    st = oracle_fm_state(**kw)
    L.rxgpu_deemph_state(C.addressof(d2)).contents.value = 0
    raw = iq.copy()
    L.rxgpu_callback(raw.ctypes.data, block_len, C.addressof(s2))
    lp = np.ctypeslib.as_array(d2.lowpassed)[:block_len].copy()
    lp_len = C.c_int(block_len)
    want = np.zeros(block_len, np.int16)
    n_want = O.rxo_fm_full_demod(C.byref(st), ptr16(lp), C.byref(lp_len), ptr16(want))
    L.rxgpu_full_demod(C.addressof(d2))
    assert d2.result_len == n_want and np.array_equal(np.ctypeslib.as_array(d2.result)[:n_want], want[:n_want])
    L.rxgpu_dropin_release(C.addressof(d2))
