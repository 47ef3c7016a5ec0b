"""The bench-size checkers themselves (tests/parity_at_size.py), on CPU: fed the ORACLE PORT's results in place of the
device's they must say "equal" (the port is pinned against the reference, test_oracle_vs_ref.py), and a single flipped
sample / bin / carry must make them say "different".  Covers the forked-worker plumbing bench.py relies on."""
import ctypes as C

import numpy as np
import pytest

import parity_at_size as PA
import support


def _fm_seq(iq, n_blocks, block_len, tail, kw):
    """what fm_gpu_sequence returns, made by the oracle port"""
    O = support.oracle()
    st = support.oracle_fm_state(**kw)
    out = np.zeros(iq.size, np.int16)
    n1 = O.rxo_fm_stream(C.byref(st), support.ptr16(iq), n_blocks, block_len, support.ptr16(out), None)
    n2 = O.rxo_fm_stream(C.byref(st), support.ptr16(iq), tail, block_len, support.ptr16(out[n1:]), None)
    return {"got": out[:n1 + n2].copy(), "carry": PA._carry_tuple(st, kw), "deemph_avg": st.deemph_avg, "fixups": 0, "gpu_s": 0.0,
            "calls": n_blocks + tail}


FM_LEGS = {
    "headline": dict(downsample=118),
    "ds6": dict(downsample=6),
    "ds5": dict(downsample=5, rate_out=240000, deemph_a=19),
    "F7": dict(downsample_passes=7),
    "F9": dict(downsample_passes=3, comp_fir_size=9),
}


def test_fm_checkers_accept_the_port_and_catch_a_flipped_sample():
    block_len, n_blocks, tail = 2 * 8192, 6, 2
    iq = support.sig_fm(n_blocks * block_len // 2, seed=4711)
    legs = {lb: (kw, _fm_seq(iq, n_blocks, block_len, tail, kw)) for lb, kw in FM_LEGS.items()}
    v = PA.fm_check_many(iq, n_blocks, block_len, legs)
    assert all(r["parity_ok"] for r in v.values()), v
    assert all(r["parity_checked_samples"] == (n_blocks + tail) * block_len // 2 for r in v.values())
    # one output sample off in one leg, one carry off in another
    legs["ds6"][1]["got"][-3] ^= 1
    c = list(legs["F7"][1]["carry"])
    c[7] += 1
    legs["F7"][1]["carry"] = tuple(c)
    v = PA.fm_check_many(iq, n_blocks, block_len, legs)
    assert not v["ds6"]["parity_ok"] and v["ds6"]["parity_first_mismatch"] == legs["ds6"][1]["got"].size - 3
    assert not v["F7"]["parity_ok"] and v["F7"]["parity_first_mismatch"] == -1
    assert v["headline"]["parity_ok"] and v["ds5"]["parity_ok"] and v["F9"]["parity_ok"]


def _power_port(range_arg, window, boxcar, fir, h_in):
    import rx_tools_amd as R
    O = support.oracle()
    plan = R.plan_range(range_arg, 0.0, boxcar)
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs(window, n), R.sine_table(plan.bin_e)
    cfg = support.PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, boxcar, fir, 0, support.ptr32(wc), support.ptr16(sw))
    passes, tunes, buf_len = h_in.shape
    avg = np.zeros((tunes, n), np.int64)
    smp = np.zeros(tunes, np.int32)
    work = np.zeros(buf_len, np.int16)
    for p in range(passes):
        for t in range(tunes):
            s = C.c_int(int(smp[t]))
            O.rxo_power_tune(C.byref(cfg), support.ptr16(np.ascontiguousarray(h_in[p, t])), support.ptr16(work), support.ptr64(avg[t]), C.byref(s))
            smp[t] = s.value
    return plan, avg, smp


@pytest.mark.parametrize("range_arg,window,boxcar,fir,passes,amp", [
    ("24M:1.7G:1k", "hamming", 1, 0, 2, 32767),          # 599 tunes: children take tune ranges
    ("100M:100.1M:10", "rectangle", 0, 9, 5, 2000),      # one tune: children take pass ranges, partial sums added
])
def test_power_checker(range_arg, window, boxcar, fir, passes, amp):
    import rx_tools_amd as R
    plan = R.plan_range(range_arg, 0.0, boxcar)
    rng = np.random.default_rng(5)
    h_in = rng.integers(-amp, amp + 1, (passes, plan.tune_count, plan.buf_len), dtype=np.int16)
    _, avg, smp = _power_port(range_arg, window, boxcar, fir, h_in)
    r = PA.power_check(range_arg, 0.0, window, boxcar, fir, 0, h_in, avg, smp)
    assert r["parity_ok"] and r["parity_tunes_compared"] == plan.tune_count, r
    avg[plan.tune_count // 2, 17] += 1
    r = PA.power_check(range_arg, 0.0, window, boxcar, fir, 0, h_in, avg, smp)
    assert not r["parity_ok"] and r["parity_first_mismatch"] == [plan.tune_count // 2, 17]


def test_chan_checker():
    import rx_tools_amd as R
    bin_e, first_bin, n_ch, block_len, n_blocks = 6, 20, 16, 2 * 1024, 9
    sw = R.sine_table(bin_e)
    iq = support.sig_noise(n_blocks * block_len, seed=31, amp=3000)
    O = support.oracle()

    class Cfg(C.Structure):
        _fields_ = [("bin_e", C.c_int), ("first_bin", C.c_int), ("n_channels", C.c_int), ("custom_atan", C.c_int), ("sinewave", support.i16p)]
    cfg = Cfg(bin_e, first_bin, n_ch, 1, support.ptr16(sw))
    O.rxo_chan_block.argtypes = [C.c_void_p, support.i16p, C.c_int, support.intp, support.i16p, C.c_size_t]
    wpb = block_len // 2 >> bin_e
    got = np.zeros((n_ch, n_blocks * wpb), np.int16)
    pre = np.zeros(2 * n_ch, np.int32)
    tmp = np.zeros((n_ch, wpb), np.int16)
    for b in range(n_blocks):
        O.rxo_chan_block(C.byref(cfg), support.ptr16(iq[b * block_len:(b + 1) * block_len]), block_len, support.ptr32(pre), support.ptr16(tmp), wpb)
        got[:, b * wpb:(b + 1) * wpb] = tmp
    r = PA.chan_check(iq, n_blocks, block_len, bin_e, first_bin, n_ch, 1, sw, got, pre)
    assert r["parity_ok"] and r["parity_windows_compared"] == n_blocks * wpb, r
    assert r["parity_checker"].startswith("reference" if support.have_ref() else "port")
    got[3, 5 * wpb + 1] ^= 2
    r = PA.chan_check(iq, n_blocks, block_len, bin_e, first_bin, n_ch, 1, sw, got, pre)
    assert not r["parity_ok"] and r["parity_first_bad_block"] == 5
