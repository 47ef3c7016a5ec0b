"""CPU: the oracle restatement (oracle/rx_oracle.c) against golden vectors that were produced by
EXECUTING THE REFERENCE'S OWN CODE (oracle/gen_golden.py -> tests/golden/*.npz)."""
import ctypes as C
import os

import numpy as np
import pytest

from support import (GOLDEN_DIR, oracle, oracle_fm_stream, PowerCfg, ptr16, ptr32, ptr64)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def test_scalar_kats():
    k = load("kats.npz")
    O = oracle()
    assert [O.rxo_fast_atan2(int(y), int(x)) for y, x in k["atan_yx"]] == list(k["atan_out"])
    assert [O.rxo_polar_disc_fast(*map(int, r)) for r in k["disc_in"]] == list(k["disc_fast"])
    assert [O.rxo_polar_discriminant(*map(int, r)) for r in k["disc_in"]] == list(k["disc_libm"])
    assert [O.rxo_fix_mpy(int(a), int(b)) for a, b in k["mpy_in"]] == list(k["mpy_out"])
    x = np.arange(-32768, 32768)
    assert np.array_equal(np.array([O.rxo_scale_sample(int(v)) for v in x], np.int16), k["scale_map"])


def test_survey_kats():
    """the hand-checkable vectors of SURVEY.md section 8(c), re-derived from the reference in this build"""
    O = oracle()
    assert O.rxo_polar_disc_fast(15104, -15104, -15104, 15104) == 12289      # int32 wrap in fast_atan2
    assert O.rxo_fix_mpy(-32768, -32768) == -32768
    buf = np.arange(1, 17, dtype=np.int16)
    O.rxo_rotate_90(ptr16(buf), 16)
    assert list(buf) == [1, 2, -4, 3, -5, -6, 8, -7, 9, 10, -12, 11, -13, -14, 16, -15]
    lp = np.arange(1, 17, dtype=np.int16)
    r, j, p = C.c_int(0), C.c_int(0), C.c_int(0)
    n = O.rxo_low_pass(ptr16(lp), 16, 3, C.byref(r), C.byref(j), C.byref(p))
    assert n == 4 and list(lp[:4]) == [9, 12, 27, 30] and (r.value, j.value, p.value) == (28, 30, 2)
    d = np.array([1000, 1000, 1000, -1000, -1000, 5, 6, 7, 0, 0, 32767, -32768], np.int16)
    avg = C.c_int(0)
    O.rxo_deemph(ptr16(d), 12, 13, C.byref(avg))
    assert list(d) == [77, 148, 214, 121, 35, 33, 31, 29, 27, 25, 2544, -172]
    y = np.arange(100, 2500, 100, dtype=np.int16)
    a, b = C.c_int(0), C.c_int(0)
    n = O.rxo_low_pass_real(ptr16(y), 24, 170000, 32000, C.byref(a), C.byref(b))
    assert n == 4 and list(y[:4]) == [420, 900, 1400, 2340] and (a.value, b.value) == (4700, 88000)
    sw = np.zeros(6, np.int16)
    O.rxo_sine_table(3, ptr16(sw))
    assert list(sw) == [0, 23170, 32767, 23170, 0, -23170]
    ramp = np.array([1000, -2000, 3000, 4000, -5000, 6000, 7000, -8000, 9000, 10000, -11000, 12000, 13000, -14000,
                     15000, 16000], np.int16)
    O.rxo_fix_fft(ptr16(ramp), 3, ptr16(sw))
    assert list(ramp) == [4000, 3000, 616, 1634, 1250, 5750, -8273, -5695, 500, -3000, 2384, -134, -750, -1750, 1273, -1805]


def _fm_names(z):
    return sorted({k.split("__")[0] for k in z.files})


def test_fm_cases_bit_exact():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from gen_golden import FM_CASES
    z = load("fm_cases.npz")
    params = {c[0]: c[4] for c in FM_CASES}
    assert _fm_names(z) == sorted(params)
    for name in _fm_names(z):
        iq, want = z[name + "__iq"], z[name + "__out"]
        got, lens, st = oracle_fm_stream(iq, int(z[name + "__block_len"][0]), **params[name])
        assert np.array_equal(got, want), name
        assert np.array_equal(lens, z[name + "__lens"]), name
        c = z[name + "__carry"]
        assert [st.now_r, st.now_j, st.prev_index, st.pre_r, st.pre_j, st.now_lpr, st.prev_lpr_index] == list(c[:7]), name
        hist = np.concatenate([np.ctypeslib.as_array(st.lp_i_hist).ravel(), np.ctypeslib.as_array(st.lp_q_hist).ravel(),
                               np.ctypeslib.as_array(st.droop_i_hist), np.ctypeslib.as_array(st.droop_q_hist)])
        assert np.array_equal(hist, z[name + "__hist"]), name


def test_config1_plumbing_sizes():
    """BASELINE config 1 (-M wbfm -s 240000, CPU plumbing): 1 s = 1.2 M samples in blocks of 131072
    (9 full + one of 20352) -> 32000 int16 out"""
    from support import sig_fm, oracle_fm_state, FmState
    O = oracle()
    iq = sig_fm(1200000, seed=5)
    st = oracle_fm_state(downsample=5, rate_out=240000, deemph_a=19)
    total = 0
    lp = np.zeros(262144, np.int16)
    out = np.zeros(131072, np.int16)
    pos = 0
    for n in [131072] * 9 + [20352]:
        blk = np.ascontiguousarray(iq[2 * pos:2 * (pos + n)])
        total += O.rxo_fm_block(C.byref(st), ptr16(blk), 2 * n, ptr16(lp), None, ptr16(out))
        pos += n
    assert total == 32000


def test_power_cases_bit_exact():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from gen_golden import POWER_CASES
    z = load("power_cases.npz")
    O = oracle()
    for name, rng, crop, window, flags, amp, passes, max_tunes in POWER_CASES:
        bin_e, buf_len, ds, ds_p, rate, n_all = map(int, z[name + "__meta"])
        n = 1 << bin_e
        data = z[name + "__in"]
        tunes = data.shape[1]
        wc = np.zeros(n, np.int32)
        O.rxo_window_coefs(window.encode(), n, ptr32(wc))
        assert np.array_equal(wc, z[name + "__window"]), name
        sw = np.zeros(max(1, n * 3 // 4), np.int16)
        O.rxo_sine_table(bin_e, ptr16(sw))
        cfg = PowerCfg(bin_e, buf_len, ds, ds_p, flags[0], flags[1], flags[2], ptr32(wc), ptr16(sw))
        avg = np.zeros((tunes, n), np.int64)
        samples = np.zeros(tunes, np.int32)
        work = np.zeros(buf_len, np.int16)
        for p in range(passes):
            for t in range(tunes):
                s = C.c_int(int(samples[t]))
                O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[p, t])), ptr16(work), ptr64(avg[t]), C.byref(s))
                samples[t] = s.value
        assert np.array_equal(avg, z[name + "__avg"]), name
        assert np.array_equal(samples, z[name + "__samples"]), name
        buf = C.create_string_buffer(1 << 20)
        for t in range(tunes):
            s = C.c_int(int(samples[t]))
            O.rxo_csv_row(buf, len(buf), int(z[name + "__freqs"][t]), rate, bin_e, ds, float(z[name + "__crop"][0]),
                          ptr64(avg[t]), C.byref(s))
            assert buf.value.decode().rstrip("\n") == str(z[name + "__csv"][t]), name
