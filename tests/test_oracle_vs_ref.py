"""CPU, this container only: the oracle restatement against the reference's own code compiled
unmodified (oracle/_ref), on seeded random and adversarial inputs beyond the golden fixtures."""
import ctypes as C

import numpy as np
import pytest

from support import (have_ref, ref_fm, ref_power, oracle, ref_fm_stream, oracle_fm_stream, sig_fm, sig_noise,
                     sig_alternating, PowerCfg, ptr16, ptr32, ptr64)

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")]


@pytest.mark.parametrize("params", [
    dict(downsample=6), dict(downsample=118), dict(downsample=5, rate_out=240000, deemph_a=19),
    dict(downsample_passes=3), dict(downsample_passes=3, comp_fir_size=9), dict(downsample_passes=7),
    dict(downsample=9, custom_atan=0), dict(downsample=118, offset_tuning=1), dict(downsample=6, mute=4096),
    dict(downsample=6, mode=1, output_scale=3, deemph=0), dict(downsample=6, mode=2, output_scale=2),
    dict(downsample=6, mode=3, deemph=0, rate_out2=-1), dict(downsample=6, mode=4),
    dict(downsample=10, custom_atan=2), dict(downsample=10, custom_atan=3),
    dict(downsample=6, squelch_level=40), dict(downsample=6, squelch_level=2000), dict(downsample=6, dc_block_audio=1),
    dict(downsample_passes=3, dc_block_audio=1, squelch_level=100, mode=1, output_scale=1),
    dict(downsample=4, post_downsample=4), dict(downsample=2, post_downsample=2, dc_block_audio=1), dict(downsample_passes=3, post_downsample=4),
    dict(downsample=118, dc_block_raw=1), dict(downsample=6, dc_block_raw=1, rdc_block_const=2, dc_avgI=11, dc_avgQ=-3),
    dict(downsample_passes=3, comp_fir_size=9, dc_block_raw=1, post_downsample=2),
])
def test_fm_stream_matches_reference(params):
    L = ref_fm()
    sigs = [sig_fm(4 * 8192), sig_noise(8 * 8192), sig_alternating(8 * 8192), np.zeros(8 * 8192, np.int16)]
    for iq in sigs:
        for bl in (8192, 16384, 4096 + 8):
            if params.get("downsample_passes") and (bl // 2) % (1 << params["downsample_passes"]):
                continue
            if params.get("post_downsample", 1) > 1:
                # low_pass_simple needs whole groups (rtl_fm.c:374); it reads stale memory otherwise
                per = (bl // 2) >> params["downsample_passes"] if params.get("downsample_passes") else (bl // 2) // params["downsample"]
                if ((bl // 2) % params.get("downsample", 1) and not params.get("downsample_passes")) or per % params["post_downsample"]:
                    continue
            a, la, d = ref_fm_stream(L, iq, bl, **params)
            b, lb, st = oracle_fm_stream(iq, bl, **params)
            assert np.array_equal(a, b) and np.array_equal(la, lb)
            assert (d.now_r, d.now_j, d.prev_index, d.pre_r, d.pre_j, d.now_lpr, d.prev_lpr_index, d.squelch_hits, d.dc_avg,
                    d.dc_avgI, d.dc_avgQ) == \
                (st.now_r, st.now_j, st.prev_index, st.pre_r, st.pre_j, st.now_lpr, st.prev_lpr_index, st.squelch_hits, st.dc_avg,
                 st.dc_avgI, st.dc_avgQ)


def test_fm_stream_random_parameter_sweep():
    """seeded random parameter sets (every switch of the chain at once) through the reference and the restatement"""
    L = ref_fm()
    rng = np.random.default_rng(20260925)
    for case in range(40):
        passes = int(rng.choice([0, 0, 0, 1, 2, 3, 5]))
        block = int(rng.choice([2048, 4096, 8192, 16384])) * 2
        n = block // 2
        ds = int(rng.choice([1, 2, 3, 4, 6, 7, 16, 118, 250]))
        post = int(rng.choice([1, 1, 1, 2, 4]))
        params = dict(downsample=ds, downsample_passes=passes, comp_fir_size=int(rng.choice([0, 9])),
                      custom_atan=int(rng.integers(0, 4)), deemph=int(rng.integers(0, 2)), deemph_a=int(rng.choice([2, 3, 8, 13, 19, 40, 100])),
                      rate_out=int(rng.choice([170000, 240000, 48000])), rate_out2=int(rng.choice([-1, 32000, 48000, 8000])),
                      offset_tuning=int(rng.integers(0, 2)), mode=int(rng.choice([0, 0, 0, 1, 2, 3, 4])), output_scale=int(rng.integers(1, 4)),
                      squelch_level=int(rng.choice([0, 0, 50, 3000])), dc_block_audio=int(rng.integers(0, 2)),
                      dc_block_raw=int(rng.integers(0, 2)), rdc_block_const=int(rng.integers(1, 12)), post_downsample=post)
        if params["rate_out2"] > params["rate_out"]:
            params["rate_out2"] = -1
        per = (n >> passes) if passes else n // ds
        if post > 1 and ((not passes and n % ds) or per % post):
            params["post_downsample"] = 1
        iq = [sig_fm(3 * n, seed=case), sig_noise(3 * block, seed=case), sig_noise(3 * block, seed=case, amp=300)][case % 3]
        a, la, d = ref_fm_stream(L, iq, block, **params)
        b, lb, st = oracle_fm_stream(iq, block, **params)
        assert np.array_equal(a, b) and np.array_equal(la, lb), (case, params)
        assert (d.now_r, d.now_j, d.prev_index, d.pre_r, d.pre_j, d.now_lpr, d.prev_lpr_index, d.squelch_hits, d.dc_avg, d.dc_avgI, d.dc_avgQ) == \
            (st.now_r, st.now_j, st.prev_index, st.pre_r, st.pre_j, st.now_lpr, st.prev_lpr_index, st.squelch_hits, st.dc_avg, st.dc_avgI, st.dc_avgQ), (case, params)


def test_struct_layout_matches_reference():
    from rx_tools_amd.structs import DemodState, DongleState, TuningState
    L, P = ref_fm(), ref_power()
    assert C.sizeof(DemodState) == L.ref_fm_sizeof_demod_state() == 1049160
    assert C.sizeof(DongleState) == L.ref_fm_sizeof_dongle_state()
    assert C.sizeof(TuningState) == P.ref_power_sizeof_tuning_state()
    names = ['lowpassed', 'lp_len', 'lp_i_hist', 'result', 'result_len', 'rate_in', 'now_r', 'downsample', 'deemph',
             'now_lpr', 'mode_demod', 'rw', 'output_target', 'droop_i_hist', 'dc_block_audio']
    for i, n in enumerate(names):
        assert getattr(DemodState, n).offset == L.ref_fm_offsetof(i), n
    for i, n in zip((15, 16, 17, 18), ('buf16', 'mute', 'demod_target', 'offset_tuning')):
        assert getattr(DongleState, n).offset == L.ref_fm_offsetof(i), n


@pytest.mark.parametrize("rng,crop,window,flags,amp,passes", [
    ("88M:108M:125k", 0.0, "rectangle", (1, 0, 0), 100, 2),
    ("88M:108M:125k", 0.2, "blackman-harris", (1, 0, 1), 3000, 2),
    ("100M:100.1M:100", 0.0, "rectangle", (1, 0, 0), 2000, 1),
    ("100M:100.1M:100", 0.0, "youssef", (0, 9, 0), 2000, 1),
    ("100M:101M:1k", 0.0, "bartlett", (1, 0, 0), 500, 2),
    ("100M:110M:1M", 0.0, "rectangle", (1, 0, 1), 5000, 2),
    ("24M:50M:1k", 0.0, "hann-poisson", (1, 0, 0), 32768, 1),
])
def test_power_scan_matches_reference(rng, crop, window, flags, amp, passes, capfd):
    from rx_tools_amd.structs import TuningState
    P, O = ref_power(), oracle()
    P.ref_power_set_flags(*flags)
    n = P.ref_power_setup(rng.encode(), crop, window.encode())
    tunes = (TuningState * n).from_address(P.ref_power_tunes())
    t0 = tunes[0]
    buf_len, N = t0.buf_len, 1 << t0.bin_e
    data = sig_noise(passes * n * buf_len, seed=777, amp=amp)
    P.ref_power_scan(ptr16(data), passes)
    ref_avg = np.stack([np.ctypeslib.as_array(tunes[i].avg, (N,)).copy() for i in range(n)])
    ref_samples = [tunes[i].samples for i in range(n)]
    sw = np.zeros(max(1, N * 3 // 4), np.int16)
    O.rxo_sine_table(t0.bin_e, ptr16(sw))
    wc = np.zeros(N, np.int32)
    O.rxo_window_coefs(window.encode(), N, ptr32(wc))
    assert np.array_equal(wc, np.ctypeslib.as_array(P.ref_power_window_coefs(), (N,)))
    cfg = PowerCfg(t0.bin_e, buf_len, t0.downsample, t0.downsample_passes, flags[0], flags[1], flags[2], ptr32(wc), ptr16(sw))
    avg = np.zeros((n, N), np.int64)
    samples = np.zeros(n, np.int32)
    work = np.zeros(buf_len, np.int16)
    d3 = data.reshape(passes, n, buf_len)
    for p in range(passes):
        for i in range(n):
            s = C.c_int(int(samples[i]))
            O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(d3[p, i])), ptr16(work), ptr64(avg[i]), C.byref(s))
            samples[i] = s.value
    assert np.array_equal(avg, ref_avg) and list(samples) == ref_samples


RAGGED_PARAMS = [
    dict(downsample=6), dict(downsample=118), dict(downsample=7, custom_atan=0, deemph=0, rate_out2=-1),
    dict(downsample_passes=1), dict(downsample_passes=3), dict(downsample_passes=3, comp_fir_size=9), dict(downsample_passes=7),
    dict(downsample_passes=4, mode=1, output_scale=2), dict(downsample_passes=2, mode=4), dict(downsample_passes=5, mode=2, deemph=0),
    dict(downsample_passes=3, squelch_level=40), dict(downsample_passes=2, comp_fir_size=9, squelch_level=3000, dc_block_audio=1),
    dict(downsample=6, squelch_level=60, mode=3),
]


def ragged_lengths(seed, count=10):
    """int16 counts of callback blocks: readStream may return any element count (rtl_fm.c:894-899)"""
    rng = np.random.default_rng(seed)
    fixed = [2 * p for p in (97, 251, 1009, 4099, 8191, 65537, 131071)]
    return fixed + [2 * int(v) for v in rng.integers(300, 131072, count)]


def benign(params, n, prev_index):
    """does a block of n samples leave every read of the reference inside lowpassed[] (no lp[-1], lp[-2])?"""
    p = params.get("downsample_passes", 0)
    if p:
        return ((2 * n) >> p) >= 2
    return (prev_index + n) // params["downsample"] >= 1


@pytest.mark.parametrize("params", RAGGED_PARAMS)
def test_fm_ragged_blocks_match_reference(params):
    """one stream cut into callback blocks of arbitrary even lengths, a different one every call: the restatement follows the
    reference through every odd `lp_len >> i` of the -F cascade (I and Q yielding different counts, an odd final lp_len,
    pre_r/pre_j taken from lp[lp_len-2], lp[lp_len-1]) and through low_pass windows that straddle several short blocks"""
    from rx_tools_amd.structs import DemodState
    from support import ref_fm_reset, oracle_fm_state
    L, O = ref_fm(), oracle()
    lens = ragged_lengths(11) + [2 * v for v in (20, 33, 64, 100, 7, 129)]
    iq = sig_fm(sum(lens) // 2, seed=5)
    d, s = ref_fm_reset(L, **params)
    st = oracle_fm_state(**params)
    pos = 0
    lp = np.zeros(262144, np.int16)
    out = np.zeros(262144, np.int16)
    for ln in lens:
        if not benign(params, ln // 2, st.prev_index):
            continue
        blk = np.ascontiguousarray(iq[pos:pos + ln])
        pos += ln
        buf = blk.copy()
        L.ref_fm_callback(ptr16(buf), C.c_uint32(ln), C.byref(s))
        L.full_demod(C.byref(d))
        lp_len = C.c_int(0)
        n = O.rxo_fm_block(C.byref(st), ptr16(blk.copy()), ln, ptr16(lp), C.byref(lp_len), ptr16(out))
        assert n == d.result_len and lp_len.value == d.lp_len, (ln, params)
        assert np.array_equal(np.ctypeslib.as_array(d.result)[:n], out[:n]), (ln, params)
        assert np.array_equal(np.ctypeslib.as_array(d.lowpassed)[:d.lp_len], lp[:lp_len.value]), (ln, params)
        assert (d.now_r, d.now_j, d.prev_index, d.pre_r, d.pre_j, d.now_lpr, d.prev_lpr_index, d.squelch_hits, d.dc_avg) == \
            (st.now_r, st.now_j, st.prev_index, st.pre_r, st.pre_j, st.now_lpr, st.prev_lpr_index, st.squelch_hits, st.dc_avg), (ln, params)
        assert bytes(d.lp_i_hist) == bytes(st.lp_i_hist) and bytes(d.lp_q_hist) == bytes(st.lp_q_hist)
        assert bytes(d.droop_i_hist) == bytes(st.droop_i_hist) and bytes(d.droop_q_hist) == bytes(st.droop_q_hist)


def test_plan_range_random_sweep_matches_frequency_range(capfd):
    """rxgpu_power_plan_range (closed form: bisections over the reference's own expressions) == the reference's frequency_range
    (rtl_power.c:431-543, counting loops) on 300 random ranges: narrow spans (downsampling, boxcar and fifth_order flavour), wide spans
    (many hops), cropped edges, giant bins, spans no hop count covers"""
    import rx_tools_amd as R
    from rx_tools_amd.structs import TuningState
    P = ref_power()
    rnd = np.random.RandomState(2468)
    checked = 0
    for case in range(300):
        kind = case % 5
        lower = int(rnd.randint(24, 1700)) * 10 ** 6 + int(rnd.randint(0, 10 ** 6))
        if kind == 0:
            span = int(rnd.randint(2_000, 900_000))                       # one hop, downsampled
        elif kind == 1:
            span = int(rnd.randint(900_000, 6_000_000))                   # around the MINIMUM / MAXIMUM_RATE edges
        elif kind == 2:
            span = int(rnd.randint(6_000_000, 400_000_000))               # many hops
        elif kind == 3:
            span = int(rnd.choice([2_800_000, 2_800_001, 5_600_000, 999_999, 1_000_000, 1_000_001, 2_799_999]))
        else:
            span = int(rnd.randint(100_000_000, 2_000_000_000)) * int(rnd.choice([1, 3]))   # some beyond 1499 hops of 2.8 MHz
        binw = int(rnd.choice([1, 10, 100, 125, 333, 1000, 5000, 12_500, 100_000, 999_999, 1_000_000, 2_000_000, int(rnd.randint(1, 300_000))]))
        crop = float(rnd.choice([0.0, 0.0, 0.1, 0.25, 0.5, 0.73]))
        boxcar = int(rnd.randint(0, 2))
        rng = "%d:%d:%d" % (lower, lower + span, binw)
        try:
            p = R.plan_range(rng, crop, boxcar)
        except R.RxGpuError:
            continue                                                     # the reference exit(1)s on those (unsupported bandwidth / too wide)
        if p.tune_count * (8 << p.bin_e) > (48 << 20) or p.tune_count * p.buf_len * 4 > (48 << 20):
            continue                                                     # frequency_range mallocs every tune's avg[] and buf16 and never frees
        P.ref_power_set_flags(boxcar, 0, 0)
        tc = P.ref_power_setup(rng.encode(), crop, b"rectangle")
        capfd.readouterr()
        assert tc == p.tune_count, (rng, crop, boxcar, tc, p.tune_count)
        if tc == 0:
            continue
        arr = (TuningState * tc).from_address(P.ref_power_tunes())
        t0 = arr[0]
        got = (p.bin_e, p.buf_len, p.downsample, p.downsample_passes, p.rate, p.first_freq, p.crop)
        want = (t0.bin_e, t0.buf_len, t0.downsample, t0.downsample_passes, t0.rate, t0.freq, t0.crop)
        assert got == want, (rng, crop, boxcar, got, want)
        if tc > 1:
            assert arr[1].freq - arr[0].freq == p.bw_seen and arr[tc - 1].freq == p.first_freq + (tc - 1) * p.bw_seen, rng
        checked += 1
    assert checked >= 150, checked
