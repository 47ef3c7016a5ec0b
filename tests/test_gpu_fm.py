"""-m gpu: the rx_fm HIP path against the oracle, through the C ABI (librxgpu.so)."""
import ctypes as C

import numpy as np
import pytest

import rx_tools_amd as R

from support import (oracle_fm_stream, oracle_fm_state, sig_fm, sig_noise, sig_alternating, oracle)

pytestmark = pytest.mark.gpu


def _signals(n_int16):
    return {
        "fm": sig_fm(n_int16 // 2),
        "noise_full": sig_noise(n_int16, seed=777),
        "noise_small": sig_noise(n_int16, seed=5, amp=700),
        "alternating": sig_alternating(n_int16),
        "zeros": np.zeros(n_int16, np.int16),
        "dc": np.full(n_int16, 3000, np.int16),
    }


def _check(iq, block_len, n_runs=1, pipelined=False, carry_in=None, **params):
    from gpu_support import gpu_fm_stream, carry_tuple, carry_from_oracle_state
    want, want_lens, st = oracle_fm_stream(iq, block_len, **params)
    got, got_lens, carry, _ = gpu_fm_stream(iq, block_len, n_runs=n_runs, pipelined=pipelined, carry=carry_in, **params)
    assert len(got) == len(want)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at %d: got %d want %d (%d bad of %d)" % (
        bad[0], got[bad[0]], want[bad[0]], bad.size, len(want))
    assert np.array_equal(got_lens, want_lens)
    assert carry_tuple(carry)[:8] == carry_tuple(carry_from_oracle_state(st))[:8]
    return carry, st


@pytest.mark.parametrize("sig", ["fm", "noise_full", "noise_small", "alternating", "zeros", "dc"])
@pytest.mark.parametrize("ds", [118, 6])
def test_low_pass_chain_bit_exact(sig, ds):
    """config 2 geometry (ds=118) and the -M wbfm default (ds=6), 16 blocks of 8192 complex"""
    iq = _signals(16 * 16384)[sig]
    _check(iq, 16384, downsample=ds)


@pytest.mark.parametrize("ds", [4, 8, 9, 13, 16, 17, 18, 19, 24, 25, 31, 32, 33, 48, 63, 64, 65, 100, 200, 500, 2000])
@pytest.mark.parametrize("sig", ["fm", "noise_full"])
def test_every_decimator_regime(ds, sig):
    """the span geometry of the small-decimation kernel changes at ds = 18/19 (256 or 64 windows per unit of a span), the
    discriminator's 24-bit remainder multiply stops at 16, the small kernel hands over to the prefix-scan kernel at 33 and that
    one switches to 16-byte slot records at 64: a pipelined sequence of blocks that are no multiple of ds on each of them"""
    block_len = 2 * 10240 + 8
    iq = _signals(13 * block_len)[sig]
    _check(iq, block_len, n_runs=3, pipelined=True, downsample=ds)


@pytest.mark.parametrize("tw", ["1", "2", "3", "5", None])
@pytest.mark.parametrize("ds", [4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32])
def test_lane_owned_window_decimator(ds, tw, monkeypatch):
    """k_fm_decimate_lane (a lane owns W whole windows, no barrier): every ds it takes, pipelined runs whose lengths are no multiple of
    ds * 4 so that the carried phase p0 walks through all four 16-byte alignments (one template instance each), waves of 1 / 2 / 3 / 5 / 4
    tiles ($RXGPU_DL_TW: halo lanes, the SGPR carries across tiles, and the vmcnt arithmetic of the two-stage ring at the start and the
    end of a wave's walk), the tiled pcm layout (de-emphasis + resampler behind it) and the linear one (deemph=0), offset tuning (no
    rotate16_90)"""
    if tw:
        monkeypatch.setenv("RXGPU_DL_TW", tw)
    block_len = 2 * (4096 + 4 * 3)              # 4108 samples per block: 4108 % ds walks p0
    iq = sig_noise(15 * block_len, seed=100 + ds)
    for extra in (dict(), dict(deemph=0), dict(offset_tuning=1)):
        _check(iq, block_len, n_runs=5, pipelined=True, downsample=ds, **extra)
    if tw is None:
        iq = sig_fm(40 * 16384, seed=ds)        # longer runs: several waves of four tiles, whole tiles of the tiled layout
        _check(iq, 2 * 16384, n_runs=3, pipelined=True, downsample=ds)


@pytest.mark.parametrize("ds,block_len", [(118, 2 * 131072), (5, 2 * 20352), (7, 4096 + 8), (1, 4096), (3, 8192),
                                          (250, 8192), (118, 2 * 1180)])
def test_low_pass_geometries(ds, block_len):
    """fast and generic decimator, block lengths that are / are not multiples of 8 int16"""
    n_blocks = 3 if block_len > 100000 else 9
    iq = sig_fm(n_blocks * block_len // 2, seed=99)
    _check(iq, block_len, downsample=ds)


@pytest.mark.parametrize("ds", [4, 127, 128, 512, 513, 966, 967, 1000, 4096, 8191, 8192])
def test_low_pass_decimator_regimes(ds):
    """the fast decimator's regimes: 24-bit / 32-bit window-edge division (switch near ds = 966), sparse / full
    lowpassed[] (ds <= 512), one window per several lanes up to one window per half span (ds = 8192)"""
    block_len = 2 * 32768
    for sig, n_runs in (("fm", 1), ("noise_full", 2)):
        iq = _signals(6 * block_len)[sig]
        _check(iq, block_len, n_runs=n_runs, downsample=ds)
        _check(iq, block_len, n_runs=n_runs, pipelined=True, downsample=ds, offset_tuning=1)


def test_config1_wbfm_240k():
    """BASELINE config 1: -M wbfm -s 240000 (ds=5, deemph_a=19), 1 s = 1.2 M samples as
    9 blocks of 131072 -- through the device path (the CPU plumbing case is in test_oracle)"""
    iq = sig_fm(9 * 131072, seed=4)
    _check(iq, 2 * 131072, downsample=5, rate_out=240000, deemph_a=19)


@pytest.mark.parametrize("n_runs", [2, 5])
def test_carry_across_runs(n_runs):
    """several rxgpu_fm_stream_run calls == one: every carry crosses the call boundary"""
    iq = sig_fm(20 * 8192, seed=1)
    _check(iq, 16384, n_runs=n_runs, downsample=118)
    _check(sig_noise(20 * 16384, seed=3), 16384, n_runs=n_runs, downsample=6)


@pytest.mark.parametrize("params", [dict(downsample=118), dict(downsample=6), dict(downsample=7, deemph=0),
                                    dict(downsample_passes=3, comp_fir_size=9), dict(downsample=3)])
def test_pipelined_runs(params):
    """rxgpu_fm_stream_run_async x7 + one wait == the oracle: carries chained on the device, the
    decimator of run r+1 overlapping the audio stages of run r on a second stream"""
    for iq in (sig_fm(28 * 8192, seed=77), sig_noise(28 * 16384, seed=78)):
        carry, st = _check(iq, 16384, n_runs=7, pipelined=True, **params)
        if params.get("downsample_passes"):
            from gpu_support import carry_tuple, carry_from_oracle_state
            p = params["downsample_passes"]
            want, got = carry_tuple(carry_from_oracle_state(st)), carry_tuple(carry)
            assert got[8][:12 * p] == want[8][:12 * p] and got[9][:12 * p] == want[9][:12 * p]
            assert got[10] == want[10] and got[11] == want[11]


@pytest.mark.parametrize("params", [
    dict(downsample=118, deemph=0),
    dict(downsample=118, rate_out2=-1),
    dict(downsample=118, deemph=0, rate_out2=-1),
    dict(downsample=10, custom_atan=0),
    dict(downsample=118, offset_tuning=1),
    dict(downsample=20, deemph_a=40),          # 64-lane candidate groups
    dict(downsample=20, deemph_a=200),         # serial de-emphasis kernel
    dict(downsample=20, deemph_a=1),
    dict(downsample=20, deemph_a=2),
    dict(downsample=20, deemph_a=5),           # odd, below the signed 24-bit step's range (needs a >= 9): the generic step
    dict(downsample=20, deemph_a=7),
    dict(downsample=8, rate_out=48000, rate_out2=48000),
])
def test_stage_switches(params):
    iq = sig_fm(8 * 8192, seed=7)
    _check(iq, 16384, **params)
    _check(sig_noise(8 * 16384, seed=8, amp=2000), 16384, **params)


@pytest.mark.parametrize("passes,fir", [(3, 0), (3, 9), (7, 0), (1, 9), (5, 9)])
@pytest.mark.parametrize("sig", ["fm", "noise_full", "zeros"])
def test_fifth_order_chain_bit_exact(passes, fir, sig):
    """-F path: fifth_order cascade (+ droop FIR), including the block-seam rule"""
    iq = _signals(6 * 16384)[sig]
    carry, st = _check(iq, 16384, downsample_passes=passes, comp_fir_size=fir)
    from gpu_support import carry_tuple, carry_from_oracle_state
    want = carry_tuple(carry_from_oracle_state(st))
    got = carry_tuple(carry)
    # histories of the passes that ran
    assert got[8][:12 * passes] == want[8][:12 * passes]
    assert got[9][:12 * passes] == want[9][:12 * passes]
    if fir == 9:
        assert got[10] == want[10] and got[11] == want[11]


@pytest.mark.parametrize("passes,fir,n", [(7, 0, 16384), (7, 9, 32768), (6, 0, 32768), (5, 9, 16384), (4, 0, 32768),
                                          (7, 0, 131072), (7, 9, 262144), (8, 0, 262144), (9, 9, 131072)])   # a third fused group: pass 7 at 1/64 rate
def test_fifth_order_two_fused_stages(passes, fir, n):
    """blocks long enough for the second fused group (passes 4-6 on the 1/8-rate stream, int arithmetic)"""
    from gpu_support import carry_tuple, carry_from_oracle_state
    for iq in (sig_fm(3 * n, seed=55), sig_noise(6 * n, seed=56), np.full(6 * n, 32767, np.int16)):
        carry, st = _check(iq, 2 * n, downsample_passes=passes, comp_fir_size=fir)
        want, got = carry_tuple(carry_from_oracle_state(st)), carry_tuple(carry)
        assert got[8][:12 * passes] == want[8][:12 * passes] and got[9][:12 * passes] == want[9][:12 * passes]
    iq = sig_noise(8 * n, seed=57, amp=30000)
    _check(iq, 2 * n, n_runs=2, pipelined=True, downsample_passes=passes, comp_fir_size=fir, offset_tuning=1)


@pytest.mark.parametrize("passes", [3, 4, 5, 7, 9])
def test_first_group_depths(passes):
    """the first group of a cascade: three passes are the whole chain in the register kernel (k_fm_fifth_regn<., 3, DD>), four or more put four
    into it (<., 4, 0>) and the rest into LDS-tiled groups of up to three -- seams and histories included, on inputs that reach the 32-bit sums"""
    from gpu_support import carry_tuple, carry_from_oracle_state
    n = 131072
    # +32767 everywhere without the rotation: every tap is +128 after the scale and the level-4 tap sum reaches 2^15 (32-bit sums there)
    for iq, off in ((sig_fm(3 * n, seed=61), 0), (sig_noise(6 * n, seed=62), 1), (np.full(6 * n, -32768, np.int16), 0), (np.full(6 * n, 32767, np.int16), 1)):
        carry, st = _check(iq, 2 * n, downsample_passes=passes, comp_fir_size=9, offset_tuning=off)
        want, got = carry_tuple(carry_from_oracle_state(st)), carry_tuple(carry)
        assert got[8][:12 * passes] == want[8][:12 * passes] and got[9][:12 * passes] == want[9][:12 * passes]
    _check(sig_noise(8 * n, seed=63, amp=30000), 2 * n, n_runs=2, pipelined=True, downsample_passes=passes, comp_fir_size=0)


@pytest.mark.parametrize("fir", [9, 0])
@pytest.mark.parametrize("extra", [dict(), dict(offset_tuning=1), dict(deemph=0), dict(rate_out2=0), dict(dc_block_audio=1),
                                   dict(rate_out=250000, rate_out2=48000, deemph_a=7)])
def test_whole_chain_in_the_cascade_kernel(fir, extra):
    """three fifth_order passes, the droop FIR and the -A fast discriminator in ONE launch (k_fm_fifth_regn<.., 3, DD> + k_fm_fifth_tails +
    k_fm_dd_edges): oracle bits and carries for every audio tail behind it, one run and chained pipelined runs (histories, pre_r/pre_j and
    the FIR's nine samples crossing block and run seams), and the same carries as the separate kernels, which -A std takes (custom_atan=0)"""
    from gpu_support import gpu_fm_stream, carry_tuple, carry_from_oracle_state
    n = 16384
    params = dict(downsample_passes=3, comp_fir_size=fir, **extra)
    for iq in (sig_fm(5 * n, seed=81, amp=9000.0, noise=900), sig_noise(10 * n, seed=82), np.full(10 * n, -32768, np.int16), sig_alternating(10 * n)):
        carry, st = _check(iq, 2 * n, **params)
        want, got = carry_tuple(carry_from_oracle_state(st)), carry_tuple(carry)
        assert got[8][:36] == want[8][:36] and got[9][:36] == want[9][:36]
        if fir == 9:
            assert got[10] == want[10] and got[11] == want[11]
    iq = sig_fm(13 * n, seed=83, amp=12000.0, noise=2000)
    _check(iq, 2 * n, n_runs=4, pipelined=True, **params)
    _check(iq, 2 * n, n_runs=3, **params)
    # the separate kernels (k_fm_fifth_fused + k_fm_droop_disc) on the same capture: another discriminator, the same cascade and FIR histories
    fused = gpu_fm_stream(iq, 2 * n, n_runs=4, pipelined=True, **params)
    plain = gpu_fm_stream(iq, 2 * n, n_runs=4, pipelined=True, custom_atan=0, **params)
    assert carry_tuple(fused[2])[8:] == carry_tuple(plain[2])[8:]
    _check(iq, 2 * n, n_runs=4, pipelined=True, custom_atan=0, **params)


def test_whole_chain_kernel_block_shapes_and_libm_records(monkeypatch):
    """block lengths from one tile (2048 samples: 256 outputs, two waves; one tile per wave) to 2^18 (two tiles walked per wave), and every
    block's first sample flagged for the host (flag_all): the records k_fm_dd_edges writes are re-evaluated with libm and patched into
    the tiled pcm like k_fm_droop_disc's"""
    for n, nb in ((2048, 9), (4096, 5), (6144, 3), (262144, 2), (2048 * 59, 2), (2048 * 15, 3)):
        iq = sig_fm(nb * n, seed=90 + nb, amp=7000.0, noise=500)
        _check(iq, 2 * n, downsample_passes=3, comp_fir_size=9)
        _check(iq, 2 * n, n_runs=2, pipelined=True, downsample_passes=3, comp_fir_size=0, offset_tuning=1)
    monkeypatch.setenv("RXGPU_FLAG_ALL", "2")
    from gpu_support import gpu_fm_stream
    iq = sig_fm(12 * 8192, seed=95, amp=7000.0, noise=500)
    want, _, _ = oracle_fm_stream(iq, 16384, downsample_passes=3, comp_fir_size=9)
    got, _, _, fix = gpu_fm_stream(iq, 16384, n_runs=3, pipelined=True, downsample_passes=3, comp_fir_size=9)
    assert np.array_equal(got, want) and fix >= 11


def test_fifth_order_carry_across_runs():
    iq = sig_noise(12 * 16384, seed=21, amp=20000)
    _check(iq, 16384, n_runs=3, downsample_passes=3, comp_fir_size=9)


@pytest.mark.parametrize("ds,block_len,std", [(118, 16384, 0), (118, 2 * 131072, 0), (6, 4096 + 8, 0), (20, 2 * 5000, 0), (118, 16384, 1)])
def test_host_reevaluation_of_libm_samples(ds, block_len, std, monkeypatch):
    """RXGPU_FLAG_ALL reports every libm discriminator sample as undecided, so each one is re-evaluated by the host
    (polar_discriminant with glibc's atan2) from the stored lowpassed[] pair and the audio stages are redone:
    same output, and the fix-up path -- incl. the windows the seam kernel sums again when lowpassed[] is kept
    sparse -- is exercised"""
    from gpu_support import gpu_fm_stream
    monkeypatch.setenv("RXGPU_FLAG_ALL", "1")
    n_blocks = 4 if block_len > 100000 else 12
    iq = sig_fm(n_blocks * block_len // 2, seed=31)
    params = dict(downsample=ds, custom_atan=0 if std else 1)
    want, want_lens, st = oracle_fm_stream(iq, block_len, **params)
    got, got_lens, carry, fixups = gpu_fm_stream(iq, block_len, **params)
    assert np.array_equal(got, want)
    assert fixups >= n_blocks - 1                       # an exactly zero angle is never flagged


@pytest.mark.parametrize("params,block_len", [
    (dict(downsample=4, post_downsample=4), 16384),                   # -o 4
    (dict(downsample=8, post_downsample=2, deemph=0, rate_out2=-1), 16384),
    (dict(downsample=2, post_downsample=4, dc_block_audio=1), 8192),  # the reference's own -o geometry (ds=2, generic decimator)
    (dict(downsample_passes=3, post_downsample=4), 16384),
    (dict(downsample=6, mode=1, output_scale=2, post_downsample=2), 2 * 6 * 512),   # am + -o
    (dict(downsample=118, dc_block_raw=1), 16384),                    # -E rdc
    (dict(downsample=6, dc_block_raw=1, rdc_block_const=3), 2 * 6000),
    (dict(downsample_passes=3, comp_fir_size=9, dc_block_raw=1), 16384),
    (dict(downsample=10, dc_block_raw=1, offset_tuning=1, custom_atan=0), 8192),
    # blocks of whole 16384-sample spans in front of the span decimator: the averages are subtracted inside it (k_fm_decimate<.., RDC>), no corrected copy
    (dict(downsample=118, dc_block_raw=1), 2 * 16384),
    (dict(downsample=118, dc_block_raw=1, rdc_block_const=1), 2 * 32768),             # two spans per block, the fastest-moving average
    (dict(downsample=40, dc_block_raw=1, rdc_block_const=5, custom_atan=0), 2 * 16384),  # 4-byte slots, the separate discriminator (-A std)
    (dict(downsample=64, dc_block_raw=1, offset_tuning=1), 2 * 16384),                 # no rotation: one dc word for all four phases; wide slots
    (dict(downsample=33, dc_block_raw=1, mode=1, deemph=0, squelch_level=20), 2 * 16384),  # am + squelch behind it
])
def test_post_downsample_and_raw_dc_block(params, block_len):
    """-o (low_pass_simple, rtl_fm.c:373-387) and -E rdc (dc_block_raw_filter, rtl_fm.c:699-721), several runs with
    the dc averages carried from run to run"""
    for sig in ("fm", "noise_full", "dc"):
        iq = _signals(12 * block_len)[sig]
        carry, st = _check(iq, block_len, n_runs=3, **params)
        assert (carry.dc_avgI, carry.dc_avgQ) == (st.dc_avgI, st.dc_avgQ)


@pytest.mark.parametrize("ds", [1, 2, 3])
@pytest.mark.parametrize("extra", [dict(), dict(mode=4, deemph=0, rate_out2=-1), dict(mode=1, deemph=0), dict(offset_tuning=1, custom_atan=0), dict(dc_block_raw=1)])
def test_decimation_below_four(ds, extra):
    """ds = 1, 2, 3 (k_fm_decimate_tiny: twelve samples per thread, the windows that end in them, the samples in front of the thread's edge read again,
    the run's first window from the carried now_r/now_j): blocks of 4100 samples -- no multiple of 3 or 12, so the phase prev_index walks through
    every value from run to run and the last thread of a run is ragged -- fm / raw / am / unrotated / -E rdc chains, three runs, against the oracle"""
    block_len = 2 * 4100
    for sig in ("fm", "noise_full"):
        iq = _signals(12 * block_len)[sig]
        _check(iq, block_len, n_runs=3, downsample=ds, **extra)


def test_random_parameter_sweep():
    """seeded random parameter sets -- every switch of the chain at once -- HIP against the oracle, two runs each"""
    rng = np.random.default_rng(20260925)
    for case in range(48):
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
        iq = [sig_fm(4 * n, seed=case), sig_noise(4 * block, seed=case), sig_noise(4 * block, seed=case, amp=300)][case % 3]
        try:
            _check(iq, block, n_runs=2, pipelined=bool(case & 1), **params)
        except Exception as e:
            raise AssertionError("case %d block %d %r: %s" % (case, block, params, e))


def test_post_downsample_needs_whole_groups():
    from gpu_support import to_dev
    import torch
    s = R.FmStream(R.FmParams.wbfm(downsample=6, post_downsample=4), 2, 16384)      # 8192 % 6 != 0
    d = to_dev(np.zeros(2 * 16384, np.int16))
    o = torch.zeros(8192, dtype=torch.int16, device="cuda")
    with pytest.raises(R.RxGpuError, match="multiple"):
        s.run(d.data_ptr(), 2, 16384, o.data_ptr(), o.numel())
    s.close()


@pytest.mark.parametrize("topcap", ["1", "3"])
def test_deemph_multi_level_tree(topcap, monkeypatch):
    """forces the up/down levels of the de-emphasis tree scan (normally only for >4M audio samples)"""
    monkeypatch.setenv("RXGPU_DEEMPH_TOPCAP", topcap)
    for sig in ("fm", "zeros", "noise_small"):
        iq = _signals(40 * 16384)[sig]
        _check(iq, 16384, downsample=4)
        _check(iq, 16384, downsample=4, deemph_a=33)


@pytest.mark.parametrize("params", [
    dict(downsample=6, mode=1, output_scale=3, deemph=0),             # -M am
    dict(downsample=6, mode=2, output_scale=2),                       # -M usb (+deemph, resample)
    dict(downsample=6, mode=3, deemph=0, rate_out2=-1),               # -M lsb
    dict(downsample=6, mode=4),                                       # -M raw
    dict(downsample=118, mode=4),
    dict(downsample=10, custom_atan=2),                               # -A lut
    dict(downsample=10, custom_atan=3),                               # -A ale
    dict(downsample=6, squelch_level=40),                             # -l
    dict(downsample=6, squelch_level=2000),
    dict(downsample=118, squelch_level=900),
    dict(downsample=6, dc_block_audio=1),                             # -E adc
    dict(downsample=118, dc_block_audio=1, deemph=0),
    dict(downsample=118, dc_block_audio=1, deemph=0, rate_out2=-1),
    dict(downsample_passes=3, dc_block_audio=1, squelch_level=100, mode=1, output_scale=1),
    dict(downsample_passes=3, comp_fir_size=9, mode=4),
])
@pytest.mark.parametrize("n_runs", [1, 3])
def test_other_demodulators_and_filters(params, n_runs):
    """-M am|usb|lsb|raw, -A lut|ale, -l squelch, -E adc: SURVEY section 8(f) rank 3, same parity bar"""
    from gpu_support import gpu_fm_stream
    import rx_tools_amd as R
    for iq in (sig_fm(9 * 8192, seed=31), sig_noise(9 * 16384, seed=32), sig_noise(9 * 16384, seed=33, amp=300),
               np.zeros(9 * 16384, np.int16)):
        want, want_lens, st = oracle_fm_stream(iq, 16384, **params)
        c0 = R.FmCarry()
        c0.squelch_hits = 11                                            # demod_init, rtl_fm.c:1091
        got, got_lens, carry, _ = gpu_fm_stream(iq, 16384, n_runs=n_runs, carry=c0, **params)
        assert len(got) == len(want) and np.array_equal(got, want)
        assert np.array_equal(got_lens, want_lens)
        assert (carry.squelch_hits, carry.dc_avg, carry.now_lpr, carry.prev_lpr_index, carry.pre_r, carry.pre_j) == \
            (st.squelch_hits, st.dc_avg, st.now_lpr, st.prev_lpr_index, st.pre_r, st.pre_j)


def test_scale_formula_exhaustive_on_device():
    """F0 on the device for all 65536 int16 values == the reference's fp64 expression"""
    from gpu_support import gpu_fm_stream
    x = np.arange(-32768, 32768, dtype=np.int16)
    iq = np.zeros(4 * 65536, np.int16)
    iq[0::4] = x          # I of even samples; ds=1, no rotation: the decimated I is the scaled value
    iq[3::4] = x[::-1]    # Q of odd samples
    O = oracle()
    want = np.array([O.rxo_scale_sample(int(v)) for v in x], dtype=np.int16)
    # ds=1 + offset tuning + raw output of the discriminator is awkward to read back; use
    # the boxcar with ds=4 on a stream where only one of four samples is non-zero instead
    iq2 = np.zeros(8 * 65536, np.int16)
    iq2[0::8] = x
    iq2[1::8] = x[::-1]
    from support import oracle_fm_stream
    got, _, c, _ = gpu_fm_stream(iq2, 8 * 65536, downsample=4, offset_tuning=1, deemph=0, rate_out2=-1)
    ref, _, _ = oracle_fm_stream(iq2, 8 * 65536, downsample=4, offset_tuning=1, deemph=0, rate_out2=-1)
    assert np.array_equal(got, ref)
    assert want[0] == -127 and want[-1] == 128


def test_full_size_block_and_properties():
    """2^22 samples (32 blocks of 131072) at ds=118: bit-exact vs oracle, plus the
    size-independent properties used at bench size: output count and carry closed forms"""
    n_blocks, bl = 32, 2 * 131072
    iq = np.tile(sig_fm(4 * 131072, seed=12345), n_blocks // 4)
    carry, st = _check(iq, bl, downsample=118)
    T = n_blocks * 131072
    assert carry.prev_index == T % 118
    assert carry.prev_lpr_index == ((T // 118) * 32000) % 170000


def test_config1_ragged_last_block():
    """BASELINE config 1 exactly: 1.2 M samples = 9 callback blocks of 131072 + one of 20352 (what a 1 s capture
    at 1.2 Msps delivers), -M wbfm -s 240000 -> 32000 int16; two runs because the last block is shorter"""
    import ctypes as C
    import rx_tools_amd as R
    from gpu_support import to_dev, torch_cuda
    from support import oracle, oracle_fm_state, ptr16
    torch = torch_cuda()
    iq = sig_fm(1200000, seed=5)
    kw = dict(downsample=5, rate_out=240000, deemph_a=19)
    O, st = oracle(), oracle_fm_state(**kw)
    lp, res, want = np.zeros(262144, np.int16), np.zeros(131072, np.int16), []
    pos = 0
    for n in [131072] * 9 + [20352]:
        blk = np.ascontiguousarray(iq[2 * pos:2 * (pos + n)])
        k = O.rxo_fm_block(C.byref(st), ptr16(blk), 2 * n, ptr16(lp), None, ptr16(res))
        want.append(res[:k].copy())
        pos += n
    want = np.concatenate(want)
    assert len(want) == 32000
    p = R.FmParams.wbfm(**kw)
    s = R.FmStream(p, 9, 2 * 131072)
    d_iq = to_dev(iq)
    d_out = torch.zeros(40000, dtype=torch.int16, device="cuda")
    n1, _ = s.run(d_iq.data_ptr(), 9, 2 * 131072, d_out.data_ptr(), 40000)
    n2, _ = s.run(d_iq.data_ptr() + 9 * 131072 * 4, 1, 2 * 20352, d_out.data_ptr() + 2 * n1, 40000 - n1)
    got = d_out[:n1 + n2].cpu().numpy()
    assert n1 + n2 == 32000 and np.array_equal(got, want)
    c = s.get_carry()
    assert (c.prev_index, c.now_r, c.now_j, c.now_lpr, c.prev_lpr_index) == (st.prev_index, st.now_r, st.now_j, st.now_lpr, st.prev_lpr_index)
    s.close()


SOAK_CHAINS = [
    dict(downsample_passes=3, comp_fir_size=9),                    # the whole chain inside the cascade kernel
    dict(downsample_passes=3),
    dict(downsample_passes=4),
    dict(downsample_passes=7, comp_fir_size=9),                    # three fused groups
    dict(downsample=5, rate_out=240000, deemph_a=19),              # BASELINE configs[0]: the small-decimation kernel, 256-sample chunks
    dict(downsample=6),                                            # -M wbfm default
    dict(downsample=118),                                          # the headline's prefix-scan decimator
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", SOAK_CHAINS)
def test_soak_of_320_pipelined_runs_of_random_shapes(kw, seed, monkeypatch):
    """ONE stream object, 320 rxgpu_fm_stream_run_async calls without a wait in between, every run of another shape drawn at random --
    whole tiles, ragged block lengths (the literal per-block kernels on the -F chains), blocks barely longer than the decimation, one to
    four blocks per run -- with a pseudo-random eighth of the libm samples handed to the host WRONG ($RXGPU_FLAG_ALL=3), so that runs
    with and without fix-ups, and fix-ups that re-run the audio stages of the run behind them, interleave with the four-stream
    pipeline's buffer hand-overs.  Output of the whole sequence and every carry against the oracle; three shuffles per chain."""
    import ctypes as C
    from gpu_support import to_dev, torch_cuda, carry_tuple, carry_from_oracle_state
    from support import oracle, oracle_fm_state, ptr16
    torch = torch_cuda()
    monkeypatch.setenv("RXGPU_FLAG_ALL", "3")
    rnd = np.random.RandomState(1000 * seed + 7 * kw.get("downsample", 0) + kw.get("downsample_passes", 0) + kw.get("comp_fir_size", 0))
    passes = kw.get("downsample_passes", 0)
    least = (2 << passes) if passes else kw["downsample"]          # shortest block the stream API takes (shorter ones: drop-in only)
    whole = [2048, 4096, 8192, 16384, 6144]
    ragged = [1009, 777, 4100, 2050, 12290, 3 * 2048 + 2]
    short = [least + 2, least + 9, 2 * least + 1, 300]
    shapes = []
    for _ in range(320):
        kind = rnd.choice(3, p=[0.5, 0.3, 0.2])
        n = int(rnd.choice([whole, ragged, short][kind]))
        if n % 2 and passes and ((2 * n) >> passes) < 2:
            n = 2 * least + 2
        shapes.append((int(rnd.randint(1, 5)), max(n, least)))
    total = sum(nb * n for nb, n in shapes)
    iq = sig_fm(total, seed=500 + seed, amp=9000.0, noise=1200)
    O, st = oracle(), oracle_fm_state(**kw)
    lp, res, want = np.zeros(262144, np.int16), np.zeros(131072, np.int16), []
    pos = 0
    for nb, n in shapes:
        for _ in range(nb):
            blk = np.ascontiguousarray(iq[2 * pos:2 * (pos + n)])
            k = O.rxo_fm_block(C.byref(st), ptr16(blk), 2 * n, ptr16(lp), None, ptr16(res))
            want.append(res[:k].copy())
            pos += n
    want = np.concatenate(want)
    s = R.FmStream(R.FmParams.wbfm(**kw), 4, 2 * 16384)
    d_iq = to_dev(iq)
    d_out = torch.zeros(len(want) + 4096, dtype=torch.int16, device="cuda")
    pos = got_n = 0
    for nb, n in shapes:
        k, _ = s.run_async(d_iq.data_ptr() + 4 * pos, nb, 2 * n, d_out.data_ptr() + 2 * got_n, d_out.numel() - got_n)
        got_n += k
        pos += nb * n
    s.wait()
    got = d_out[:got_n].cpu().numpy()
    assert got_n == len(want)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at output %d of %d: got %d want %d (%d bad)" % (bad[0], len(want), got[bad[0]], want[bad[0]], bad.size)
    assert s.host_fixups > 20                                      # the forced flags did go through the host
    cg, cw = carry_tuple(s.get_carry()), carry_tuple(carry_from_oracle_state(st))
    assert cg[:8] == cw[:8]
    if passes:
        assert cg[8][:12 * passes] == cw[8][:12 * passes] and cg[9][:12 * passes] == cw[9][:12 * passes]
    if kw.get("comp_fir_size") == 9:
        assert cg[10] == cw[10] and cg[11] == cw[11]
    s.close()


@pytest.mark.parametrize("kw", [dict(downsample_passes=3, comp_fir_size=9), dict(downsample_passes=4), dict(downsample_passes=7, comp_fir_size=9),
                                dict(downsample_passes=3, comp_fir_size=9, custom_atan=0)])
def test_pipelined_runs_of_different_shapes(kw):
    """one pipelined sequence whose runs change shape -- whole tiles, ragged blocks (the literal per-block path), a block length with a
    shorter first group -- so that the cascade and droop histories change hands between the seam stream and stream B from run to run;
    block by block against the oracle, carries included"""
    import ctypes as C
    from gpu_support import to_dev, torch_cuda, carry_tuple, carry_from_oracle_state
    from support import oracle, oracle_fm_state, ptr16
    torch = torch_cuda()
    shapes = [(4, 4096), (3, 1009), (2, 16384), (1, 4100), (3, 4096), (2, 2048), (2, 777), (5, 8192)]    # (blocks, complex samples per block)
    total = sum(nb * n for nb, n in shapes)
    iq = sig_fm(total, seed=123, amp=9000.0, noise=1200)
    O, st = oracle(), oracle_fm_state(**kw)
    lp, res, want = np.zeros(262144, np.int16), np.zeros(131072, np.int16), []
    pos = 0
    for nb, n in shapes:
        for _ in range(nb):
            blk = np.ascontiguousarray(iq[2 * pos:2 * (pos + n)])
            k = O.rxo_fm_block(C.byref(st), ptr16(blk), 2 * n, ptr16(lp), None, ptr16(res))
            want.append(res[:k].copy())
            pos += n
    want = np.concatenate(want)
    for rep in range(3):                                     # the hand-over is a matter of timing: a few times
        s = R.FmStream(R.FmParams.wbfm(**kw), 8, 2 * 16384)
        d_iq = to_dev(iq)
        d_out = torch.zeros(len(want) + 4096, dtype=torch.int16, device="cuda")
        pos = got_n = 0
        for nb, n in shapes:
            k, _ = s.run_async(d_iq.data_ptr() + 4 * pos, nb, 2 * n, d_out.data_ptr() + 2 * got_n, d_out.numel() - got_n)
            got_n += k
            pos += nb * n
        s.wait()
        got = d_out[:got_n].cpu().numpy()
        assert got_n == len(want) and np.array_equal(got, want)
        cg, cw = carry_tuple(s.get_carry()), carry_tuple(carry_from_oracle_state(st))
        np_ = 12 * kw["downsample_passes"]
        assert cg[:8] == cw[:8] and cg[8][:np_] == cw[8][:np_] and cg[9][:np_] == cw[9][:np_]
        if kw.get("comp_fir_size") == 9:
            assert cg[10] == cw[10] and cg[11] == cw[11]
        s.close()


def test_error_codes():
    """bad geometry fails loudly with the documented codes, never silently"""
    import rx_tools_amd as R
    from gpu_support import torch_cuda
    torch = torch_cuda()
    L = R.lib()
    d_iq = torch.zeros(4 * 16384, dtype=torch.int16, device="cuda")
    d_out = torch.zeros(16384, dtype=torch.int16, device="cuda")
    s = R.FmStream(R.FmParams.wbfm(downsample=6), 4, 16384)
    for args, code in [((d_iq.data_ptr(), 0, 16384, d_out.data_ptr(), 16384), -2),        # no blocks: EINVAL
                       ((d_iq.data_ptr(), 8, 16384, d_out.data_ptr(), 16384), -5),        # more than the stream holds: ECAPACITY
                       ((d_iq.data_ptr(), 4, 16384, d_out.data_ptr(), 10), -5),           # output too small: ECAPACITY
                       ((d_iq.data_ptr(), 4, 8, d_out.data_ptr(), 16384), -3),           # blocks shorter than the boxcar: EUNSUPPORTED (drop-in only)
                       ((d_iq.data_ptr(), 4, 0, d_out.data_ptr(), 16384), -2),           # empty blocks: EINVAL (drop-in only)
                       ((d_iq.data_ptr(), 4, 7, d_out.data_ptr(), 16384), -2)]:          # an odd int16 count: EINVAL
        n = __import__("ctypes").c_size_t(0)
        assert L.rxgpu_fm_stream_run(s._h, *args, __import__("ctypes").byref(n), None) == code, L.rxgpu_last_error()
    s.close()
    s = R.FmStream(R.FmParams.wbfm(downsample_passes=5), 4, 16384)
    n = __import__("ctypes").c_size_t(0)
    # a -F block that is not a multiple of 2^passes is taken (the literal per-block path) ...
    assert L.rxgpu_fm_stream_run(s._h, d_iq.data_ptr(), 4, 2 * 1000, d_out.data_ptr(), 16384, __import__("ctypes").byref(n), None) == 0
    # ... unless it leaves fewer than two int16 after the cascade (the reference then reads in front of lowpassed[])
    assert L.rxgpu_fm_stream_run(s._h, d_iq.data_ptr(), 4, 2 * 20, d_out.data_ptr(), 16384, __import__("ctypes").byref(n), None) == -3
    s.close()
    with pytest.raises(R.RxGpuError):
        R.FmStream(R.FmParams.wbfm(custom_atan=7), 4, 16384)
    with pytest.raises(R.RxGpuError):
        R.PowerScan(R.PowerParams(22, 1 << 24, 1, 0, 1, 0, 0), 1, np.ones(1 << 22, np.int32), np.zeros(3 << 20, np.int16))   # beyond the reference's 2^21


# ----------------------------------------------------------------------------- round 2: libm fix-ups inside the pipeline

@pytest.mark.parametrize("params,block_len,n_runs", [
    (dict(downsample=118), 16384, 5),
    (dict(downsample=6), 8192, 6),
    (dict(downsample=20, custom_atan=0), 2 * 5000, 4),               # -A std: every sample is a libm sample
    (dict(downsample_passes=3, comp_fir_size=9), 16384, 4),
    (dict(downsample=10, post_downsample=2, dc_block_audio=1), 2 * 10 * 400, 5),
    (dict(downsample=7, mode=0, deemph=0, rate_out2=-1), 2 * 7 * 300 + 2, 3),   # generic decimator (n % 4 != 0), no audio stages
])
def test_pipelined_fixups_redo_audio_stages(params, block_len, n_runs, monkeypatch):
    """RXGPU_FLAG_ALL=2: the device flags every libm sample AND stores a wrong value for it, in a pipelined sequence
    (two runs in flight).  Only the host's re-evaluation, the patch of pcm[] and the redo of the audio stages of the
    flagged run and of the run behind it -- from the snapshot of their carries-in -- can give the oracle's output and
    carries.  Nothing is rolled back."""
    monkeypatch.setenv("RXGPU_FLAG_ALL", "2")
    from gpu_support import gpu_fm_stream, carry_tuple, carry_from_oracle_state
    n_blocks = 4 * n_runs
    iq = sig_fm(n_blocks * block_len // 2, seed=77)
    want, want_lens, st = oracle_fm_stream(iq, block_len, **params)
    got, got_lens, carry, fixups = gpu_fm_stream(iq, block_len, n_runs=n_runs, pipelined=True, **params)
    assert np.array_equal(got, want)
    assert np.array_equal(got_lens, want_lens)
    assert carry_tuple(carry)[:8] == carry_tuple(carry_from_oracle_state(st))[:8]
    assert fixups >= n_blocks - 1


def test_libm_flag_rate_on_a_million_distinct_blocks():
    """2^20 callback blocks of 16 noise samples (every block's first decimated sample is a libm sample, all of them
    different) through a pipelined sequence: bit-exact, and the number of samples the device could not decide stays
    at the 2 * 2^-33 per sample the window predicts (expected 2.4e-4 here; 1e-6 -- the old window -- would be ~2)."""
    from gpu_support import gpu_fm_stream
    n_blocks, block_len = 1 << 20, 32
    iq = sig_noise(n_blocks * block_len, seed=2026, amp=20000)
    params = dict(downsample=4)
    want, want_lens, st = oracle_fm_stream(iq, block_len, **params)
    got, got_lens, carry, fixups = gpu_fm_stream(iq, block_len, n_runs=4, pipelined=True, **params)
    assert np.array_equal(got, want)
    assert fixups <= 1
    # -A std: 2^22 libm samples in one go
    iq = sig_noise(2 * (1 << 22) * 4, seed=2027, amp=3000)
    params = dict(downsample=4, custom_atan=0, deemph=0, rate_out2=-1)
    want, _, _ = oracle_fm_stream(iq, 2 * 65536, **params)
    got, _, _, fixups = gpu_fm_stream(iq, 2 * 65536, n_runs=2, pipelined=True, **params)
    assert np.array_equal(got, want)
    assert fixups <= 1


@pytest.mark.parametrize("params,block_len,n_blocks", [
    (dict(downsample=118), 16384, 40),
    (dict(downsample=6), 8192, 37),
    (dict(downsample_passes=3, comp_fir_size=9), 16384, 19),
    (dict(downsample=5, mode=4, deemph=0, rate_out2=-1), 8192, 21),      # raw_demod: 2 int16 per decimated sample
    (dict(downsample=6, squelch_level=40), 8192, 9),                     # squelch: one chunk
])
def test_run_host_chunked(params, block_len, n_blocks, monkeypatch):
    """rxgpu_fm_stream_run_host with a capture of many chunks (RXGPU_HOST_CHUNK shrinks them): H2D of chunk c+1 on the
    copy stream while chunk c is demodulated, three staging buffers in rotation, outputs concatenated == the oracle."""
    monkeypatch.setenv("RXGPU_HOST_CHUNK", str(4 * block_len * 2))
    from gpu_support import carry_tuple, carry_from_oracle_state
    iq = sig_fm(n_blocks * block_len // 2, seed=99)
    want, want_lens, st = oracle_fm_stream(iq, block_len, **params)
    p = R.FmParams.wbfm()
    for k, v in params.items():
        setattr(p, k, v)
    s = R.FmStream(p, n_blocks, block_len)
    if "squelch_level" in params:
        s.set_carry(R.FmCarry(squelch_hits=11))
    out = np.zeros(len(iq) + 64, np.int16)
    R.check(R.lib().rxgpu_pin(iq.ctypes.data, iq.nbytes))
    try:
        n, lens = s.run_host(iq.ctypes.data, n_blocks, block_len, out.ctypes.data, out.size, True)
    finally:
        R.check(R.lib().rxgpu_unpin(iq.ctypes.data))
    assert n == len(want) and np.array_equal(out[:n], want)
    assert np.array_equal(np.array(lens, np.int32), want_lens)
    assert carry_tuple(s.get_carry())[:8] == carry_tuple(carry_from_oracle_state(st))[:8]
    s.close()


def test_raw_mode_full_capacity_block_through_run_host():
    """-M raw hands lowpassed[] through: 2 int16 per decimated sample.  A block that fills the stream's capacity
    (what dongle_thread_fn delivers: 131072 complex samples) must fit the host-fed path's staging."""
    block_len = 2 * 131072
    iq = sig_fm(block_len // 2, seed=3)
    params = dict(downsample=1, mode=4, deemph=0, rate_out2=-1)
    want, _, _ = oracle_fm_stream(iq, block_len, **params)
    p = R.FmParams.wbfm()
    for k, v in params.items():
        setattr(p, k, v)
    s = R.FmStream(p, 1, block_len)
    out = np.zeros(block_len + 64, np.int16)
    n, _ = s.run_host(iq.ctypes.data, 1, block_len, out.ctypes.data, out.size)
    assert n == len(want) == block_len and np.array_equal(out[:n], want)
    s.close()


@pytest.mark.parametrize("block_len", [2 * 6001, 2 * 4098, 2 * 777])
def test_raw_dc_block_any_block_length(block_len):
    """-E rdc no longer needs blocks of a multiple of 4 samples (readStream may return any count)"""
    iq = sig_fm(7 * block_len // 2, seed=8)
    _check(iq, block_len, downsample=6, dc_block_raw=1, rdc_block_const=4)


# ----------------------------------------------------------------------------- round 2: the tiled audio path

@pytest.mark.parametrize("params,block_len,n_runs", [
    (dict(downsample=6), 8192, 1),
    (dict(downsample=6), 8192, 3),
    (dict(downsample=5, rate_out=240000, deemph_a=19), 2 * 20000, 2),            # GS=64, chunks of 256 (BASELINE configs[0])
    (dict(downsample=4, deemph_a=9), 2 * 4096, 2),                                # -c eu
    (dict(downsample=12, rate_out=96000, rate_out2=48000, deemph_a=7), 2 * 12 * 700, 2),   # ratio 2
    (dict(downsample=4, rate_out=1000000, rate_out2=32000), 2 * 65536, 2),        # ratio 31, windows of 31/32 samples
    (dict(downsample=118), 2 * 131072, 2),
    (dict(downsample=7, deemph_a=63), 2 * 7 * 1234 + 8, 3),                       # odd block geometry, a at the top of the GS=64 range
])
@pytest.mark.parametrize("tiled", [True, False])
def test_tiled_audio_path_and_staged_path(params, block_len, n_runs, tiled):
    """the lane-per-chunk kernels on the tiled stream (k_fm_deemph_scan_t / up0 / down0 / apply_rs_t: de-emphasis with the
    resampler inline) and the LDS-staged kernels -- what the same chain takes when `a` is even (here: a + 1) -- both reproduce the
    oracle, sample for sample and carry for carry, also pipelined across runs whose lengths are not multiples of a chunk"""
    if not tiled:
        a = params.get("deemph_a", 13)                            # FmParams.wbfm(): 13 (75 us at 170 kHz)
        params = dict(params, deemph_a=(a + 1) if a % 2 else a)
    n_blocks = 4 * n_runs + 1
    iq = sig_fm(n_blocks * block_len // 2, seed=55, amp=9000.0, noise=900)
    _check(iq, block_len, n_runs=n_runs, pipelined=n_runs > 1, **params)


@pytest.mark.parametrize("sig", ["noise_full", "zeros", "dc", "alternating"])
def test_tiled_audio_path_hostile_signals(sig):
    """constant input (no chunk ever merges its candidates: the tree has to carry the exact state), full-scale noise (every
    int16 wrap of the discriminator), and the multi-level tree forced by a long run"""
    iq = _signals(40 * 8192)[sig]
    _check(iq, 8192, n_runs=2, pipelined=True, downsample=4)


@pytest.mark.parametrize("params,block_len,n_runs", [
    (dict(downsample=6), 8192, 3),
    (dict(downsample=4, deemph_a=9), 2 * 4096, 2),
    (dict(downsample=12, rate_out=96000, rate_out2=48000, deemph_a=15), 2 * 12 * 700, 2),
    (dict(downsample=118), 2 * 131072, 2),
])
def test_tiled_audio_path_long_run_chunks(params, block_len, n_runs, monkeypatch):
    """runs of 2^25 and more demodulated samples take 256-sample chunks even where the warm-up would fit 128
    (the scan's warm-up per chunk weighs half as much); $RXGPU_DEEMPH_CHUNK=256 forces that choice on a short run"""
    monkeypatch.setenv("RXGPU_DEEMPH_CHUNK", "256")
    n_blocks = 4 * n_runs + 1
    iq = sig_fm(n_blocks * block_len // 2, seed=56, amp=9000.0, noise=900)
    _check(iq, block_len, n_runs=n_runs, pipelined=n_runs > 1, **params)
    _check(_signals(40 * 8192)["dc"], 8192, n_runs=2, pipelined=True, downsample=4)


@pytest.mark.parametrize("params,block_len,n_runs", [
    (dict(downsample=6), 8192, 3),
    (dict(downsample=4, deemph_a=9), 2 * 4096 + 8, 2),
    (dict(downsample=5, deemph_a=15), 2 * 5000, 2),
    (dict(downsample=118), 2 * 131072, 2),
])
@pytest.mark.parametrize("scan", ["0", "1"])
def test_tiled_audio_path_both_scans(params, block_len, n_runs, scan, monkeypatch):
    """the two scans of 128-sample chunks: k_fm_deemph_scan_r (chunk in registers, warm-up from the neighbouring lane, 63
    chunks per wave -- what runs of 2^25 and more demodulated samples take) and k_fm_deemph_scan_t; $RXGPU_SCAN_T picks one"""
    monkeypatch.setenv("RXGPU_SCAN_T", scan)
    n_blocks = 4 * n_runs + 1
    iq = sig_fm(n_blocks * block_len // 2, seed=57, amp=9000.0, noise=900)
    _check(iq, block_len, n_runs=n_runs, pipelined=n_runs > 1, **params)
    _check(_signals(40 * 8192)["dc"], 8192, n_runs=2, pipelined=True, downsample=4)


def test_tiled_audio_path_multi_level_tree(monkeypatch):
    monkeypatch.setenv("RXGPU_DEEMPH_TOPCAP", "3")
    iq = sig_fm(64 * 16384 // 2, seed=66)
    _check(iq, 16384, n_runs=2, pipelined=True, downsample=4)


@pytest.mark.parametrize("params", [
    dict(downsample_passes=1), dict(downsample_passes=3), dict(downsample_passes=3, comp_fir_size=9), dict(downsample_passes=7),
    dict(downsample_passes=4, mode=1, output_scale=2), dict(downsample_passes=2, mode=4), dict(downsample_passes=5, mode=2, deemph=0),
    dict(downsample_passes=3, squelch_level=40), dict(downsample_passes=2, comp_fir_size=9, squelch_level=3000, dc_block_audio=1),
    dict(downsample_passes=3, custom_atan=0), dict(downsample_passes=4, custom_atan=2, offset_tuning=1), dict(downsample_passes=3, dc_block_raw=1),
])
@pytest.mark.parametrize("n", [97, 1009, 4099, 65537, 8190, 2048 + 4, 12])
def test_fifth_order_on_ragged_blocks(params, n):
    """-F on callback blocks whose sample count is not a multiple of 2^passes (readStream may return any count, rtl_fm.c:894-899):
    the literal per-block kernels follow every odd `lp_len >> i` of rtl_fm.c:764-769 -- different I and Q output counts, an odd
    final lp_len, pre_r/pre_j from lp[lp_len-2], lp[lp_len-1] -- through several blocks and two chained runs"""
    p = params["downsample_passes"]
    if ((2 * n) >> p) < 2:
        pytest.skip("the reference reads in front of lowpassed[] here: drop-in only (test_gpu_dropin.py)")
    block_len = 2 * n
    iq = sig_fm(5 * n, seed=1234 + n) if n % 2 else sig_noise(5 * block_len, seed=n, amp=9000)
    # squelch_hits starts at demod_init's 11 (rtl_fm.c:1091) on both sides
    carry, st = _check(iq, block_len, n_runs=2, carry_in=R.FmCarry(squelch_hits=11) if params.get("squelch_level") else None, **params)
    assert bytes(carry.lp_i_hist) == bytes(st.lp_i_hist) and bytes(carry.lp_q_hist) == bytes(st.lp_q_hist)
    assert bytes(carry.droop_i_hist) == bytes(st.droop_i_hist) and bytes(carry.droop_q_hist) == bytes(st.droop_q_hist)
    assert carry.dc_avg == st.dc_avg and (not params.get("squelch_level") or carry.squelch_hits == st.squelch_hits)


def test_stream_refuses_blocks_only_the_dropin_can_reproduce():
    """a block that leaves fewer than two int16 after the cascade, or low_pass blocks shorter than the decimation: the reference then
    reads pre_r/pre_j from in front of lowpassed[] -- struct memory the batched stream does not have"""
    from gpu_support import torch_cuda
    torch = torch_cuda()
    d_iq = torch.zeros(4096, dtype=torch.int16, device="cuda")
    d_out = torch.zeros(4096, dtype=torch.int16, device="cuda")
    for kw, block_len, n_blocks in ((dict(downsample_passes=7), 2 * 100, 2), (dict(downsample=118), 2 * 50, 4)):
        s = R.FmStream(R.FmParams.wbfm(**kw), n_blocks, block_len)
        with pytest.raises(R.RxGpuError, match="lowpassed"):
            s.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), d_out.numel())
        s.close()
