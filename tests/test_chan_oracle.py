"""CPU: the channeliser specification (oracle/rx_oracle.c rxo_chan_block -- fix_fft per window + fm_demod per
channel, both already pinned against the reference) behaves like a filter bank."""
import ctypes as C

import numpy as np

import rx_tools_amd as R
from support import oracle, ptr16, i16p, intp


class ChanCfg(C.Structure):
    _fields_ = [("bin_e", C.c_int), ("first_bin", C.c_int), ("n_channels", C.c_int), ("custom_atan", C.c_int), ("sinewave", i16p)]


def run(iq, bin_e, first_bin, n_channels, custom_atan, pre=None):
    O = oracle()
    O.rxo_chan_block.argtypes = [C.POINTER(ChanCfg), i16p, C.c_int, intp, i16p, C.c_size_t]
    sw = R.sine_table(bin_e)
    cfg = ChanCfg(bin_e, first_bin, n_channels, custom_atan, ptr16(sw))
    windows = len(iq) // 2 >> bin_e
    out = np.zeros((n_channels, windows), np.int16)
    pre = np.zeros(2 * n_channels, np.int32) if pre is None else pre
    O.rxo_chan_block(C.byref(cfg), ptr16(iq), len(iq), pre.ctypes.data_as(intp), ptr16(out), windows)
    return out, pre


def tone(n, k, windows, amp, dphi=0.0):
    """carrier at bin k of an n-point bank, plus a phase step of dphi radians per window"""
    t = np.arange(windows * n)
    ph = 2 * np.pi * k / n * t + dphi * (t // n)
    iq = np.empty(2 * len(t), np.int16)
    iq[0::2] = np.rint(amp * np.cos(ph))
    iq[1::2] = np.rint(amp * np.sin(ph))
    return iq


def test_carrier_lands_in_its_channel_only():
    out, pre = run(tone(256, 77, 40, 500), 8, 0, 256, 0)
    assert np.all(np.abs(out[77, 1:]) <= 40)                     # constant phasor -> zero frequency
    assert np.count_nonzero(out[(77 + 128) % 256]) == 0          # nothing half a band away
    assert pre[2 * 77] != 0 or pre[2 * 77 + 1] != 0              # the channel carries its last sample


def test_phase_steps_become_discriminator_output():
    """a phase advance of dphi per window is what fm_demod measures: dphi / pi * 2^14"""
    dphi = 0.5
    out, _ = run(tone(256, 30, 40, 500, dphi), 8, 30, 1, 0)
    want = dphi / 3.14159 * 16384
    assert np.all(np.abs(out[0, 2:] - want) < 80)


def test_block_structure_and_carry():
    """two calls with the carry == one stream, except that each call's first window is a libm sample"""
    iq = (np.random.RandomState(5).randint(-3000, 3000, size=2 * 256 * 16)).astype(np.int16)
    a, pre_a = run(iq[: 2 * 256 * 8].copy(), 8, 5, 50, 0)
    b, pre_b = run(iq[2 * 256 * 8:].copy(), 8, 5, 50, 0, pre_a)
    whole, pre_w = run(iq.copy(), 8, 5, 50, 0)
    assert np.array_equal(np.concatenate([a, b], axis=1), whole)   # -A std: every sample is the libm one
    assert np.array_equal(pre_b, pre_w)


# ----------------------------------------------------------------------------- pinned against the reference itself

import pytest                                                      # noqa: E402
from support import have_ref, oracle_chan_stream, ref_chan_stream, oracle_chan_nco_stream, ref_chan_nco_stream, sig_fm, sig_noise    # noqa: E402

skip_without_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")

# the geometries, discriminator modes and audio configurations tests/test_gpu_chan.py holds the device to (block counts kept, the
# largest blocks shortened where the case only needs "several windows per block": the reference legs run on one core here)
GEOMETRIES = [(10, 384, 256, 2 * 131072, 3), (10, 900, 256, 2 * 8192, 5), (8, 0, 256, 2 * 4096, 4), (12, 100, 7, 2 * 8192, 6), (5, 3, 20, 2 * 1024, 3),
              (9, 17, 100, 2 * 16384, 5), (11, 2000, 96, 2 * 131072, 2), (10, 0, 1024, 2 * 16384, 3)]
AUDIO = [(10, 256, 2 * 131072, 3, 1, 2, 19531, 8000, 1), (8, 64, 2 * 65536, 4, 1, 13, 170000, 32000, 1), (8, 64, 2 * 65536, 4, 1, 13, 170000, -1, 0),
         (9, 100, 2 * 32768, 5, 0, 0, 48000, 8000, 1), (8, 32, 2 * 65536, 3, 1, 64, 24000, 12000, 1), (8, 32, 2 * 16384, 3, 1, 200, 24000, 6000, 1),
         (8, 16, 2 * 1024, 6, 1, 9, 24000, 8000, 1)]


@pytest.mark.ref
@skip_without_ref
@pytest.mark.parametrize("bin_e,first_bin,n_channels,block_len,n_blocks", GEOMETRIES)
@pytest.mark.parametrize("custom_atan", [1, 0])
def test_chan_oracle_equals_reference_built_chain(bin_e, first_bin, n_channels, block_len, n_blocks, custom_atan):
    """rxo_chan_block == [the reference's fix_fft per window] -> [the reference's full_demod per channel and callback block]: every output
    sample and every channel's pre_r/pre_j, on the FM signal and on full-scale noise (int32 wraps of fast_atan2, int16 wraps of the butterflies)"""
    for iq in (sig_fm(n_blocks * block_len // 2, seed=70, amp=9000), sig_noise(n_blocks * block_len, seed=71)):
        want, want_pre, _ = ref_chan_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        got, pre, _ = oracle_chan_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        assert got.shape == want.shape
        bad = np.argwhere(got != want)
        assert bad.size == 0, "first mismatch at %s: port %d reference %d (%d bad)" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])], len(bad))
        assert np.array_equal(pre, want_pre)


@pytest.mark.ref
@skip_without_ref
@pytest.mark.parametrize("bin_e,n_channels,block_len,n_blocks,deemph,a,rate_out,rate_out2,custom_atan", AUDIO)
def test_chan_audio_oracle_equals_reference_built_chain(bin_e, n_channels, block_len, n_blocks, deemph, a, rate_out, rate_out2, custom_atan):
    """... and with the per-channel audio stages: the reference's full_demod runs deemph_filter (its one static accumulator forced to the
    channel's carried value in front of every call and read back after it) and low_pass_real on its own demod_state; output and the
    channel's (avg, now_lpr, prev_lpr_index) after every block, including a second run that starts from carried state"""
    iq = sig_noise(n_blocks * block_len, seed=4 + bin_e, amp=2500)
    half = (n_blocks + 1) // 2 * block_len
    kw = dict(deemph=deemph, a=a, rate_out=rate_out, rate_out2=rate_out2)
    want, want_pre, want_state = ref_chan_stream(iq, block_len, bin_e, 5, n_channels, custom_atan, **kw)
    got, pre, state = oracle_chan_stream(iq, block_len, bin_e, 5, n_channels, custom_atan, **kw)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(pre, want_pre) and np.array_equal(state, want_state)
    # two runs with the carries handed over == one
    a1, p1, s1 = ref_chan_stream(iq[:half], block_len, bin_e, 5, n_channels, custom_atan, **kw)
    a2, p2, s2 = ref_chan_stream(iq[half:], block_len, bin_e, 5, n_channels, custom_atan, pre=p1, audio=s1, **kw)
    assert np.array_equal(np.concatenate([a1, a2], axis=1), want) and np.array_equal(p2, want_pre) and np.array_equal(s2, want_state)



NCO_GEOMETRIES = [(6, 10, 24, 2 * 1024, 3), (8, 200, 64, 2 * 4096, 4), (10, 384, 16, 2 * 8192, 2), (5, 0, 32, 2 * 256, 5), (12, 4000, 6, 2 * 8192, 2)]


@pytest.mark.ref
@skip_without_ref
@pytest.mark.parametrize("bin_e,first_bin,n_channels,block_len,n_blocks", NCO_GEOMETRIES)
@pytest.mark.parametrize("custom_atan", [1, 0])
def test_chan_nco_oracle_equals_reference_built_chain(bin_e, first_bin, n_channels, block_len, n_blocks, custom_atan):
    """SURVEY 8(f)2's literal definition (rxo_chan_nco_block) == [the reference's callback scale] -> [NCO on its Sinewave table, products
    rounded by its FIX_MPY] -> [the reference's full_demod at downsample N: low_pass + fm_demod] per channel: every sample, every carry"""
    for iq in (sig_fm(n_blocks * block_len // 2, seed=91, amp=9000), sig_noise(n_blocks * block_len, seed=92)):
        want, want_pre = ref_chan_nco_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        got, pre = oracle_chan_nco_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        bad = np.argwhere(got != want)
        assert bad.size == 0, "first mismatch at %s: port %d reference %d (%d bad)" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])], len(bad))
        assert np.array_equal(pre, want_pre)


def test_nco_mode_sees_the_same_channels_as_the_bank():
    """the two definitions are the same filter bank in two fixed-point roundings: an unmodulated carrier at bin k lands in channel k of
    both, and a phase step per window comes out of both discriminators as the same frequency (to the rounding)"""
    n, k = 256, 77
    dphi = 0.4
    iq = tone(n, k, 40, 6000, dphi)
    a, _ = run(iq, 8, 70, 16, 0)
    b, _ = oracle_chan_nco_stream(iq, len(iq), 8, 70, 16, 0)
    want = dphi / 3.14159 * 16384
    assert np.all(np.abs(a[7, 2:] - want) < 80) and np.all(np.abs(b[7, 2:] - want) < 400)
