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
