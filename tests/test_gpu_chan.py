"""-m gpu: the 256-channel NBFM channeliser (BASELINE configs[4], an extension specified from reference
primitives: fix_fft per window + fm_demod per channel) against its oracle, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import rx_tools_amd as R
from support import oracle, sig_fm, sig_noise, ptr16, i16p, intp, have_ref, ref_chan_stream, oracle_chan_nco_stream, ref_chan_nco_stream

pytestmark = pytest.mark.gpu


class ChanCfg(C.Structure):
    _fields_ = [("bin_e", C.c_int), ("first_bin", C.c_int), ("n_channels", C.c_int), ("custom_atan", C.c_int), ("sinewave", i16p)]


def oracle_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan, pre=None):
    O = oracle()
    O.rxo_chan_block.argtypes = [C.POINTER(ChanCfg), i16p, C.c_int, intp, i16p, C.c_size_t]
    sw = R.sine_table(bin_e)
    cfg = ChanCfg(bin_e, first_bin, n_channels, custom_atan, ptr16(sw))
    n = 1 << bin_e
    n_blocks = len(iq) // block_len
    wpb = block_len // 2 // n
    total = wpb * n_blocks
    out = np.zeros((n_channels, total), np.int16)
    pre = np.zeros(2 * n_channels, np.int32) if pre is None else pre.copy()
    tmp = np.zeros((n_channels, wpb), np.int16)
    for b in range(n_blocks):
        blk = np.ascontiguousarray(iq[b * block_len:(b + 1) * block_len])
        O.rxo_chan_block(C.byref(cfg), ptr16(blk), block_len, pre.ctypes.data_as(intp), ptr16(tmp), wpb)
        out[:, b * wpb:(b + 1) * wpb] = tmp
    return out, pre


def gpu_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan, n_runs=1):
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    n = 1 << bin_e
    n_blocks = len(iq) // block_len
    wpb = block_len // 2 // n
    per = (n_blocks + n_runs - 1) // n_runs
    ch = R.Channeliser(R.ChanParams(bin_e, first_bin, n_channels, custom_atan), per, block_len, R.sine_table(bin_e))
    d_iq = to_dev(iq)
    outs = []
    b = 0
    while b < n_blocks:
        nb = min(per, n_blocks - b)
        d_out = torch.zeros((n_channels, nb * wpb), dtype=torch.int16, device="cuda")
        w = ch.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), nb * wpb)
        assert w == nb * wpb
        outs.append(d_out.cpu().numpy())
        b += nb
    pre = ch.get_carry()
    fix = ch.host_fixups
    ch.close()
    return np.concatenate(outs, axis=1), pre, fix


@pytest.mark.parametrize("bin_e,first_bin,n_channels,block_len,n_blocks", [
    (10, 384, 256, 2 * 131072, 3),        # configs[4]: 256 channels of a 1024-bin bank, callback blocks of 131072
    (10, 900, 256, 2 * 8192, 5),          # channel range wrapping through bin 0
    (8, 0, 256, 2 * 4096, 4),             # every bin
    (12, 100, 7, 2 * 8192, 6),            # two windows per block
    (5, 3, 20, 2 * 1024, 3),
    (9, 17, 100, 2 * 16384, 5),           # 32 windows per block: the fused demodulator, one group of 32 per block
    (11, 2000, 96, 2 * 131072, 2),        # 64 windows per block: two groups
    (10, 0, 1024, 2 * 16384, 3),          # every bin of a 1024 bank: groups of 16 (LDS), fused
])
@pytest.mark.parametrize("custom_atan", [1, 0])
def test_channeliser_bit_exact(bin_e, first_bin, n_channels, block_len, n_blocks, custom_atan):
    for iq in (sig_fm(n_blocks * block_len // 2, seed=70, amp=9000), sig_noise(n_blocks * block_len, seed=71)):
        want, want_pre = oracle_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        got, pre, _ = gpu_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        assert got.shape == want.shape
        bad = np.argwhere(got != want)
        assert bad.size == 0, "first mismatch at %s: got %d want %d (%d bad)" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])], len(bad))
        assert np.array_equal(pre, want_pre)
        if have_ref():
            # ... and directly against reference-built code: the reference's own fix_fft per window, its own full_demod per channel and block
            ref_out, ref_pre, _ = ref_chan_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
            assert np.array_equal(got, ref_out) and np.array_equal(pre, ref_pre)


def test_channeliser_carry_across_runs():
    iq = sig_noise(8 * 2 * 8192, seed=72, amp=6000)
    want, want_pre = oracle_chan(iq, 2 * 8192, 10, 10, 64, 1)
    got, pre, _ = gpu_chan(iq, 2 * 8192, 10, 10, 64, 1, n_runs=4)
    assert np.array_equal(got, want) and np.array_equal(pre, want_pre)


def test_channeliser_fused_carry_across_runs():
    """the fused path (-A fast, whole groups of 32 windows per block) run by run"""
    iq = sig_fm(6 * 32768, seed=73, amp=12000)
    want, want_pre = oracle_chan(iq, 2 * 32768, 10, 500, 256, 1)      # 32 windows per block
    got, pre, _ = gpu_chan(iq, 2 * 32768, 10, 500, 256, 1, n_runs=3)
    assert np.array_equal(got, want) and np.array_equal(pre, want_pre)


@pytest.mark.parametrize("flag_all", ["1", "2"])
@pytest.mark.parametrize("bin_e,first_bin,n_channels,block_len,n_blocks,custom_atan", [
    (10, 384, 24, 2 * 65536, 5, 1),       # fused: 64 windows per block, the kernel keeps [channel][run] first / last windows only
    (10, 384, 24, 2 * 131072, 3, 1),      # fused, two runs of 64 per block: only every other run start is a libm sample
    (10, 20, 8, 2 * 4096, 6, 0),          # dense: -A std, every sample is libm (4 windows per block)
    (9, 17, 30, 2 * 16384, 4, 1),         # fused with groups of 16 in a run of 32
])
def test_channeliser_host_fixups_forced(monkeypatch, flag_all, bin_e, first_bin, n_channels, block_len, n_blocks, custom_atan):
    """$RXGPU_FLAG_ALL hands every libm sample of the channeliser to the host (2: after storing a wrong value), so the host's addressing of
    the bins the FFT kernel kept -- compact [channel][run] edges in the fused form, the dense array otherwise -- is what makes the output right"""
    monkeypatch.setenv("RXGPU_FLAG_ALL", flag_all)
    iq = sig_fm(n_blocks * block_len // 2, seed=75, amp=9000)
    want, want_pre = oracle_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
    got, pre, fix = gpu_chan(iq, block_len, bin_e, first_bin, n_channels, custom_atan, n_runs=2)
    assert fix > 0
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatch at %s: got %d want %d (%d bad)" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])], len(bad))
    assert np.array_equal(pre, want_pre)


@pytest.mark.parametrize("gpw", [4, 2, 1])
def test_channeliser_window_groups_and_runs(gpw):
    """every shape of the fused kernel's run -- 16 windows per group x the 4, 2 or 1 groups per workgroup that divide a callback block's windows --
    gives the same samples"""
    for bin_e, first_bin, n_channels, blocks16, n_blocks in ((10, 384, 256, 8, 3), (9, 500, 100, 12, 5), (11, 2000, 96, 4, 3)):
        # a block of (blocks16 + odd part) * gpw groups of 16 windows: gpw = 4 -> a multiple of 64 windows, 2 -> of 32 only, 1 -> of 16 only
        wpb = 16 * (blocks16 * 4 if gpw == 4 else (blocks16 * 2 + 1) * 2 if gpw == 2 else blocks16 * 4 + 1)
        block_len = 2 * (wpb << bin_e)
        iq = sig_fm(n_blocks * block_len // 2, seed=76, amp=9000)
        want, want_pre = oracle_chan(iq, block_len, bin_e, first_bin, n_channels, 1)
        got, pre, _ = gpu_chan(iq, block_len, bin_e, first_bin, n_channels, 1, n_runs=2)
        assert np.array_equal(got, want) and np.array_equal(pre, want_pre)


def test_channeliser_finds_the_carrier():
    """sanity of the specification itself: an unmodulated carrier at bin k*fs/N shows up in channel k only"""
    n, k = 1024, 300
    t = np.arange(64 * n)
    ph = 2 * np.pi * k / n * t
    iq = np.empty(2 * len(t), np.int16)
    # amplitude 500: above ~724 the reference's fast_atan2 wraps in int32 (rtl_fm.c:498), by design reproduced
    iq[0::2] = np.rint(500 * np.cos(ph))
    iq[1::2] = np.rint(500 * np.sin(ph))
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    ch = R.Channeliser(R.ChanParams(10, 0, 1024, 1), 1, len(iq), R.sine_table(10))
    d_out = torch.zeros((1024, 64), dtype=torch.int16, device="cuda")
    ch.run(to_dev(iq).data_ptr(), 1, len(iq), d_out.data_ptr(), 64)
    ch.close()
    # the decimated IQ of channel k is a constant phasor -> discriminator output 0 after the first window
    out = d_out.cpu().numpy()
    assert np.all(np.abs(out[k, 1:]) <= 40), out[k, :8]
    # and next to nothing leaks into far-away channels: their bins stay (near) zero
    assert np.count_nonzero(out[(k + 512) % 1024]) == 0


# ----------------------------------------------------------------------------- round 2: per-channel audio stages

def oracle_chan_audio(iq, block_len, bin_e, first_bin, n_channels, custom_atan, deemph, a, rate_out, rate_out2):
    """every channel a demod_state of its own: after fm_demod, deemph_filter and low_pass_real on the channel's samples, callback block
    after callback block, with the channel's own carried avg / now_lpr / prev_lpr_index (support.oracle_chan_stream; pinned against the
    reference's own fix_fft + full_demod in tests/test_chan_oracle.py)"""
    from support import oracle_chan_stream
    return oracle_chan_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan, deemph, a, rate_out, rate_out2)


@pytest.mark.parametrize("bin_e,n_channels,block_len,n_blocks,deemph,a,rate_out,rate_out2,custom_atan", [
    (10, 256, 2 * 131072, 3, 1, 2, 19531, 8000, 1),       # configs[4] bank: 75 us at 19.5 kHz gives a = 2 (even, generic step), -r 8000
    (8, 64, 2 * 65536, 4, 1, 13, 170000, 32000, 1),       # the wbfm constants on a coarse bank: odd a, three-instruction range
    (8, 64, 2 * 65536, 4, 1, 13, 170000, -1, 0),          # de-emphasis only, in place; -A std
    (9, 100, 2 * 32768, 5, 0, 0, 48000, 8000, 1),         # resampler only, ratio 6
    (8, 32, 2 * 65536, 3, 1, 64, 24000, 12000, 1),        # a at the top of the mask range
    (8, 32, 2 * 16384, 3, 1, 200, 24000, 6000, 1),        # a > 64: one thread per channel
    (8, 16, 2 * 1024, 6, 1, 9, 24000, 8000, 1),           # four windows per block: shorter than any warm-up
    (8, 24, 2 * 65536, 4, 1, 5, 24000, 8000, 1),          # k_ch_audio at a = 5, 6, 8 (24-bit division from a = 5)
    (8, 24, 2 * 65536, 4, 1, 6, 24000, 12000, 1),
    (8, 24, 2 * 65536, 4, 1, 8, 24000, -1, 1),
])
def test_channeliser_per_channel_audio_stages(bin_e, n_channels, block_len, n_blocks, deemph, a, rate_out, rate_out2, custom_atan):
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    iq = sig_noise(n_blocks * block_len, seed=4 + bin_e, amp=2500)
    want, want_pre, want_state = oracle_chan_audio(iq, block_len, bin_e, 5, n_channels, custom_atan, deemph, a, rate_out, rate_out2)
    n = 1 << bin_e
    wpb = block_len // 2 // n
    per = (n_blocks + 1) // 2                                  # two runs: the per-channel carries cross a run boundary
    ch = R.Channeliser(R.ChanParams(bin_e, 5, n_channels, custom_atan, deemph, a, rate_out, rate_out2), per, block_len, R.sine_table(bin_e))
    d_iq = to_dev(iq)
    outs, b = [], 0
    while b < n_blocks:
        nb = min(per, n_blocks - b)
        d_out = torch.zeros((n_channels, nb * wpb), dtype=torch.int16, device="cuda")
        w = ch.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), nb * wpb)
        outs.append(d_out[:, :w].cpu().numpy())
        b += nb
    got = np.concatenate(outs, axis=1)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    if have_ref():
        ref_out, ref_pre, ref_state = ref_chan_stream(iq, block_len, bin_e, 5, n_channels, custom_atan, deemph, a, rate_out, rate_out2)
        assert np.array_equal(got, ref_out) and np.array_equal(ch.get_carry(), ref_pre) and np.array_equal(ch.get_audio_carry().reshape(n_channels, 3), ref_state)
    assert np.array_equal(ch.get_carry(), want_pre)
    state = ch.get_audio_carry().reshape(n_channels, 3)
    bad = np.argwhere(state != want_state)
    assert bad.size == 0, "first differing carries (channel, field): %s got %s want %s" % (bad[:5].tolist(), state[bad[:5, 0]].tolist(), want_state[bad[:5, 0]].tolist())
    ch.close()


# round 5: runs in flight
@pytest.mark.parametrize("custom_atan,flag_all", [(1, None), (0, None), (1, "2"), (0, "1")])
def test_channeliser_async_runs_chain_their_carries_on_the_device(custom_atan, flag_all, monkeypatch):
    """rxgpu_chan_run_async: seven runs of different lengths enqueued back to back, two in flight, carries chained on the device, one
    rxgpu_chan_wait at the end == the oracle's stream over the concatenation, == the same runs through the synchronous call; also with the
    host settling libm samples run by run ($RXGPU_FLAG_ALL: every block-first / -A std sample handed to the host, 2: with a wrong value first)"""
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    if flag_all:
        monkeypatch.setenv("RXGPU_FLAG_ALL", flag_all)
    bin_e, first_bin, n_channels, block_len = 7, 20, 48, 2 * 2048
    lens = [3, 1, 4, 1, 5, 2, 6] if not (flag_all and custom_atan == 0) else [1, 1, 1]     # -A std flags every sample: the cap is per run
    n_blocks = sum(lens)
    iq = sig_noise(n_blocks * block_len, seed=77, amp=3000)
    from support import oracle_chan_stream
    want, want_pre, _ = oracle_chan_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
    wpb = block_len // 2 >> bin_e
    d_iq = to_dev(iq)
    res = {}
    for mode in ("async", "sync"):
        ch = R.Channeliser(R.ChanParams(bin_e, first_bin, n_channels, custom_atan), max(lens), block_len, R.sine_table(bin_e))
        outs, b = [], 0
        for nb in lens:
            d_out = torch.zeros((n_channels, nb * wpb), dtype=torch.int16, device="cuda")
            outs.append(d_out)
            if mode == "async":
                ch.run_async(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), nb * wpb)
            else:
                assert ch.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), nb * wpb) == nb * wpb
            b += nb
        if mode == "async":
            assert ch.wait() == lens[-1] * wpb
            if flag_all:
                assert ch.host_fixups > 0
        res[mode] = (np.concatenate([o.cpu().numpy() for o in outs], axis=1), ch.get_carry())
        ch.close()
    for got, pre in res.values():
        assert np.array_equal(got, want) and np.array_equal(pre, want_pre)
    R.lib().rxgpu_knobs_reload()


# round 5: the audio stages on a (segment, channel) grid -- rows long enough to be cut into >= 64 chunks of >= warm-up length
@pytest.mark.parametrize("bin_e,n_channels,block_len,n_blocks,a,rate_out,rate_out2,custom_atan,pad", [
    (6, 40, 2 * 65536, 10, 7, 19531, 8000, 1, 0),         # 5120-sample rows: 80 chunks, one segment; odd a below the 24-bit step
    (4, 12, 2 * 65536, 10, 7, 19531, 8000, 1, 0),         # 20480-sample rows: 320 chunks, two segments
    (4, 16, 2 * 65536, 10, 13, 170000, 32000, 0, 0),      # the wbfm constants: odd a in the three-instruction range; -A std
    (4, 9, 2 * 65536, 10, 2, 19531, -1, 1, 0),            # even a, de-emphasis only (in place), 8 segments
    (4, 9, 2 * 65536, 6, 64, 24000, 12000, 1, 5),         # a at the top of the mask range; rows that start off any 16-byte boundary
    (5, 20, 2 * 131072, 8, 19, 240000, 32000, 1, 0),      # 16384-sample rows, a = 19
    (4, 6, 2 * 64, 802, 2, 19531, 8000, 1, 0),            # 1604-sample rows of a scratch whose odd rows sit off the 16-byte grid: the sample-by-sample loops
    (4, 6, 2 * 65536, 6, 7, 48000, 32000, 1, 3),          # ratio 1 (two outputs per three samples): the most outputs a workgroup can stage
    (4, 8, 2 * 65536, 6, 5, 24000, 8000, 1, 0),           # a = 5, 6, 8: the smallest that take the two-multiply 24-bit division (odd and even step)
    (4, 8, 2 * 65536, 6, 6, 24000, 12000, 1, 0),
    (4, 8, 2 * 65536, 6, 8, 24000, -1, 1, 0),
])
def test_channeliser_audio_stages_segmented(bin_e, n_channels, block_len, n_blocks, a, rate_out, rate_out2, custom_atan, pad):
    """the (segment, channel) form of the per-channel audio stages == the oracle (and the reference where built), across a run boundary
    (rows too short for the grid take the one-workgroup-per-channel kernel: test_channeliser_audio_stages covers those)"""
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    iq = sig_noise(n_blocks * block_len, seed=40 + bin_e + a, amp=2500)
    want, want_pre, want_state = oracle_chan_audio(iq, block_len, bin_e, 3, n_channels, custom_atan, 1, a, rate_out, rate_out2)
    n = 1 << bin_e
    wpb = block_len // 2 // n
    per = (n_blocks + 1) // 2
    d_iq = to_dev(iq)
    results = []
    for _ in range(1):
        ch = R.Channeliser(R.ChanParams(bin_e, 3, n_channels, custom_atan, 1, a, rate_out, rate_out2), per, block_len, R.sine_table(bin_e))
        outs, b = [], 0
        while b < n_blocks:
            nb = min(per, n_blocks - b)
            stride = nb * wpb + pad
            d_out = torch.zeros((n_channels, stride), dtype=torch.int16, device="cuda")
            w = ch.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), stride)
            outs.append(d_out[:, :w].cpu().numpy())
            b += nb
        results.append((np.concatenate(outs, axis=1), ch.get_carry(), ch.get_audio_carry().reshape(n_channels, 3)))
        ch.close()
    for got, pre, state in results:
        assert got.shape == want.shape and np.array_equal(got, want)
        assert np.array_equal(pre, want_pre) and np.array_equal(state, want_state)
    if have_ref():
        ref_out, ref_pre, ref_state = ref_chan_stream(iq, block_len, bin_e, 3, n_channels, custom_atan, 1, a, rate_out, rate_out2)
        assert np.array_equal(results[0][0], ref_out) and np.array_equal(results[0][1], ref_pre) and np.array_equal(results[0][2], ref_state)


def test_channeliser_audio_segmented_hostile_rows():
    """rows on which trajectories never merge (constant input: every chunk keeps its full candidate range) and rows that slam between the
    extremes: the segmented form's candidate walk and chunk tables still give the serial filter's samples"""
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    bin_e, n_channels, block_len, n_blocks, a = 4, 16, 2 * 65536, 4, 7
    n = 1 << bin_e
    for kind in ("zeros", "dc", "alternate"):
        if kind == "zeros":
            iq = np.zeros(n_blocks * block_len, np.int16)
        elif kind == "dc":
            iq = np.full(n_blocks * block_len, 20000, np.int16)
        else:
            iq = np.tile(np.array([32767, 32767, -32768, -32768], np.int16), n_blocks * block_len // 4)
        want, want_pre, want_state = oracle_chan_audio(iq, block_len, bin_e, 0, n_channels, 1, 1, a, 19531, 8000)
        ch = R.Channeliser(R.ChanParams(bin_e, 0, n_channels, 1, 1, a, 19531, 8000), n_blocks, block_len, R.sine_table(bin_e))
        wpb = block_len // 2 // n
        d_out = torch.zeros((n_channels, n_blocks * wpb), dtype=torch.int16, device="cuda")
        w = ch.run(to_dev(iq).data_ptr(), n_blocks, block_len, d_out.data_ptr(), n_blocks * wpb)
        assert w == want.shape[1] and np.array_equal(d_out[:, :w].cpu().numpy(), want), kind
        assert np.array_equal(ch.get_audio_carry().reshape(n_channels, 3), want_state), kind
        ch.close()


# ----------------------------------------------------------------------------- round 4: the NCO -> low_pass mode (SURVEY 8(f)2's literal definition)

@pytest.mark.parametrize("bin_e,first_bin,n_channels,block_len,n_blocks", [
    (10, 384, 256, 2 * 8192, 3),          # the configs[4] bank, eight windows per block
    (8, 200, 64, 2 * 4096, 4),            # channels wrapping through bin 0
    (6, 10, 24, 2 * 1024, 3),
    (10, 0, 700, 2 * 2048, 2),            # more channels than a workgroup has threads
    (12, 4000, 6, 2 * 8192, 2),           # the largest window of this mode
    (3, 1, 5, 2 * 64, 6),
])
@pytest.mark.parametrize("custom_atan", [1, 0])
def test_channeliser_nco_mode_bit_exact(bin_e, first_bin, n_channels, block_len, n_blocks, custom_atan):
    """rxgpu_chan_params.nco = 1: callback scale -> per-channel integer NCO -> low_pass at downsample N -> fm_demod, against the oracle's
    restatement and (where oracle/_ref travels) the reference-built chain; full-scale noise included (low_pass's int16 store wraps)"""
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    n = 1 << bin_e
    wpb = block_len // 2 // n
    for iq in (sig_fm(n_blocks * block_len // 2, seed=93, amp=9000), sig_noise(n_blocks * block_len, seed=94)):
        want, want_pre = oracle_chan_nco_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
        per = (n_blocks + 1) // 2                               # two runs: the carries cross a run boundary
        ch = R.Channeliser(R.ChanParams(bin_e, first_bin, n_channels, custom_atan, 0, 0, 0, -1, 1), per, block_len, R.sine_table(bin_e))
        d_iq = to_dev(iq)
        outs, b = [], 0
        while b < n_blocks:
            nb = min(per, n_blocks - b)
            d_out = torch.zeros((n_channels, nb * wpb), dtype=torch.int16, device="cuda")
            w = ch.run(d_iq.data_ptr() + b * block_len * 2, nb, block_len, d_out.data_ptr(), nb * wpb)
            assert w == nb * wpb
            outs.append(d_out.cpu().numpy())
            b += nb
        got, pre = np.concatenate(outs, axis=1), ch.get_carry()
        ch.close()
        bad = np.argwhere(got != want)
        assert bad.size == 0, "first mismatch at %s: got %d want %d (%d bad)" % (bad[0], got[tuple(bad[0])], want[tuple(bad[0])], len(bad))
        assert np.array_equal(pre, want_pre)
        if have_ref() and n_channels <= 256:
            ref_out, ref_pre = ref_chan_nco_stream(iq, block_len, bin_e, first_bin, n_channels, custom_atan)
            assert np.array_equal(got, ref_out) and np.array_equal(pre, ref_pre)


def test_channeliser_nco_mode_with_audio_stages_and_limits():
    """the per-channel audio stages behind the NCO mode are the bank's; windows beyond 2^12 and nco values other than 0/1 are refused"""
    from gpu_support import to_dev, torch_cuda
    from support import oracle
    torch = torch_cuda()
    bin_e, n_channels, block_len, n_blocks = 8, 32, 2 * 65536, 3
    iq = sig_noise(n_blocks * block_len, seed=95, amp=6000)
    dem, _ = oracle_chan_nco_stream(iq, block_len, bin_e, 5, n_channels, 1)
    O = oracle()
    O.rxo_deemph.argtypes = [i16p, C.c_int, C.c_int, intp]
    O.rxo_low_pass_real.argtypes = [i16p, C.c_int, C.c_int, C.c_int, intp, intp]
    wpb = block_len // 2 >> bin_e
    want = []
    for c in range(n_channels):
        avg, now, idx = C.c_int(0), C.c_int(0), C.c_int(0)
        rows = []
        for b in range(n_blocks):
            row = np.ascontiguousarray(dem[c, b * wpb:(b + 1) * wpb])
            O.rxo_deemph(ptr16(row), wpb, 13, C.byref(avg))
            k = O.rxo_low_pass_real(ptr16(row), wpb, 170000, 32000, C.byref(now), C.byref(idx))
            rows.append(row[:k].copy())
        want.append(np.concatenate(rows))
    want = np.stack(want)
    ch = R.Channeliser(R.ChanParams(bin_e, 5, n_channels, 1, 1, 13, 170000, 32000, 1), n_blocks, block_len, R.sine_table(bin_e))
    d_out = torch.zeros((n_channels, n_blocks * wpb), dtype=torch.int16, device="cuda")
    w = ch.run(to_dev(iq).data_ptr(), n_blocks, block_len, d_out.data_ptr(), n_blocks * wpb)
    ch.close()
    assert w == want.shape[1] and np.array_equal(d_out[:, :w].cpu().numpy(), want)
    for bad in (R.ChanParams(13, 0, 4, 1, 0, 0, 0, -1, 1), R.ChanParams(8, 0, 4, 1, 0, 0, 0, -1, 2)):
        with pytest.raises(R.RxGpuError):
            R.Channeliser(bad, 1, 2 * (1 << bad.bin_e), R.sine_table(bad.bin_e))
