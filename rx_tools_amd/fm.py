"""Python mirror of the rx_fm part of include/rxgpu.h (batched stream object)."""
import ctypes as C

from ._lib import lib, check


class FmParams(C.Structure):
    """struct rxgpu_fm_params: the demod_state fields full_demod reads (rtl_fm.c:136-151)."""
    _fields_ = [(n, C.c_int) for n in (
        "downsample", "downsample_passes", "comp_fir_size", "custom_atan", "deemph", "deemph_a",
        "rate_out", "rate_out2", "offset_tuning", "prescaled",
        "mode", "output_scale", "squelch_level", "dc_block_audio", "adc_block_const",
        "post_downsample", "dc_block_raw", "rdc_block_const")]

    @classmethod
    def for_mode(cls, mode, freq=100000000, rate_in=None, fifth_order=None, edge=0, time_constant_us=75, **kw):
        """What rx_fm's main() derives for `-M mode [-s rate_in] [-F fifth_order] ...`: demod_init + the -M switch
        (rxgpu_fm_params_init), then optimal_settings and deemph_a (rxgpu_fm_plan_settings).  kw: fields set between the
        two, like command-line flags (post_downsample, offset_tuning, deemph ...).  Returns (params, plan)."""
        p = cls()
        rin = C.c_int(0)
        check(lib().rxgpu_fm_params_init(C.byref(p), mode.encode(), C.byref(rin)))
        if rate_in is not None:                     # -s sets both, rtl_fm.c:1255-1258
            rin.value = rate_in
            p.rate_out = rate_in
        if fifth_order is not None:                 # -F, rtl_fm.c:1305-1308
            p.downsample_passes = 1
            p.comp_fir_size = fifth_order
        for k, v in kw.items():
            setattr(p, k, v)
        plan = FmPlan()
        check(lib().rxgpu_fm_plan_settings(C.byref(p), freq, rin.value, edge, time_constant_us, C.byref(plan)))
        return p, plan

    @classmethod
    def wbfm(cls, downsample=None, **kw):
        """`-M wbfm` as main() leaves it (rtl_fm.c:1331-1341, 960-997, 1410-1415: downsample 6, deemph_a 13 for 75 us at
        170 kHz); `downsample` / `downsample_passes` / any other field can then be overridden, e.g. the 20 Msps geometry
        of BASELINE configs[1] is downsample=118."""
        p, _ = cls.for_mode("wbfm")
        if downsample is not None:
            p.downsample = downsample
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class FmPlan(C.Structure):
    """struct rxgpu_fm_plan: what optimal_settings (rtl_fm.c:960-997) and the deemph_a formula (1410-1415) decide."""
    _fields_ = [("rate_in", C.c_int), ("downsample", C.c_int), ("downsample_passes", C.c_int), ("output_scale", C.c_int),
                ("deemph_a", C.c_int), ("capture_freq", C.c_uint32), ("capture_rate", C.c_uint32)]


class FmCarry(C.Structure):
    """struct rxgpu_fm_carry: everything the chain carries between calls."""
    _fields_ = [
        ("now_r", C.c_int), ("now_j", C.c_int), ("prev_index", C.c_int),
        ("pre_r", C.c_int), ("pre_j", C.c_int),
        ("lp_i_hist", (C.c_int16 * 6) * 10), ("lp_q_hist", (C.c_int16 * 6) * 10),
        ("droop_i_hist", C.c_int16 * 9), ("droop_q_hist", C.c_int16 * 9),
        ("deemph_avg", C.c_int), ("now_lpr", C.c_int), ("prev_lpr_index", C.c_int),
        ("squelch_hits", C.c_int), ("dc_avg", C.c_int), ("dc_avgI", C.c_int), ("dc_avgQ", C.c_int),
    ]


class FmStream:
    """rxgpu_fm_stream: callback pre-stage + full_demod over many blocks resident in HBM."""

    def __init__(self, params, max_blocks, block_len):
        self._h = C.c_void_p()
        self.params = params
        check(lib().rxgpu_fm_stream_create(C.byref(self._h), C.byref(params), max_blocks, block_len))

    def close(self):
        if self._h:
            lib().rxgpu_fm_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_carry(self, carry):
        check(lib().rxgpu_fm_stream_set_carry(self._h, C.byref(carry)))

    def get_carry(self):
        c = FmCarry()
        check(lib().rxgpu_fm_stream_get_carry(self._h, C.byref(c)))
        return c

    def run(self, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_cap, want_block_lens=False):
        """d_iq_ptr / d_out_ptr: device addresses (int).  Returns (out_len, block_lens|None)."""
        n = C.c_size_t(0)
        lens = (C.c_int * n_blocks)() if want_block_lens else None
        check(lib().rxgpu_fm_stream_run(self._h, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_cap, C.byref(n), lens))
        return n.value, (list(lens) if lens is not None else None)

    def run_async(self, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_cap, want_block_lens=False):
        """Enqueue and return (pipelined with the previous run); call wait() before reading results."""
        n = C.c_size_t(0)
        lens = (C.c_int * n_blocks)() if want_block_lens else None
        check(lib().rxgpu_fm_stream_run_async(self._h, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_cap, C.byref(n), lens))
        return n.value, (list(lens) if lens is not None else None)

    def wait(self):
        check(lib().rxgpu_fm_stream_wait(self._h))

    def run_host(self, h_iq_ptr, n_blocks, block_len, h_out_ptr, out_cap, want_block_lens=False):
        n = C.c_size_t(0)
        lens = (C.c_int * n_blocks)() if want_block_lens else None
        check(lib().rxgpu_fm_stream_run_host(self._h, h_iq_ptr, n_blocks, block_len, h_out_ptr, out_cap, C.byref(n), lens))
        return n.value, (list(lens) if lens is not None else None)

    @property
    def host_fixups(self):
        return lib().rxgpu_fm_stream_host_fixups(self._h)


class ChanParams(C.Structure):
    """struct rxgpu_chan_params (channeliser extension, include/rxgpu.h)"""
    _fields_ = [(n, C.c_int) for n in ("bin_e", "first_bin", "n_channels", "custom_atan", "deemph", "deemph_a", "rate_out", "rate_out2", "nco")]


class Channeliser:
    """rxgpu_chan: fix_fft per window + fm_demod per channel, capture resident in HBM."""

    def __init__(self, params, max_blocks, block_len, sinewave):
        import numpy as np
        self._h = C.c_void_p()
        self.params = params
        sw = np.ascontiguousarray(sinewave, dtype=np.int16)
        check(lib().rxgpu_chan_create(C.byref(self._h), C.byref(params), max_blocks, block_len, sw.ctypes.data))

    def close(self):
        if self._h:
            lib().rxgpu_chan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_carry(self, pre):
        check(lib().rxgpu_chan_set_carry(self._h, pre.ctypes.data))

    def get_carry(self):
        import numpy as np
        pre = np.zeros(2 * self.params.n_channels, dtype=np.int32)
        check(lib().rxgpu_chan_get_carry(self._h, pre.ctypes.data))
        return pre

    def set_audio_carry(self, audio):
        check(lib().rxgpu_chan_set_audio_carry(self._h, audio.ctypes.data))

    def get_audio_carry(self):
        import numpy as np
        a = np.zeros(3 * self.params.n_channels, dtype=np.int32)
        check(lib().rxgpu_chan_get_audio_carry(self._h, a.ctypes.data))
        return a

    def run(self, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride):
        n = C.c_size_t(0)
        check(lib().rxgpu_chan_run(self._h, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride, C.byref(n)))
        return n.value

    def run_async(self, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride):
        """up to two runs in flight, carries chained on the device; wait() retires them"""
        check(lib().rxgpu_chan_run_async(self._h, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride))

    def wait(self):
        n = C.c_size_t(0)
        check(lib().rxgpu_chan_wait(self._h, C.byref(n)))
        return n.value

    @property
    def host_fixups(self):
        return lib().rxgpu_chan_host_fixups(self._h)
