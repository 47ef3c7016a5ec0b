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
    def wbfm(cls, downsample=6, **kw):
        """`-M wbfm` defaults, rtl_fm.c:1331-1341 (deemph_a for 75 us at 170 kHz, 1410-1412)."""
        p = cls(downsample=downsample, downsample_passes=0, comp_fir_size=0, custom_atan=1, deemph=1,
                deemph_a=13, rate_out=170000, rate_out2=32000, offset_tuning=0, prescaled=0,
                mode=0, output_scale=1, squelch_level=0, dc_block_audio=0, adc_block_const=9,
                post_downsample=1, dc_block_raw=0, rdc_block_const=9)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


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
    _fields_ = [(n, C.c_int) for n in ("bin_e", "first_bin", "n_channels", "custom_atan")]


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

    def run(self, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride):
        n = C.c_size_t(0)
        check(lib().rxgpu_chan_run(self._h, d_iq_ptr, n_blocks, block_len, d_out_ptr, out_stride, C.byref(n)))
        return n.value

    @property
    def host_fixups(self):
        return lib().rxgpu_chan_host_fixups(self._h)
