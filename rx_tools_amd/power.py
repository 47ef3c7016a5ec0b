"""Python mirror of the rx_power part of include/rxgpu.h."""
import ctypes as C

import numpy as np

from ._lib import lib, check


class PowerParams(C.Structure):
    """struct rxgpu_power_params"""
    _fields_ = [(n, C.c_int) for n in (
        "bin_e", "buf_len", "downsample", "downsample_passes", "boxcar", "comp_fir_size", "peak_hold")]


class PowerPlan(C.Structure):
    """struct rxgpu_power_plan: what frequency_range (rtl_power.c:431-543) decides."""
    _fields_ = [
        ("tune_count", C.c_int), ("bin_e", C.c_int), ("buf_len", C.c_int), ("downsample", C.c_int),
        ("downsample_passes", C.c_int), ("rate", C.c_int),
        ("first_freq", C.c_int64), ("bw_seen", C.c_int64), ("crop", C.c_double),
    ]


def plan_range(rng, crop=0.0, boxcar=1):
    p = PowerPlan()
    check(lib().rxgpu_power_plan_range(rng.encode(), crop, boxcar, C.byref(p)))
    return p


def sine_table(log2n):
    n = 1 << log2n
    out = np.zeros(max(1, n * 3 // 4), dtype=np.int16)
    check(lib().rxgpu_sine_table(log2n, out.ctypes.data))
    return out


def window_coefs(name, length):
    out = np.zeros(length, dtype=np.int32)
    check(lib().rxgpu_window_coefs(name.encode(), length, out.ctypes.data))
    return out


class PowerScan:
    """rxgpu_power_scan: scanner()'s per-tune chain over many tunes/passes resident in HBM."""

    def __init__(self, params, max_tunes, window, sinewave):
        self._h = C.c_void_p()
        self.params = params
        w = np.ascontiguousarray(window, dtype=np.int32) if window is not None else None
        s = np.ascontiguousarray(sinewave, dtype=np.int16) if sinewave is not None else None
        check(lib().rxgpu_power_scan_create(C.byref(self._h), C.byref(params), max_tunes,
                                            w.ctypes.data if w is not None else None,
                                            s.ctypes.data if s is not None else None))

    def close(self):
        if self._h:
            lib().rxgpu_power_scan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, d_in_ptr, passes, tunes, d_avg_ptr, d_samples_ptr):
        """Asynchronous on the library stream; call lib().rxgpu_sync() before reading results."""
        check(lib().rxgpu_power_scan_run(self._h, d_in_ptr, passes, tunes, d_avg_ptr, d_samples_ptr))
