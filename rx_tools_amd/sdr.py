"""rx_sdr's -F output conversions and rx_fm's WAV header (include/rxgpu.h, rtl_sdr.c:354-391, rtl_fm.c:1174-1206).

ctypes plumbing over librxgpu only; torch tensors are the device buffers.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check

# name of the -F output format -> (conversion id, numpy dtype of the output, outputs per complex element)
SDR_CONVERSIONS = {
    "CU8": (0, np.uint8, 2),
    "CS8": (1, np.int8, 2),
    "CF32": (2, np.float32, 2),
    "CS16": (3, np.int16, 2),          # from packed CS12 input
}


def _n_elems(fmt, n_in):
    return n_in // 3 if fmt == "CS16" else n_in // 2


def sdr_convert(fmt, d_in, d_out=None):
    """Device path.  d_in: cuda tensor, int16 (interleaved I,Q) or uint8 (packed CS12 when fmt == "CS16").
    Returns a cuda tensor in the output format; asynchronous on the library stream (lib().rxgpu_sync())."""
    import torch
    conv, dt, per = SDR_CONVERSIONS[fmt]
    n = _n_elems(fmt, d_in.numel())
    tdt = {np.uint8: torch.uint8, np.int8: torch.int8, np.float32: torch.float32, np.int16: torch.int16}[dt]
    if d_out is None:
        d_out = torch.empty(per * n, dtype=tdt, device=d_in.device)
    check(lib().rxgpu_sdr_convert(conv, d_in.data_ptr(), n, d_out.data_ptr()))
    return d_out


def sdr_convert_host(fmt, data):
    """Host path (what rx_sdr's read loop calls): numpy in, numpy out, synchronous."""
    conv, dt, per = SDR_CONVERSIONS[fmt]
    data = np.ascontiguousarray(data)
    n = _n_elems(fmt, data.size)
    out = np.empty(per * n, dtype=dt)
    check(lib().rxgpu_sdr_convert_host(conv, data.ctypes.data, n, out.ctypes.data))
    return out


def wav_header(rate, raw_mode=False):
    buf = (C.c_ubyte * 44)()
    lib().rxgpu_wav_header(int(rate), 1 if raw_mode else 0, buf)
    return bytes(buf)
