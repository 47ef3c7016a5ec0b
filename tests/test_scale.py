"""CPU: the two closed forms used for rtlsdr_callback's fp64 scale (rtl_fm.c:846) reproduce it on
all 65536 inputs: the fp32-fma form the kernels use (emulated here with one rounding) and the pure
integer form kept as documentation.  The device itself is checked in test_gpu_fm.py."""
import os

import numpy as np

from support import GOLDEN_DIR


def test_scale_closed_forms_exhaustive():
    ref = np.load(os.path.join(GOLDEN_DIR, "kats.npz"))["scale_map"].astype(np.int64)
    x = np.arange(-32768, 32768, dtype=np.int64)
    assert (ref[0], ref[-1], ref[32768 - 358], ref[32768 - 359], ref[32768 + 153], ref[32768 + 154]) == (-127, 128, 0, -1, 0, 1)
    # fp32 fma: exact product + addend in fp64 (40 + 24 significant bits fit), one rounding to fp32, truncate
    c1, c2 = np.float32(128.0 / 32767.0), np.float32(0.4)
    v = (x.astype(np.float64) * np.float64(c1) + np.float64(c2)).astype(np.float32)
    assert np.array_equal(np.trunc(v).astype(np.int64), ref)
    # integer form: y + 128 = floor((128 u + c) / 32767), u = x + 32768, c = 45745 below -102 else 12978
    u = x + 32768
    n = 128 * u + np.where(x < -102, 45745, 12978)
    q = (n + (n >> 15) + 1) >> 15
    assert np.array_equal(q - 128, ref)
