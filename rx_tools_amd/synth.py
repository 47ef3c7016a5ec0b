"""Synthetic cs16 IQ for tests and bench.py (SURVEY.md section 8(d) signal set).

Deterministic from a seed: a 32-bit LCG x = x*1664525 + 1013904223 drives all noise.
No network, no files: there are no datasets for this path.
"""
import numpy as np


def lcg_stream(n, seed):
    """32-bit LCG x = x*1664525 + 1013904223; returns uint32[n]."""
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    out = np.empty(n, dtype=np.uint32)
    x = np.uint64(seed & 0xFFFFFFFF)
    # block-vectorised: x_{k+j} = A_j x_k + C_j
    B = 4096
    A = np.empty(B, dtype=np.uint64)
    Cc = np.empty(B, dtype=np.uint64)
    aa, cc = np.uint64(1), np.uint64(0)
    mask = np.uint64(0xFFFFFFFF)
    for j in range(B):
        aa = (aa * a) & mask
        cc = (cc * a + c) & mask
        A[j] = aa
        Cc[j] = cc
    pos = 0
    while pos < n:
        m = min(B, n - pos)
        vals = (A[:m] * x + Cc[:m]) & mask
        out[pos:pos + m] = vals.astype(np.uint32)
        x = vals[m - 1]
        pos += m
    return out


def sig_fm(n_complex, seed=12345, fs=20.06e6, amp=20000.0, tone=1000.0, dev=75e3, noise=128):
    """Signal (A): FM carrier at -fs/4 (rotate16_90 brings it to DC), 1 kHz tone,
    75 kHz deviation, plus uniform noise of +-`noise` LSB from the seeded LCG."""
    t = np.arange(n_complex, dtype=np.float64)
    phase = 2 * np.pi * (-0.25) * t + (dev / tone) * np.sin(2 * np.pi * tone / fs * t)
    r = lcg_stream(2 * n_complex, seed)
    nz = ((r >> 16).astype(np.int64) % (2 * noise + 1)) - noise
    i = np.rint(amp * np.cos(phase)).astype(np.int64) + nz[0::2]
    q = np.rint(amp * np.sin(phase)).astype(np.int64) + nz[1::2]
    out = np.empty(2 * n_complex, dtype=np.int16)
    out[0::2] = np.clip(i, -32768, 32767)
    out[1::2] = np.clip(q, -32768, 32767)
    return out


def sig_noise(n_int16, seed=777, amp=32768):
    """Signal (B): uniform noise in [-amp, amp) from the seeded LCG (full scale forces
    every int16/int32 wrap on the path)."""
    r = lcg_stream(n_int16, seed)
    v = ((r >> 8).astype(np.int64) % (2 * amp)) - amp
    return np.clip(v, -32768, 32767).astype(np.int16)


def sig_alternating(n_int16):
    """Signal (C3): +-32768/32767 alternation."""
    out = np.empty(n_int16, dtype=np.int16)
    out[0::2] = -32768
    out[1::2] = 32767
    out[2::4] = 32767
    out[3::4] = -32768
    return out
