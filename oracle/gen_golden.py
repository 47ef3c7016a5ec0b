#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE ITSELF (oracle/_ref).

TEST INFRASTRUCTURE.  Needs /root/reference (this container): `make -C oracle ref` compiles
rtl_fm.c / rtl_power.c / rtl_sdr.c where they lie, unmodified; this script drives those objects through
ctypes and stores inputs + outputs.  The fixtures pin oracle/rx_oracle.c (test_oracle_golden.py)
and, on the GPU box where /root/reference does not exist, the HIP path (test_gpu_golden.py).

    python oracle/gen_golden.py          # rewrites tests/golden/
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support  # noqa: E402
from rx_tools_amd.structs import TuningState  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def kats():
    """known-answer vectors of the scalar helpers, straight from the reference's functions"""
    F, P = support.ref_fm(), support.ref_power()
    rng = np.random.RandomState(1)
    yx = np.concatenate([
        np.array([(0, 1), (1, 0), (0, -1), (-1, 0), (1, 1), (1, -1), (-1, -1), (-1, 1), (3, 4), (-3, 4), (100, -7),
                  (7, -22), (0, 0), (2 ** 30, 2 ** 30), (-2 ** 31, 5), (123456789, -987654321)], dtype=np.int64),
        rng.randint(-2 ** 31, 2 ** 31, size=(200, 2)), rng.randint(-2000, 2000, size=(200, 2))]).astype(np.int32)
    at = np.array([F.fast_atan2(int(y), int(x)) for y, x in yx], dtype=np.int32)
    abcd = np.concatenate([
        np.array([(10, 20, 30, -40), (100, 0, 0, 100), (-128, 127, 127, -128), (15104, -15104, -15104, 15104),
                  (-32768, -32768, -32768, -32768), (32767, -32768, -32768, 32767)], dtype=np.int64),
        rng.randint(-32768, 32768, size=(300, 4))]).astype(np.int32)
    pdf = np.array([F.polar_disc_fast(*map(int, r)) for r in abcd], dtype=np.int32)
    pdl = np.array([F.polar_discriminant(*map(int, r)) for r in abcd], dtype=np.int32)
    ab = np.concatenate([
        np.array([(32767, 32767), (16384, 16384), (-32768, 32767), (-32768, -32768), (3, 5), (1, 16384), (1, 8192),
                  (-1, 16384), (23170, -23170)], dtype=np.int64), rng.randint(-32768, 32768, size=(300, 2))]).astype(np.int16)
    mpy = np.array([P.FIX_MPY(int(a), int(b)) for a, b in ab], dtype=np.int16)
    # the callback's cs16 scale for every int16 (offset tuning on, so no rotation)
    d, s = support.ref_fm_reset(F, offset_tuning=1)
    x = np.arange(-32768, 32768, dtype=np.int16)
    buf = x.copy()
    F.ref_fm_callback(support.ptr16(buf), C.c_uint32(65536), C.byref(s))
    scale = np.ctypeslib.as_array(d.lowpassed)[:65536].copy()
    np.savez_compressed(os.path.join(OUT, "kats.npz"), atan_yx=yx, atan_out=at, disc_in=abcd, disc_fast=pdf,
                        disc_libm=pdl, mpy_in=ab, mpy_out=mpy, scale_map=scale)


FM_CASES = [
    # name, signal, n_blocks, block_len, params
    ("wbfm_ds6", "fm", 6, 8192, dict(downsample=6)),
    ("wbfm_ds118", "fm", 3, 2 * 11800, dict(downsample=118)),
    ("wbfm_ds118_noise", "noise", 4, 8192, dict(downsample=118)),
    ("config1_240k", "fm", 3, 2 * 20352, dict(downsample=5, rate_out=240000, deemph_a=19)),
    ("alt_ds6", "alt", 4, 4096, dict(downsample=6)),
    ("zeros_ds118", "zeros", 4, 8192, dict(downsample=118)),
    ("fifth3", "fm", 5, 8192, dict(downsample_passes=3)),
    ("fifth3_fir9_noise", "noise", 5, 8192, dict(downsample_passes=3, comp_fir_size=9)),
    ("fifth7", "fm", 3, 16384, dict(downsample_passes=7)),
    ("std_atan", "fm", 3, 4096, dict(downsample=10, custom_atan=0)),
    ("odd_block", "fm", 5, 4096 + 8, dict(downsample=7)),
    ("no_deemph_no_resample", "noise", 3, 4096, dict(downsample=6, deemph=0, rate_out2=-1)),
    # the other demodulators and filters
    ("am", "fm", 3, 8192, dict(downsample=6, mode=1, output_scale=3, deemph=0)),
    ("usb", "noise", 3, 8192, dict(downsample=6, mode=2, output_scale=2)),
    ("lsb", "fm", 3, 8192, dict(downsample=6, mode=3, deemph=0, rate_out2=-1)),
    ("raw", "fm", 3, 8192, dict(downsample=6, mode=4)),
    ("atan_lut", "fm", 3, 8192, dict(downsample=10, custom_atan=2)),
    ("atan_ale", "noise", 3, 8192, dict(downsample=10, custom_atan=3)),
    ("squelch", "fm", 4, 8192, dict(downsample=6, squelch_level=2000)),
    ("adc", "noise", 4, 8192, dict(downsample=6, dc_block_audio=1)),
    ("post4", "fm", 4, 8192, dict(downsample=4, post_downsample=4)),
    ("rdc", "noise", 5, 8192, dict(downsample=6, dc_block_raw=1, rdc_block_const=3)),
]


def fm_signal(kind, n_int16, seed):
    if kind == "fm":
        return support.sig_fm(n_int16 // 2, seed=seed)
    if kind == "noise":
        return support.sig_noise(n_int16, seed=seed)
    if kind == "alt":
        return support.sig_alternating(n_int16)
    return np.zeros(n_int16, np.int16)


def fm_cases():
    F = support.ref_fm()
    out = {}
    for i, (name, kind, n_blocks, block_len, params) in enumerate(FM_CASES):
        iq = fm_signal(kind, n_blocks * block_len, 1000 + i)
        res, lens, d = support.ref_fm_stream(F, iq, block_len, **params)
        carry = np.array([d.now_r, d.now_j, d.prev_index, d.pre_r, d.pre_j, d.now_lpr, d.prev_lpr_index, d.lp_len], np.int64)
        hist = np.concatenate([np.ctypeslib.as_array(d.lp_i_hist).ravel(), np.ctypeslib.as_array(d.lp_q_hist).ravel(),
                               np.ctypeslib.as_array(d.droop_i_hist), np.ctypeslib.as_array(d.droop_q_hist)]).astype(np.int16)
        out[name + "__iq"] = iq
        out[name + "__out"] = res
        out[name + "__lens"] = lens
        out[name + "__carry"] = carry
        out[name + "__hist"] = hist
        out[name + "__block_len"] = np.array([block_len])
    np.savez_compressed(os.path.join(OUT, "fm_cases.npz"), **out)


POWER_CASES = [
    # name, range, crop, window, (boxcar, comp_fir, peak_hold), amp, passes, max_tunes
    ("cfg3_small_amp", "24M:1.7G:1k", 0.0, "rectangle", (1, 0, 0), 100, 2, 2),
    ("cfg3_full_scale", "24M:1.7G:1k", 0.0, "hamming", (1, 0, 0), 32768, 1, 2),
    ("n32_peak", "88M:108M:125k", 0.2, "blackman-harris", (1, 0, 1), 4000, 2, 3),
    ("boxcar_ds28", "100M:100.1M:100", 0.0, "rectangle", (1, 0, 0), 2000, 1, 1),
    ("fifth_ds16_fir9", "100M:100.1M:100", 0.0, "youssef", (0, 9, 0), 2000, 1, 1),
    ("rms_path", "100M:110M:1M", 0.0, "rectangle", (1, 0, 0), 5000, 2, 4),
]


def power_cases():
    P = support.ref_power()
    out = {}
    for i, (name, rng, crop, window, flags, amp, passes, max_tunes) in enumerate(POWER_CASES):
        P.ref_power_set_flags(*flags)
        n = P.ref_power_setup(rng.encode(), crop, window.encode())
        tunes = (TuningState * n).from_address(P.ref_power_tunes())
        buf_len, nb = tunes[0].buf_len, 1 << tunes[0].bin_e
        use = min(n, max_tunes)
        # the reference scans all n tunes; feed real data to the first `use`, zeros to the rest
        data = np.zeros((passes, n, buf_len), np.int16)
        data[:, :use, :] = support.sig_noise(passes * use * buf_len, seed=2000 + i, amp=amp).reshape(passes, use, buf_len)
        P.ref_power_scan(support.ptr16(np.ascontiguousarray(data)), passes)
        avg = np.stack([np.ctypeslib.as_array(tunes[t].avg, (nb,)).copy() for t in range(use)])
        samples = np.array([tunes[t].samples for t in range(use)], np.int32)
        meta = np.array([tunes[0].bin_e, buf_len, tunes[0].downsample, tunes[0].downsample_passes, tunes[0].rate, n], np.int64)
        freqs = np.array([tunes[t].freq for t in range(use)], np.int64)
        path = os.path.join(OUT, "_tmp.csv")
        P.ref_power_csv(path.encode())
        rows = open(path).read().splitlines()[:use]
        os.unlink(path)
        out[name + "__in"] = np.ascontiguousarray(data[:, :use, :])
        out[name + "__avg"] = avg
        out[name + "__samples"] = samples
        out[name + "__meta"] = meta
        out[name + "__freqs"] = freqs
        out[name + "__crop"] = np.array([tunes[0].crop])
        out[name + "__csv"] = np.array(rows)
        out[name + "__window"] = np.ctypeslib.as_array(P.ref_power_window_coefs(), (nb,)).copy()
    np.savez_compressed(os.path.join(OUT, "power_cases.npz"), **out)


def plans():
    """frequency_range() geometry for a handful of -f arguments"""
    P = support.ref_power()
    rows = []
    for rng, crop, boxcar in [("24M:1.7G:1k", 0.0, 1), ("88M:108M:125k", 0.0, 1), ("88M:108M:125k", 0.2, 1),
                              ("100M:100.1M:10", 0.0, 1), ("100M:100.1M:10", 0.0, 0), ("100M:1G:1M", 0.0, 1),
                              ("100M:100.3M:100", 0.0, 1), ("433M:435M:500", 0.5, 0), ("50M:60M:10k", 0.3, 1),
                              ("100M:102M:40", 0.0, 1), ("100M:102.8M:20", 0.0, 1), ("100M:102.8M:2", 0.0, 1), ("100M:101M:2", 0.1, 0)]:
        P.ref_power_set_flags(boxcar, 0, 0)
        n = P.ref_power_setup(rng.encode(), crop, b"rectangle")
        t = (TuningState * n).from_address(P.ref_power_tunes())
        rows.append((rng, crop, boxcar, n, t[0].bin_e, t[0].buf_len, t[0].downsample, t[0].downsample_passes, t[0].rate,
                     t[0].freq, (t[1].freq - t[0].freq) if n > 1 else 0, t[0].crop))
    np.savez_compressed(os.path.join(OUT, "plans.npz"), rows=np.array([repr(r) for r in rows]))


FM_PLAN_ARGS = [
    ["-M", "wbfm", "-f", "100M"],
    ["-M", "wbfm", "-f", "100M", "-F", "9"],
    ["-M", "wbfm", "-f", "97.3M", "-c", "eu"],
    ["-M", "wbfm", "-f", "100M", "-o", "4", "-E", "rdc"],
    ["-M", "fm", "-f", "145.5M", "-s", "240000", "-E", "deemp"],
    ["-M", "fm", "-f", "145.5M", "-s", "240000", "-E", "deemp", "-c", "120"],
    ["-M", "fm", "-f", "162.55M"],
    ["-M", "am", "-f", "118.3M", "-s", "12000"],
    ["-M", "usb", "-f", "14.2M", "-s", "6000", "-F", "0"],
    ["-M", "lsb", "-f", "7.1M", "-s", "6000", "-E", "edge"],
    ["-M", "raw", "-f", "433M", "-s", "48000", "-E", "offset"],
    ["-M", "fm", "-f", "433M", "-s", "1000000"],
    ["-M", "fm", "-f", "433M", "-s", "2000000", "-A", "lut"],
    ["-M", "fm", "-f", "433M", "-s", "300000", "-F", "9", "-r", "48000", "-A", "fast"],
]


def fm_plans():
    """the parameters rx_fm's main() derives (getopt, demod_init, rate_in *= post_downsample, optimal_settings, deemph_a),
    dumped from the reference's own globals after its main() ran on the fake device (tests/dropin_runner.py plan)"""
    import json
    import subprocess
    import tempfile
    rows = []
    runner = os.path.join(ROOT, "tests", "dropin_runner.py")
    for args in FM_PLAN_ARGS:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "plan.json")
            subprocess.run([sys.executable, runner, "plan", out] + args, capture_output=True, timeout=120)
            rows.append(json.dumps({"args": args, "state": json.load(open(out))}))
    np.savez_compressed(os.path.join(OUT, "fm_plans.npz"), rows=np.array(rows))


def sdr_cases():
    """rx_sdr -F conversions through the reference's own main(): every int16 value once, a packed CS12 stream,
    and the WAV headers of rx_fm -E wav"""
    x = np.arange(-32768, 32768, dtype=np.int16)
    out = {}
    for fmt in ("CU8", "CS8", "CF32"):
        out["all_" + fmt] = support.ref_sdr_convert(fmt, x)
    cs12 = np.random.default_rng(1212).integers(0, 256, size=3 * 4099, dtype=np.uint8)
    out["cs12_in"] = cs12
    out["cs12_out"] = support.ref_sdr_convert("CS16", cs12, chunk=777)
    F = support.ref_fm()
    hdr_args = [(32000, 0), (48000, 1), (170000, 0), (24000, 0), (1000000, 1)]
    hdrs = np.zeros((len(hdr_args), 44), dtype=np.uint8)
    for i, (rate, raw) in enumerate(hdr_args):
        buf = (C.c_ubyte * 64)()
        n = F.ref_fm_wav_header(rate, raw, buf, 64)
        assert n == 44
        hdrs[i] = np.frombuffer(bytes(buf)[:44], dtype=np.uint8)
    out["wav_args"] = np.array(hdr_args, dtype=np.int64)
    out["wav_headers"] = hdrs
    np.savez_compressed(os.path.join(OUT, "sdr_cases.npz"), **out)


if __name__ == "__main__":
    if not support.have_ref():
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)          # the reference's frequency_range reports on stderr
    try:
        kats(); fm_cases(); power_cases(); plans(); fm_plans(); sdr_cases()
    finally:
        os.dup2(saved, 2)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
