"""End to end through the reference's OWN main(), threads and SoapySDR read loop:
  * CPU: the untouched reference (libref_fm.so / libref_power.so = rtl_fm.c / rtl_power.c compiled unmodified) reproduces the oracle
    -> validates the harness
  * GPU: the PRODUCT'S drop-in executables -- rx_fm and rx_power as dropin/Makefile builds them from the reference checkout
    (sources compiled where they lie; full_demod / scanner / csv_dbm redirected to librxgpu at link time; the PATCH=1 flavour also
    with rxgpu_callback at rtl_fm.c:899), linked against the capture-replay SoapySDR stand-in -- give the same S16LE bytes / CSV rows.
    The binaries are built in this container (make -C oracle ref -> oracle/_ref/dropin_bin*/) and travel to the GPU box.
Flags are the reference's: -M wbfm -f 100M [-F 9] [-l N -L N], -f lo:hi:bin -w hamming -i 2 -1."""
import os
import subprocess
import sys

import numpy as np
import pytest

from support import ROOT, have_ref, oracle_fm_stream, sig_fm

RUNNER = os.path.join(ROOT, "tests", "dropin_runner.py")
pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")

BLOCK = 2 * 131072        # dongle_thread_fn reads MAXIMUM_BUF_LENGTH/2 complex samples per readStream (rtl_fm.c:871)


def run_ref_main(mode, iq, tmp_path, extra, pace=None):
    iq_path, out_path = str(tmp_path / "iq.npy"), str(tmp_path / ("out_%s.raw" % mode))
    np.save(iq_path, iq)
    env = dict(os.environ)
    if pace is not None:
        env["DROPIN_PACE"] = str(pace)
    p = subprocess.run([sys.executable, RUNNER, mode, iq_path, out_path] + extra, env=env, capture_output=True, timeout=300)
    assert os.path.exists(out_path), p.stderr.decode()[-2000:]
    return np.fromfile(out_path, dtype=np.int16), p.stderr.decode()


def expected(iq, **kw):
    """what rx_fm writes for this capture: every block through callback + full_demod, then -- reference
    behaviour at shutdown -- main() wakes the demod thread once more (rtl_fm.c:1480) and it runs
    full_demod again on the already-decimated lowpassed[] it still holds (rtl_fm.c:921-923)"""
    import ctypes as C
    from support import oracle, oracle_fm_state, ptr16
    O = oracle()
    st = oracle_fm_state(**kw)
    lp = np.zeros(BLOCK, np.int16)
    res = np.zeros(BLOCK, np.int16)
    out = []
    lp_len = C.c_int(0)
    for b in range(len(iq) // BLOCK):
        blk = np.ascontiguousarray(iq[b * BLOCK:(b + 1) * BLOCK])
        n = O.rxo_fm_block(C.byref(st), ptr16(blk), BLOCK, ptr16(lp), C.byref(lp_len), ptr16(res))
        out.append(res[:n].copy())
    n = O.rxo_fm_full_demod(C.byref(st), ptr16(lp), C.byref(lp_len), ptr16(res))
    out.append(res[:n].copy())
    return np.concatenate(out)


@pytest.mark.ref
@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra,kw", [
    (["-M", "wbfm", "-f", "100M"], dict(downsample=6)),            # optimal_settings: 1000000/170000 + 1 (rtl_fm.c:968)
    (["-M", "wbfm", "-o", "4", "-E", "rdc", "-f", "100M"], dict(downsample=2, post_downsample=4, dc_block_raw=1)),   # rate_in x4 -> ds 2
])
def test_reference_main_untouched_matches_oracle(tmp_path, extra, kw):
    iq = sig_fm(4 * 131072, seed=2024)
    got, err = run_ref_main("cpu", iq, tmp_path, extra)
    want = expected(iq, **kw)
    assert len(got) == len(want), err[-1500:]
    assert np.array_equal(got, want)


def dropin_bin(name, patched=False):
    return os.path.join(ROOT, "oracle", "_ref", "dropin_bin_patched" if patched else "dropin_bin", name)


def run_dropin_rx_fm(iq, tmp_path, extra, patched):
    """the product's rx_fm executable on a capture file, replayed at a quarter of real time (the reference's hand-off between its
    dongle and demod threads is a single lossy slot, rtl_fm.c:858-862/921-924: a real-time source never overruns it), ^C at the end"""
    exe = dropin_bin("rx_fm", patched)
    if not os.path.exists(exe):
        pytest.skip("dropin binaries not built (make -C oracle ref)")
    iq_path, out_path = str(tmp_path / "iq.cs16"), str(tmp_path / "out.raw")
    iq.tofile(iq_path)
    env = dict(os.environ, SOAPY_FAKE_FILE=iq_path, SOAPY_FAKE_PACE="0.25", SOAPY_FAKE_EOF_SIGINT="300")
    p = subprocess.run([exe] + extra + [out_path], env=env, capture_output=True, timeout=300)
    assert os.path.exists(out_path), p.stderr.decode()[-2000:]
    return np.fromfile(out_path, dtype=np.int16), p.stderr.decode()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("patched", [False, True])
@pytest.mark.parametrize("extra,kw", [
    (["-M", "wbfm", "-f", "100M"], dict(downsample=6)),
    (["-M", "wbfm", "-F", "9", "-f", "100M"], dict(downsample_passes=3, comp_fir_size=9)),
    (["-M", "wbfm", "-o", "4", "-E", "rdc", "-f", "100M"], dict(downsample=2, post_downsample=4, dc_block_raw=1)),
])
def test_dropin_rx_fm_executable(tmp_path, extra, kw, patched):
    iq = sig_fm(5 * 131072, seed=2025)
    got, err = run_dropin_rx_fm(iq, tmp_path, extra, patched)
    want = expected(iq, **kw)                                     # 5 blocks + the shutdown wake-up
    assert len(got) == len(want), err[-1500:]
    assert np.array_equal(got, want)


LEVEL_LINE = __import__("re").compile(r"^-?[0-9.]+(e[+-]?[0-9]+)?, -?\d+, -?\d+, -?\d+$")


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra", [["-M", "wbfm", "-f", "100M", "-L", "2"], ["-M", "fm", "-s", "170k", "-f", "100M", "-l", "60", "-L", "3"]])
def test_dropin_rx_fm_prints_the_reference_level_lines(tmp_path, extra):
    """-L (rtl_fm.c:792-807): the level lines of the drop-in executable == those of the untouched reference main(), with and without
    squelch (a block the squelch zeroes is measured before it is zeroed: the device's rms)"""
    iq = sig_fm(6 * 131072, seed=2026)
    iq[2 * 262144:3 * 262144] //= 64                               # one quiet block for the squelch
    got, err_gpu = run_dropin_rx_fm(iq, tmp_path, extra, False)
    # the checker is the untouched reference main(), whose own hand-off between its threads is lossy (rtl_fm.c:858-862): on a loaded host a
    # block can go missing from the CPU run -- paced more slowly until it has taken every block the executable took
    for pace in (0.02, 0.1, 0.4):
        want, err_cpu = run_ref_main("cpu", iq, tmp_path, extra, pace)
        if len(want) >= len(got):
            break
    assert np.array_equal(got, want)
    lv_gpu = [ln for ln in err_gpu.splitlines() if LEVEL_LINE.match(ln.strip())]
    lv_cpu = [ln for ln in err_cpu.splitlines() if LEVEL_LINE.match(ln.strip())]
    assert lv_cpu and lv_gpu == lv_cpu, (lv_gpu, lv_cpu)


# ------------------------------------------------------------------ rx_power

def power_expected_rows(data, passes, rng, window, flags):
    """CSV rows (without the two strftime columns) after `passes` scanner() passes, from the oracle"""
    import ctypes as C
    import rx_tools_amd as R
    from support import oracle, PowerCfg, ptr16, ptr32, ptr64
    O = oracle()
    plan = R.plan_range(rng, 0.0, flags[0])
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs(window, n), R.sine_table(plan.bin_e)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, flags[0], flags[1], flags[2], ptr32(wc), ptr16(sw))
    d3 = data.reshape(passes, plan.tune_count, plan.buf_len)
    work = np.zeros(plan.buf_len, np.int16)
    rows = []
    buf = C.create_string_buffer(1 << 20)
    for t in range(plan.tune_count):
        avg = np.zeros(n, np.int64)
        smp = C.c_int(0)
        for p in range(passes):
            O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(d3[p, t])), ptr16(work), ptr64(avg), C.byref(smp))
        O.rxo_csv_row(buf, len(buf), plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, plan.downsample, plan.crop,
                      ptr64(avg), C.byref(smp))
        rows.append(buf.value.decode().rstrip("\n"))
    return rows, plan


def run_power_main(mode, data, plan, tmp_path, extra):
    iq_path, out_path = str(tmp_path / "iq.npy"), str(tmp_path / ("out_%s.csv" % mode))
    # every tune is retuned to, and retune() flush-reads one chunk before the data read
    # (rtl_power.c:560-576): interleave a dummy chunk before each tune buffer
    chunks = data.reshape(-1, plan.buf_len)
    laid = np.zeros((chunks.shape[0], 2, plan.buf_len), np.int16)
    laid[:, 1, :] = chunks
    np.save(iq_path, laid.ravel())
    p = subprocess.run([sys.executable, RUNNER, mode, iq_path, out_path, "--buf-len", str(plan.buf_len)] + extra,
                       capture_output=True, timeout=300)
    assert os.path.exists(out_path), p.stderr.decode()[-2000:]
    # strip "YYYY-MM-DD, HH:MM:SS, " (rtl_power.c:1046-1048): timing-dependent, masked
    return [line.split(", ", 2)[2] for line in open(out_path).read().splitlines()], p.stderr.decode()


# -i 2: the reference ticks on whole seconds of time(NULL) (rtl_power.c:1029,1041-1043); 2 s leaves >= 1 s for the three passes
POWER_ARGS = ("88M:108M:125k", "hamming", (1, 0, 0), ["-f", "88M:108M:125k", "-w", "hamming", "-i", "2", "-1"])


@pytest.mark.ref
@pytest.mark.timeout(600)
def test_reference_rx_power_main_untouched_matches_oracle(tmp_path):
    from support import sig_noise
    rng, window, flags, args = POWER_ARGS
    passes = 3
    want, plan = power_expected_rows(sig_noise(passes * 8 * 16384, seed=606, amp=3000), passes, rng, window, flags)
    got, err = run_power_main("power-cpu", sig_noise(passes * 8 * 16384, seed=606, amp=3000), plan, tmp_path, args)
    assert got == want, err[-1500:]


def run_dropin_rx_power(data, plan, tmp_path, extra):
    exe = dropin_bin("rx_power")
    if not os.path.exists(exe):
        pytest.skip("dropin binaries not built (make -C oracle ref)")
    iq_path, out_path = str(tmp_path / "iq.cs16"), str(tmp_path / "out.csv")
    np.ascontiguousarray(data).tofile(iq_path)
    # scanner() asks for buf_len ELEMENTS but consumes buf_len int16 (rtl_power.c:526/715): the device hands out buf_len/2 per read;
    # retune()'s flush reads go to the file-static `dump`, which the unit tells the replay device to leave the capture alone for
    env = dict(os.environ, SOAPY_FAKE_FILE=iq_path, SOAPY_FAKE_MAX_READ=str(plan.buf_len // 2))
    p = subprocess.run([exe] + extra + [out_path], env=env, capture_output=True, timeout=300)
    assert os.path.exists(out_path), p.stderr.decode()[-2000:]
    return [line.split(", ", 2)[2] for line in open(out_path).read().splitlines()], p.stderr.decode()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_dropin_rx_power_executable(tmp_path):
    from support import sig_noise
    rng, window, flags, args = POWER_ARGS
    passes = 3
    data = sig_noise(passes * 8 * 16384, seed=607, amp=3000)
    want, plan = power_expected_rows(data, passes, rng, window, flags)
    got, err = run_dropin_rx_power(data, plan, tmp_path, args)
    assert got == want, err[-1500:]


# BASELINE configs[2]: 599 tunes x 16384 int16 per pass, N = 4096 (rtl_power.c:431-543 plans it; 1039-1050 prints the rows in tune order)
SWEEP_ARGS = ("24M:1.7G:1k", "rectangle", (1, 0, 0), ["-f", "24M:1.7G:1k", "-i", "2", "-1"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("deferred", [None, "1", "0"])
def test_dropin_rx_power_executable_at_the_baseline_sweep(tmp_path, deferred, monkeypatch):
    """`rx_power -f 24M:1.7G:1k -i 2 -1` through the product's executable (the reference's own main(), retune/flush reads, scanner() call and
    csv_dbm loop around librxgpu) == the CSV text of the UNTOUCHED reference main() (libref_power.so) on the same capture, all 599 rows,
    timestamp columns masked -- and == the oracle's rows (one pass: 2 transforms of 4096 per tune); with the sums kept on the device between sweeps ($RXGPU_SCAN_DEFERRED=1), merged per
    sweep (=0), and the default"""
    from support import sig_noise
    if deferred is not None:
        monkeypatch.setenv("RXGPU_SCAN_DEFERRED", deferred)
    rng, window, flags, args = SWEEP_ARGS
    # ONE sweep per report at this geometry, by the reference's own clock: retune() sleeps 5 ms per hop (rtl_power.c:563), 599 hops take 3 s,
    # the 2 s tick has passed when the first scanner() returns (rtl_power.c:1040-1043)
    passes = 1
    data = sig_noise(passes * 599 * 16384, seed=608, amp=2500)
    want, plan = power_expected_rows(data, passes, rng, window, flags)
    assert plan.tune_count == 599 and plan.buf_len == 16384 and plan.bin_e == 12
    got, err = run_dropin_rx_power(data, plan, tmp_path, args)
    assert len(got) == 599, err[-1500:]
    assert got == want, err[-1500:]
    ref_rows, err_ref = run_power_main("power-cpu", data, plan, tmp_path, list(args))
    assert ref_rows == got, err_ref[-1500:]
