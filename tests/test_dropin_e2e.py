"""End to end through the reference's OWN main(), threads and SoapySDR read loop (libref_fm.so =
rtl_fm.c compiled unmodified; the prebuilt object travels to the GPU box):
  * CPU: the untouched reference binary path reproduces the oracle -> validates the harness
  * GPU: with full_demod interposed by rxgpu_full_demod (oracle/dropin_interpose.c, i.e. the
    INTEGRATION.md patch applied by the dynamic linker) the same run gives the same S16LE bytes.
Flags are the reference's: -M wbfm -f 100M [-F 9]."""
import os
import subprocess
import sys

import numpy as np
import pytest

from support import ROOT, have_ref, oracle_fm_stream, sig_fm

RUNNER = os.path.join(ROOT, "tests", "dropin_runner.py")
pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")

BLOCK = 2 * 131072        # dongle_thread_fn reads MAXIMUM_BUF_LENGTH/2 complex samples per readStream (rtl_fm.c:871)


def run_ref_main(mode, iq, tmp_path, extra):
    iq_path, out_path = str(tmp_path / "iq.npy"), str(tmp_path / ("out_%s.raw" % mode))
    np.save(iq_path, iq)
    env = dict(os.environ)
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


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra,kw", [
    (["-M", "wbfm", "-f", "100M"], dict(downsample=6)),
    (["-M", "wbfm", "-F", "9", "-f", "100M"], dict(downsample_passes=3, comp_fir_size=9)),
    (["-M", "wbfm", "-o", "4", "-E", "rdc", "-f", "100M"], dict(downsample=2, post_downsample=4, dc_block_raw=1)),
])
def test_reference_main_with_rxgpu_full_demod(tmp_path, extra, kw):
    iq = sig_fm(5 * 131072, seed=2025)
    got, err = run_ref_main("gpu", iq, tmp_path, extra)
    want = expected(iq, **kw)
    assert "dropin full_demod calls: 6" in err, err[-1500:]      # 5 blocks + the shutdown wake-up
    assert len(got) == len(want), err[-1500:]
    assert np.array_equal(got, want)


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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_reference_rx_power_main_with_rxgpu_scan(tmp_path):
    from support import sig_noise
    rng, window, flags, args = POWER_ARGS
    passes = 3
    data = sig_noise(passes * 8 * 16384, seed=607, amp=3000)
    want, plan = power_expected_rows(data, passes, rng, window, flags)
    got, err = run_power_main("power-gpu", data, plan, tmp_path, args)
    assert "dropin scanner passes: 3" in err, err[-1500:]
    assert got == want
