"""Checks of the GPU path against the CPU reference AT BENCH SIZE -- checker infrastructure for bench.py.

Every configuration bench.py times is also run once more, in the very launch shape that was timed, and compared with
the reference over the whole input: rx_fm chains sample by sample (output + every carry), rx_power avg[]/samples of
every tune, the channeliser's every window.  The reference is single-threaded and ~1000x slower than the device, so
the work is dealt to forked children of this process (one per hardware thread, capped): a child inherits the parent's
host copies of the input and of the GPU result copy-on-write, and its own private copy of the reference's globals
(oracle/_ref keeps all state in file-scope globals, so children cannot disturb each other), compares, and sends a
small verdict back through a pipe.

TEST INFRASTRUCTURE: loads oracle/ (allowed for tests/ and bench.py's checker legs only).  Reference legs:
rtl_fm.c:828-863 + 759-824 (callback + full_demod), rtl_power.c:670-772 (scanner), and for the channeliser (not a
reference feature) the oracle's restatement built from reference-pinned primitives.
"""
import ctypes as C
import os
import pickle
import select
import signal
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import support  # noqa: E402


def n_workers(limit=128):
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(limit, n))


def fork_map(fn, jobs, timeout_s=600.0):
    """fn(job) in one forked child per job, all at once; returns the list of results in job order.  A child that raises
    sends the exception text back; a child that does not answer within timeout_s is killed (by its own PID) and the
    call raises.  Children never touch HIP: they only run numpy / ctypes code on host memory inherited from the parent."""
    kids = []
    sys.stdout.flush()
    sys.stderr.flush()
    for job in jobs:
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 0
            try:
                os.close(r)
                try:
                    payload = pickle.dumps(("ok", fn(job)), protocol=pickle.HIGHEST_PROTOCOL)
                except BaseException as e:                       # noqa: BLE001 -- the parent decides what to do with it
                    payload = pickle.dumps(("error", repr(e)))
                    code = 1
                view = memoryview(payload)
                while view:
                    n = os.write(w, view[:1 << 20])
                    view = view[n:]
                os.close(w)
            finally:
                os._exit(code)                                   # no atexit handlers, no torch teardown in the child
        os.close(w)
        kids.append((pid, r))
    deadline = time.monotonic() + timeout_s
    bufs = {r: bytearray() for _, r in kids}
    open_fds = set(bufs)
    try:
        while open_fds:
            left = deadline - time.monotonic()
            if left <= 0:
                raise TimeoutError("parity workers did not finish within %.0f s" % timeout_s)
            ready, _, _ = select.select(list(open_fds), [], [], min(left, 5.0))
            for fd in ready:
                chunk = os.read(fd, 1 << 20)
                if chunk:
                    bufs[fd] += chunk
                else:
                    open_fds.discard(fd)
    except BaseException:
        for pid, _ in kids:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        raise
    finally:
        for pid, r in kids:
            try:
                os.waitpid(pid, 0)
            except OSError:
                pass
            os.close(r)
    out = []
    for _, r in kids:
        if not bufs[r]:
            raise RuntimeError("a parity worker died without an answer")
        kind, val = pickle.loads(bytes(bufs[r]))
        if kind != "ok":
            raise RuntimeError("parity worker: " + val)
        out.append(val)
    return out


def split_range(n, parts):
    """[0, n) in `parts` contiguous pieces (the first ones one longer), empty pieces dropped"""
    parts = max(1, min(parts, n))
    base, extra = divmod(n, parts)
    out, lo = [], 0
    for i in range(parts):
        hi = lo + base + (1 if i < extra else 0)
        if hi > lo:
            out.append((lo, hi))
        lo = hi
    return out


# ------------------------------------------------------------------------------------------------ rx_fm

def fm_decimation(params_kw):
    if params_kw.get("downsample_passes"):
        return 1 << params_kw["downsample_passes"]
    return params_kw.get("downsample", 6)


def _carry_tuple(c, params_kw):
    """the carries the chain of this parameter set touches, from an FmCarry or a DemodState (same field names)"""
    t = [c.pre_r, c.pre_j, c.now_lpr, c.prev_lpr_index]
    passes = params_kw.get("downsample_passes", 0)
    if passes:
        for i in range(passes):
            t += list(c.lp_i_hist[i]) + list(c.lp_q_hist[i])
        if params_kw.get("comp_fir_size") == 9:
            t += list(c.droop_i_hist) + list(c.droop_q_hist)
    else:
        t += [c.now_r, c.now_j, c.prev_index]
    return tuple(int(v) for v in t)


def fm_gpu_sequence(torch, R, d_iq, n_blocks, block_len, tail_blocks, params_kw):
    """[all n_blocks][the first tail_blocks again], two pipelined runs with the carries chained on the device -- the way the
    timed loop chains its steps.  Returns the host copy of the output, the carries, the fix-up count."""
    T = n_blocks * (block_len // 2)
    ds = fm_decimation(params_kw)
    d_out = torch.zeros((T + tail_blocks * (block_len // 2)) // ds + 4096, dtype=torch.int16, device=d_iq.device)
    s = R.FmStream(R.FmParams.wbfm(**params_kw), n_blocks, block_len)
    t0 = time.perf_counter()
    n1, _ = s.run_async(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), d_out.numel())
    n2, _ = (0, None)
    if tail_blocks:
        n2, _ = s.run_async(d_iq.data_ptr(), tail_blocks, block_len, d_out.data_ptr() + 2 * n1, d_out.numel() - n1)
    s.wait()
    gpu_s = time.perf_counter() - t0
    got = d_out[:n1 + n2].cpu().numpy()
    carry = _carry_tuple(s.get_carry(), params_kw)
    deemph_avg = int(s.get_carry().deemph_avg)
    fix = int(s.host_fixups)
    s.close()
    del d_out
    return {"got": got, "carry": carry, "deemph_avg": deemph_avg, "fixups": fix, "gpu_s": gpu_s, "calls": n_blocks + tail_blocks}


def fm_cpu_verdict(h_iq, n_blocks, block_len, seq, params_kw):
    """The reference (oracle/_ref: its own rtlsdr_callback + full_demod; the oracle port where the prebuilt object is absent)
    over the same calls; output and carries against `seq`.  Runs in a forked child (or in-process)."""
    from rx_tools_amd.structs import DemodState
    calls = seq["calls"]
    got = seq["got"]
    want = np.zeros(got.size + 4096, np.int16)
    t0 = time.perf_counter()
    if support.have_ref():
        F = support.ref_fm()
        support.ref_fm_reset(F, **params_kw)
        scratch = np.zeros(block_len, np.int16)
        produced = F.ref_fm_run_blocks(support.ptr16(h_iq), n_blocks, block_len, calls, support.ptr16(scratch), support.ptr16(want), want.size)
        d = DemodState.from_address(F.ref_fm_demod())
        ref_carry = _carry_tuple(d, params_kw)
        kind = "reference"
    else:
        O = support.oracle()
        st = support.oracle_fm_state(**params_kw)
        produced = O.rxo_fm_stream(C.byref(st), support.ptr16(h_iq), n_blocks, block_len, support.ptr16(want), None)
        if calls > n_blocks:
            produced += O.rxo_fm_stream(C.byref(st), support.ptr16(h_iq), calls - n_blocks, block_len, support.ptr16(want[produced:]), None)
        ref_carry = _carry_tuple(st, params_kw)
        kind = "port"
    cpu_s = time.perf_counter() - t0
    ok = produced == got.size and np.array_equal(got, want[:produced]) and seq["carry"] == ref_carry
    res = {"parity_ok": bool(ok), "parity_checked_samples": calls * (block_len // 2), "parity_outputs_compared": int(produced),
           "parity_checker": kind, "parity_cpu_seconds": cpu_s, "parity_gpu_seconds": seq["gpu_s"], "parity_host_fixups": seq["fixups"]}
    if not ok:
        m = min(got.size, int(produced))
        bad = np.nonzero(got[:m] != want[:m])[0]
        res["parity_first_mismatch"] = int(bad[0]) if bad.size else -1
        res["parity_carries"] = {"gpu": seq["carry"], "cpu": ref_carry, "gpu_len": int(got.size), "cpu_len": int(produced)}
    return res


def fm_check_many(h_iq, n_blocks, block_len, legs):
    """legs: {label: (params_kw, seq)}; one forked child per leg (each has its own copy of the reference's globals).
    Falls back to one leg after the other in this process if forking fails."""
    labels = list(legs)
    if support.have_ref():
        support.ref_fm()                                        # load before forking: children only call into it
    else:
        support.oracle()

    def one(label):
        kw, seq = legs[label]
        return fm_cpu_verdict(h_iq, n_blocks, block_len, seq, kw)
    t0 = time.perf_counter()
    try:
        results = fork_map(one, labels)
        how = "%d forked checkers, one per chain" % len(labels)
    except (OSError, RuntimeError, TimeoutError) as e:
        results = [one(lb) for lb in labels]
        how = "in-process, one chain after the other (fork failed: %r)" % (e,)
    wall = time.perf_counter() - t0
    out = dict(zip(labels, results))
    for r in out.values():
        r["parity_wall_seconds_all_legs"] = wall
        r["parity_how"] = how
    return out


# ------------------------------------------------------------------------------------------------ rx_power

def power_check(range_arg, crop, window, boxcar, comp_fir, peak_hold, h_in, got_avg, got_smp):
    """scanner() (rtl_power.c:670-772) over h_in [passes][tunes][buf_len] against the device's avg [tunes][N] / samples [tunes].
    Tunes are independent and avg[] is a sum over passes (a maximum with peak hold), so children take a contiguous range of
    tunes -- or, for a sweep of few tunes, a range of passes -- run the reference's own scanner() on their own copy of its
    tunes[] and return their rows; the parent adds the pass-partials (max for peak hold) and compares everything."""
    from rx_tools_amd.structs import TuningState
    passes, tunes, buf_len = h_in.shape
    n_bins = got_avg.shape[1]
    W = n_workers()
    if tunes >= W:
        jobs = [(lo, hi, 0, passes) for lo, hi in split_range(tunes, W)]
    else:
        per_tune = max(1, W // tunes)
        jobs = [(t, t + 1, lo, hi) for t in range(tunes) for lo, hi in split_range(passes, per_tune)]
    t0 = time.perf_counter()
    if support.have_ref():
        P = support.ref_power()
        P.ref_power_set_flags(boxcar, comp_fir, peak_hold)
        P.ref_power_scan_tuned.argtypes = [support.i16p, C.c_int]
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)                                      # frequency_range prints its plan to stderr
        try:
            tc = P.ref_power_setup(range_arg.encode(), crop, window.encode())
        finally:
            os.dup2(saved, 2)
            os.close(devnull)
            os.close(saved)
        assert tc == tunes, (tc, tunes)
        kind = "reference"

        def one(job):
            t_lo, t_hi, p_lo, p_hi = job
            k = t_hi - t_lo
            arr = (TuningState * tc).from_address(P.ref_power_tunes())
            if t_lo:
                C.memmove(C.addressof(arr), C.addressof(arr) + t_lo * C.sizeof(TuningState), k * C.sizeof(TuningState))
            C.c_int.in_dll(P, "tune_count").value = k
            for i in range(k):
                C.memset(arr[i].avg, 0, 8 * n_bins)
                arr[i].samples = 0
            sub = np.ascontiguousarray(h_in[p_lo:p_hi, t_lo:t_hi, :])
            P.ref_power_scan_tuned(support.ptr16(sub), p_hi - p_lo)
            avg = np.stack([np.ctypeslib.as_array(arr[i].avg, shape=(n_bins,)).copy() for i in range(k)])
            smp = np.array([arr[i].samples for i in range(k)], np.int64)
            return avg, smp
    else:
        import rx_tools_amd as R
        O = support.oracle()
        bin_e = n_bins.bit_length() - 1
        plan = R.plan_range(range_arg, crop, boxcar)
        wc, sw = R.window_coefs(window, n_bins), R.sine_table(bin_e)
        cfg = support.PowerCfg(bin_e, buf_len, plan.downsample, plan.downsample_passes, boxcar, comp_fir, peak_hold,
                               support.ptr32(wc), support.ptr16(sw))
        kind = "port"

        def one(job):
            t_lo, t_hi, p_lo, p_hi = job
            avg = np.zeros((t_hi - t_lo, n_bins), np.int64)
            smp = np.zeros(t_hi - t_lo, np.int64)
            work = np.zeros(buf_len, np.int16)
            for p in range(p_lo, p_hi):
                for t in range(t_lo, t_hi):
                    s = C.c_int(int(smp[t - t_lo]))
                    O.rxo_power_tune(C.byref(cfg), support.ptr16(np.ascontiguousarray(h_in[p, t])), support.ptr16(work),
                                     support.ptr64(avg[t - t_lo]), C.byref(s))
                    smp[t - t_lo] = s.value
            return avg, smp
    try:
        parts = fork_map(one, jobs)
        how = "%d forked checkers" % len(jobs)
    except (OSError, RuntimeError, TimeoutError) as e:
        # a bounded in-process sample instead: first, middle and last tune
        keep = sorted({0, tunes // 2, tunes - 1})
        jobs = [(t, t + 1, 0, passes) for t in keep]
        parts = [fork_map(one, [j])[0] for j in jobs]           # still forked, one at a time: `one` edits the reference's globals
        how = "first/middle/last tune only (parallel fork failed: %r)" % (e,)
    want_avg = np.zeros((tunes, n_bins), np.int64)
    want_smp = np.zeros(tunes, np.int64)
    seen = np.zeros(tunes, bool)
    for (t_lo, t_hi, _, _), (avg, smp) in zip(jobs, parts):
        if peak_hold:
            want_avg[t_lo:t_hi] = np.maximum(want_avg[t_lo:t_hi], avg)
        else:
            want_avg[t_lo:t_hi] += avg
        want_smp[t_lo:t_hi] += smp
        seen[t_lo:t_hi] = True
    rows = np.nonzero(seen)[0]
    ok_avg = np.array_equal(got_avg[rows], want_avg[rows])
    ok_smp = np.array_equal(np.asarray(got_smp, np.int64)[rows], want_smp[rows])
    res = {"parity_ok": bool(ok_avg and ok_smp), "parity_checker": kind, "parity_how": how,
           "parity_tunes_compared": int(rows.size), "parity_passes": int(passes),
           "parity_bins_compared": int(rows.size) * n_bins, "parity_input_samples": int(rows.size) * passes * (buf_len // 2),
           "parity_seconds": time.perf_counter() - t0}
    if not res["parity_ok"]:
        bad = np.argwhere(got_avg[rows] != want_avg[rows])
        res["parity_first_mismatch"] = [int(v) for v in bad[0]] if bad.size else None
        res["parity_samples_equal"] = bool(ok_smp)
    return res


# ------------------------------------------------------------------------------------------------ channeliser

def chan_check(h_iq, n_blocks, block_len, bin_e, first_bin, n_ch, custom_atan, sinewave, got, got_pre):
    """Every window of every channel over the whole capture.  Where oracle/_ref exists (this container, and the GPU box: the
    prebuilt objects travel) the checker is REFERENCE-BUILT code only -- the reference's own fix_fft per window (libref_power.so)
    and the reference's own full_demod per channel and callback block (libref_fm.so), support.ref_chan_stream -- else the oracle's
    restatement rxo_chan_block, which tests/test_chan_oracle.py pins against exactly that chain.  Block ranges are dealt to
    forked children; a child warms its per-channel pre_r/pre_j up on the block in front of its range (a block's carry-out
    depends on its last window only).  got: [n_ch][windows] int16 from a run that started with zero carries; got_pre: the
    carries it left."""
    n = 1 << bin_e
    wpb = block_len // 2 // n
    jobs = split_range(n_blocks, n_workers())
    use_ref = support.have_ref()
    if use_ref:
        support.ref_power()
        support.ref_fm()
        kind = "reference (oracle/_ref: the reference's own fix_fft per window + its own full_demod per channel and callback block)"
        stream = support.ref_chan_stream
    else:
        support.oracle()
        kind = "port (rxo_chan_block: fix_fft + fm_demod restatements; pinned against the reference-built chain in tests/test_chan_oracle.py)"
        stream = support.oracle_chan_stream_compare

    def one(job):
        lo, hi = job
        pre = None
        if lo:
            _, pre, _ = stream(h_iq[(lo - 1) * block_len:lo * block_len], block_len, bin_e, first_bin, n_ch, custom_atan,
                               compare=got[:, (lo - 1) * wpb:lo * wpb])
        bad, pre, _ = stream(h_iq[lo * block_len:hi * block_len], block_len, bin_e, first_bin, n_ch, custom_atan, pre=pre,
                             compare=got[:, lo * wpb:hi * wpb])
        return (lo + bad if bad >= 0 else -1), (pre if hi == n_blocks else None)
    t0 = time.perf_counter()
    parts = fork_map(one, jobs)
    bad = [p[0] for p in parts if p[0] >= 0]
    pre_end = parts[-1][1]
    ok = not bad and np.array_equal(pre_end, np.asarray(got_pre, np.int32))
    res = {"parity_ok": bool(ok), "parity_checker": kind,
           "parity_windows_compared": int(n_blocks * wpb), "parity_channels": int(n_ch), "parity_checked_samples": int(n_blocks * (block_len // 2)),
           "parity_how": "%d forked checkers" % len(jobs), "parity_seconds": time.perf_counter() - t0}
    if bad:
        res["parity_first_bad_block"] = int(min(bad))
    return res
