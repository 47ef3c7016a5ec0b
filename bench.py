#!/usr/bin/env python3
"""bench.py -- the rx_tools hot path on MI355X, measured the way BASELINE.json asks.

Headline (`value`): complex IQ MSample/s through the rx_fm callback pre-stage + full_demod()
chain at the "20 Msps" WBFM geometry of BASELINE configs[1] (downsample=118 -> 170 ksps ->
32 ksps audio, -A fast, de-emphasis on), blocks of 131072 complex samples, input resident in
HBM.  One step = one rxgpu_fm_stream_run over --blocks blocks (default 16384 = 2^31 samples = 8 GiB of cs16).
The capture is generated on the device, seeded and NON-REPEATING (signal (A): FM carrier at -fs/4, 1 kHz tone,
75 kHz deviation, +-128 LSB of noise on every sample), and after the timed loop the same 2^31 samples -- plus a
chained second run, the way the timed loop chains them -- go through the CPU reference (oracle/_ref, the reference's
own rtlsdr_callback + full_demod) and every output sample and carry of the pipelined GPU sequence is compared:
`parity_checked_samples` in the JSON line.

The same line carries:
  rx_fm_variants   the small-decimation chains (-M wbfm default ds=6; BASELINE configs[0] ds=5 / 240 kHz) and the -F cascade;
                   the ds=6 chain (other kernels at size) is also compared with the CPU reference, over a quarter of the capture
  host_fed         rxgpu_fm_stream_run_host from pinned host memory (PCIe-inclusive; never `value`) and the drop-in's
                   per-block latency (rxgpu_callback + rxgpu_full_demod on a struct demod_state)
  rx_power         FFT bins/s of the scanner() chain at the configs[2] geometry (-f 24M:1.7G:1k: 599 tunes x 16384 int16,
                   N=4096), tunes sharded across the ranks with ONE ncclGather per step issued by librxgpu itself
                   (rxgpu_power_scan_run_sharded), plus two more geometries (full-scale/hamming; N=16384 with -F 9)
  channeliser, sdr_convert   the extension and the rx_sdr converters

  python bench.py --gpus 1 --steps 100 --warmup 10      (the defaults)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

rx_fm does not shard (one stream, sequential carries): N>1 runs N independent replicas
("replicas only", weak scaling).  rx_power shards by tune (strong scaling of one sweep).
torch is used for device buffers, the synthetic capture and torch.distributed's rendezvous only; every sample is
processed by librxgpu.so through its C ABI.  oracle/ is used as the checker and as the timed CPU baseline, nothing else.
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PCIE_PEAK_GBS = 63.0           # MI355X_MICROARCH.md: PCIe Gen5 x16
# integer VALU issue, nominal: one wave64 instruction per quad-cycle per SIMD: 256 CUs x 4 SIMDs x 2.4 GHz / 4.  The figure bench.py
# reports is MEASURED where the tables exist: tools/valu_issue.hip (profiles/rNN_valu_issue.json) gives wave64 instructions per SIMD-cycle
# for every opcode class (v_mad_i32_i16, the packed 16-bit ops, v_bfi, v_dot2, conversions, DPP: 0.24; v_add/sub/and/ashr, v_fma_f32: 0.40-0.46
# at the reported clock), tools/kernel_mix.py the kernel's opcode mix; valu_ceiling() weights one with the other.
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4.0
SIMDS, CLOCK_GHZ = 256 * 4, 2.4


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the newest round that has it (a round that skipped a profile section keeps the last one measured), or None"""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else None


def valu_ceiling(kernel_substr):
    """(G wave-instr/s the kernel's opcode mix can issue at most, source text) from the measured per-opcode rates; (nominal, None) without them"""
    issue = mix = None
    f_issue, f_mix = newest_profile("valu_issue.json"), newest_profile("kernel_mix.json")
    try:
        issue = json.load(open(f_issue))["ops"]
        mix = json.load(open(f_mix))
        ops = next(v for k, v in mix.items() if kernel_substr in k)
    except (OSError, ValueError, KeyError, StopIteration, TypeError):
        issue = mix = None
    if issue is None:
        return VALU_PEAK_GINSTR, None
    rnd, mrnd = os.path.basename(f_issue)[:3], os.path.basename(f_mix)[:3]
    alias = {"v_pk_mad_u16": "pk_mad_i16", "v_dot2c_i32_i16": "dot2_i32_i16", "v_fma_f32": "pk_fma_f32x", "v_mov_b32_dpp": "mov_dpp_wshr", "v_or_b32": "and_b32",
             "v_xor_b32": "and_b32", "v_mov_b32": "and_b32", "v_lshrrev_b32": "lshlrev_b32", "v_add_co_u32": "add_u32", "v_addc_co_u32": "add_u32"}
    slow = min(v["wall_w8"] for k, v in issue.items() if k in ("mad_i32_i16", "pk_add_u16", "bfi_b32"))     # the half-rate class
    cycles = total = 0.0
    for op, n in ops.items():
        key = alias.get(op, op[2:])
        rate = issue.get(key, {}).get("wall_w8", slow)
        cycles += n / rate
        total += n
    rate = total / cycles
    return rate * SIMDS * CLOCK_GHZ, ("profiles/%s_valu_issue.json (tools/valu_issue.hip: measured wave64 instructions per SIMD-cycle per opcode, 8 waves/SIMD) weighted with "
                                      "profiles/%s_kernel_mix.json (the kernel's opcode counts): %.3f per SIMD-cycle x %d SIMDs x %.1f GHz" % (rnd, mrnd, rate, SIMDS, CLOCK_GHZ))


def chan_mode_valu_frac(mode, samples, seconds):
    """Fraction of the chip's VALU issue time a channeliser mode's kernels keep busy: SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD's VALU executes
    an instruction, every pass of a multi-pass fp64 op counted) summed over the mode's kernels in the newest profiles/rNN_pmc_chan_modes.json, scaled
    to this run's samples, over SIMDs x nominal clock x the live time.  (fraction, source) or (None, None)."""
    f = newest_profile("pmc_chan_modes.json")
    try:
        m = json.load(open(f))[mode]
        if mode == "nco":
            # k_ch_nco is straight-line adds, ands and shifts, half of which issue twice per quad-cycle (0.29 instructions per SIMD-cycle measured): its
            # bound is the issue ceiling of its own opcode mix, like the transform kernels' (valu_ceiling), not one instruction per quad-cycle
            peak, src = valu_ceiling("k_ch_nco")
            rate = m["valu_wave_instr_per_run"] * (samples / float(m["samples_per_run"])) / seconds / 1e9
            return rate / peak, "%s (SQ_INSTS_VALU) / live time against %s" % (os.path.relpath(f, ROOT), src or "the nominal ceiling")
        busy_s = m["valu_active_quad_cycles_per_run"] * 4.0 * (samples / float(m["samples_per_run"])) / (SIMDS * CLOCK_GHZ * 1e9)
        return busy_s / seconds, ("%s (SQ_ACTIVE_INST_VALU of %s) x 4 cycles / (%d SIMDs x %.1f GHz x live time): the measured VALU busy share, fp64 passes included"
                                  % (os.path.relpath(f, ROOT), ", ".join(sorted(k.split("<")[0] for k, v in m["kernels"].items() if v.get("valu_active_quad_cycles_per_step", 0) > 1e5)),
                                     SIMDS, CLOCK_GHZ))
    except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError):
        return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed_reps(fn, budget_s, reps=3):
    """fn(seconds) -> (units, seconds); `reps` repetitions sharing the budget: best and median rate"""
    rates = []
    for _ in range(reps):
        units, dt = fn(budget_s / reps)
        rates.append(units / dt)
    rates.sort()
    return rates[-1], rates[len(rates) // 2], rates


def cpu_baseline_fm(block_len, budget_s, lib_name="libref_fm.so", ds=118):
    """The reference's own callback + full_demod (oracle/_ref, gcc -O2) -- or, where that prebuilt object is absent, the
    oracle port -- on one host core, config-2 parameters; 3 repetitions, best and median."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    import rx_tools_amd as R
    n_buf = 16
    iq = R.synth.sig_fm(n_buf * block_len // 2, seed=12345)
    path = os.path.join(support.ORACLE_DIR, "_ref", lib_name)
    if os.path.exists(path):
        if lib_name == "libref_fm.so":
            L = support.ref_fm()
        else:
            L = C.CDLL(path)
            L.ref_fm_demod.restype = C.c_void_p
            L.ref_fm_dongle.restype = C.c_void_p
            L.ref_fm_fn.restype = C.c_void_p
            L.ref_fm_run_blocks.restype = C.c_long
            L.ref_fm_run_blocks.argtypes = [support.i16p, C.c_size_t, C.c_size_t, C.c_size_t, support.i16p, support.i16p, C.c_size_t]
            L.ref_fm_init()
        support.ref_fm_reset(L, downsample=ds)
        scratch = np.zeros(block_len, np.int16)

        def rep(seconds):
            calls, t0 = 0, time.perf_counter()
            while True:
                L.ref_fm_run_blocks(support.ptr16(iq), n_buf, block_len, 64, support.ptr16(scratch), None, 0)
                calls += 64
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    return calls * (block_len // 2), dt
        kind = "reference"
    else:
        O = support.oracle()
        st = support.oracle_fm_state(downsample=ds)
        out = np.zeros(n_buf * block_len // 2, np.int16)

        def rep(seconds):
            calls, t0 = 0, time.perf_counter()
            while True:
                O.rxo_fm_stream(C.byref(st), support.ptr16(iq), n_buf, block_len, support.ptr16(out), None)
                calls += n_buf
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    return calls * (block_len // 2), dt
        kind = "port"
    rep(0.2)                                            # pre-fault, warm caches
    best, median, rates = _timed_reps(rep, budget_s)
    return {"value": best / 1e6, "median": median / 1e6, "unit": "MSample/s", "cores": 1, "kind": kind,
            "sample_short": "3 x %.1f s, blocks of %d complex samples, rtlsdr_callback+full_demod ds=%d, 1 thread, %s" % (budget_s / 3, block_len // 2, ds, cpu_model()),
            "sample": "3 repetitions of %.1f s over blocks of %d complex samples, rtlsdr_callback+full_demod, ds=%d wbfm, 1 thread, %s (%s)"
                      % (budget_s / 3, block_len // 2, ds, "gcc -O0 (the reference's default build)" if "O0" in lib_name else "gcc -O2", cpu_model())}


def cpu_baseline_power(plan, budget_s, lib_name="libref_power.so"):
    """scanner()'s per-tune chain on one host core: the reference's own scanner() over all 599 tunes
    (oracle/_ref) where that prebuilt object exists, else the oracle port; 3 repetitions, best and median."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    import rx_tools_amd as R
    n = 1 << plan.bin_e
    path = os.path.join(support.ORACLE_DIR, "_ref", lib_name)
    if os.path.exists(path):
        P = support.ref_power() if lib_name == "libref_power.so" else C.CDLL(path)
        P.ref_power_setup.argtypes = [C.c_char_p, C.c_double, C.c_char_p]
        P.ref_power_set_flags(1, 0, 0)
        P.ref_power_scan_tuned.argtypes = [support.i16p, C.c_int]
        tunes = P.ref_power_setup(b"24M:1.7G:1k", 0.0, b"rectangle")
        data = R.synth.sig_noise(tunes * plan.buf_len, seed=777, amp=100)

        def rep(seconds):
            done, t0 = 0, time.perf_counter()
            while True:
                P.ref_power_scan_tuned(support.ptr16(data), 1)     # scanner() without retune()'s settle sleep
                done += tunes
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    return done * (plan.buf_len // 2), dt
        kind = "reference"
    else:
        tunes = 64
        data = R.synth.sig_noise(tunes * plan.buf_len, seed=777, amp=100)
        O = support.oracle()
        wc, sw = R.window_coefs("rectangle", n), R.sine_table(plan.bin_e)
        cfg = support.PowerCfg(plan.bin_e, plan.buf_len, 1, 0, 1, 0, 0, support.ptr32(wc), support.ptr16(sw))
        avg = np.zeros(n, np.int64)
        work = np.zeros(plan.buf_len, np.int16)
        smp = C.c_int(0)

        def rep(seconds):
            done, t0 = 0, time.perf_counter()
            while True:
                for t in range(tunes):
                    O.rxo_power_tune(C.byref(cfg), support.ptr16(data[t * plan.buf_len:(t + 1) * plan.buf_len]),
                                     support.ptr16(work), support.ptr64(avg), C.byref(smp))
                done += tunes
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    return done * (plan.buf_len // 2), dt
        kind = "port"
    best, median, rates = _timed_reps(rep, budget_s)
    return {"value": best / 1e6, "median": median / 1e6, "unit": "Mbins/s", "cores": 1, "kind": kind,
            "sample_short": "3 x %.1f s, scanner() over tune buffers of %d int16 (N=%d), 1 thread, %s" % (budget_s / 3, plan.buf_len, n, cpu_model()),
            "sample": "3 repetitions of %.1f s over tune buffers of %d int16 (N=%d), scanner() per-tune chain, 1 thread, %s (%s)"
                      % (budget_s / 3, plan.buf_len, n, "gcc -O0 (the reference's default build)" if "O0" in lib_name else "gcc -O2", cpu_model())}


def cpu_subprocess(which, budget_s, lib=""):
    """one baseline in a process of its own (the -O0 objects define the same globals as the -O2 ones)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", which, "--cpu-seconds", "%g" % budget_s, "--cpu-lib", lib]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout
        return json.loads(out.decode().strip().splitlines()[-1])
    except Exception as e:                                       # noqa: BLE001 -- a missing baseline must not sink the bench
        return {"error": repr(e)}


def cpu_all_cores(which, budget_s):
    """one independent replica of the single-thread baseline per hardware thread (the reference has
    one demod thread / a single-threaded scanner, and keeps its state in globals, hence processes not threads)."""
    import subprocess
    n = os.cpu_count() or 1
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", which, "--cpu-seconds", "%g" % budget_s]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(n)]
    total, ok = 0.0, 0
    for p in procs:
        out, _ = p.communicate()
        try:
            total += json.loads(out.decode().strip().splitlines()[-1])["median"]
            ok += 1
        except (ValueError, IndexError, KeyError):
            pass
    return {"value": total, "cores": ok, "note": "%d concurrent single-thread replicas, %.0f s each (sum of medians)" % (ok, budget_s)}


def cpu_worker(which, budget_s, lib):
    if which == "fm":
        r = cpu_baseline_fm(2 * 131072, budget_s, lib or "libref_fm.so")
    else:
        import types
        r = cpu_baseline_power(types.SimpleNamespace(bin_e=12, buf_len=16384), budget_s, lib or "libref_power.so")   # -f 24M:1.7G:1k geometry
    print(json.dumps(r))


def device_capture(torch, dev, n_complex, seed, fs=20.06e6, amp=20000.0, tone=1000.0, devi=75e3, noise=128, chunk=1 << 25):
    """Signal (A) of SURVEY 8(d) on the device, seeded, any length, never repeating: FM carrier at -fs/4 (rotate16_90 brings it
    to DC), 1 kHz tone at 75 kHz deviation, plus uniform noise of +-128 LSB drawn per sample from a seeded generator.
    (rx_tools_amd.synth.sig_fm is the same signal from a host LCG; tiling a few of its blocks -- what round 1 did --
    repeats every first-sample value.)"""
    out = torch.empty(2 * n_complex, dtype=torch.int16, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    two_pi = 2.0 * np.pi
    for lo in range(0, n_complex, chunk):
        m = min(chunk, n_complex - lo)
        t = torch.arange(lo, lo + m, dtype=torch.float64, device=dev)
        # carrier phase -2*pi*t/4 taken modulo one turn before the multiply; the tone argument stays below 2^31 * 3e-4
        phase = (torch.remainder(t, 4.0) * (-0.25 * two_pi)) + (devi / tone) * torch.sin((two_pi * tone / fs) * t)
        nz = torch.randint(-noise, noise + 1, (2 * m,), dtype=torch.int32, device=dev, generator=g)
        i = torch.round(amp * torch.cos(phase)).to(torch.int32) + nz[0::2]
        q = torch.round(amp * torch.sin(phase)).to(torch.int32) + nz[1::2]
        v = out[2 * lo:2 * (lo + m)]
        v[0::2] = i.clamp_(-32768, 32767).to(torch.int16)
        v[1::2] = q.clamp_(-32768, 32767).to(torch.int16)
        del t, phase, nz, i, q
    return out


def parity_module():
    """tests/parity_at_size.py: the checks against the CPU reference at bench size (checker infrastructure, loads oracle/)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_at_size
    return parity_at_size


VARIANT_SHORT = {"-M wbfm default, downsample=6": "rx_fm ds=6 (-M wbfm default)",
                 "BASELINE configs[0] geometry: -s 240000, downsample=5, deemph_a=19": "rx_fm ds=5 (configs[0], 240 kHz)",
                 "-F cascade, downsample_passes=7 (ds=128)": "rx_fm -F ds=128",
                 "-F 9 cascade as -M wbfm -F 9 sets it: downsample_passes=3 (ds=8) + droop FIR": "rx_fm -M wbfm -F 9"}


def hoist(result, parity_all, world):
    """The driver's record keeps the top-level scalars, `config`, `roofline` and `cpu_baseline` of the line and only the NAMES of
    everything else.  So what the other legs measured is repeated, compact, inside those two objects: `config` answers "did RCCL
    see N ranks, what was the sharded rx_power rate, was the gathered result bit-exact" from a SCALE record alone, `roofline.legs`
    carries every leg's fraction of its binding roofline, its HBM traffic over its algorithmic bytes and its parity verdict."""
    cfg, roof = result.get("config"), result.get("roofline")
    pw = result.get("rx_power") if isinstance(result.get("rx_power"), dict) else (result if "Mbins" in str(result.get("unit")) else None)
    if isinstance(cfg, dict) and pw is not None and pw is not result:
        c = pw["config"]
        cfg.update({
            "rx_power_Mbins_per_s": pw["value"], "rx_power_ms_per_step": pw["ms_per_step"], "rx_power_scaling": "strong (one sweep sharded by tune)",
            "rx_power_passes_per_step": c["passes_per_step"],
            "rccl_ranks": c["rccl_ranks"], "rccl_ranks_source": c["rccl_ranks_source"], "rccl_gathers_enqueued": c["rccl_gathers_enqueued"],
            "gather": c["gather"], "gather_bytes_per_rank": c["gather_bytes_per_rank"],
            "scan_us_rank0": c["scan_us_per_step_rank0"], "gather_us_rank0": c["gather_us_per_step_rank0"],
            "tunes_per_rank": c["tunes_per_rank_padded"], "tunes_rank0": c["tunes_this_rank"],
            "rx_power_parity_ok": pw.get("parity_ok"),
            "rx_power_parity": ({k: pw["parity_sharded"].get(k) for k in ("parity_checker", "parity_tunes_compared", "parity_passes", "parity_ranks",
                                                                          "parity_padding_rows_zero", "parity_inputs_regenerated_match_owner_checksums")}
                                if "parity_sharded" in pw else {k: pw.get("parity", {}).get(k) for k in ("parity_checker", "parity_tunes_compared", "parity_passes")}),
            "rx_power_cpu_baseline_Mbins_per_s_1core": (pw.get("cpu_baseline") or {}).get("value"),
            "rx_power_cpu_baseline_kind": (pw.get("cpu_baseline") or {}).get("kind"),
            "parity_all_legs": dict(parity_all), "n_ranks": world,
            # the same answers as plain scalars (what compact() prints)
            "gather_impl": ("librxgpu: one ncclGroup{ncclGather avg int64, ncclGather samples int32}" if "rxgpu_power_gather" in c["gather"]
                            else "torch.distributed.gather" if "torch.distributed" in c["gather"] else "none (single process)"),
            "rccl_library": os.path.basename(c["gather"].split(" from ")[1].split(" ")[0]) if " from " in c["gather"] else None,
            "passes_per_step": c["passes_per_step"],
            "rx_power_gather_is_product": c.get("gather_is_product"), "rccl_comm_error": c.get("rccl_comm_error"),
            "rx_power_1gpu_same_run_Mbins_per_s": pw.get("one_gpu_same_run_Mbins_per_s"),
            "rx_power_speedup_vs_1gpu": (pw["value"] / pw["one_gpu_same_run_Mbins_per_s"]) if pw.get("one_gpu_same_run_Mbins_per_s") else None,
            "rx_power_parity_tunes": (pw.get("parity_sharded") or pw.get("parity") or {}).get("parity_tunes_compared"),
            "rx_power_parity_ranks": (pw.get("parity_sharded") or {}).get("parity_ranks"),
            "rx_power_padding_rows_zero": (pw.get("parity_sharded") or {}).get("parity_padding_rows_zero"),
            "rx_fm_replicas_MSample_per_s": result.get("value")})
        pr = pw.get("projected_scaling")
        if pr:                                                    # a projection from one GPU, labelled; the full record has the inputs
            cfg["rx_power_projected_speedup"] = {w: round(v["speedup_gather_overlapped"], 3) for w, v in pr["by_world"].items()}
        # a scaling figure must not come from the fallback unnoticed (round-5 advisory): when torch.distributed.gather stood in for the
        # product's gather, the product's keys stay null and the measured numbers move to keys that say what they are
        if c.get("gather_is_product") is False:
            for k in ("rx_power_Mbins_per_s", "rx_power_ms_per_step", "rx_power_speedup_vs_1gpu"):
                cfg[k + "_torch_gather_fallback"] = cfg[k]
                cfg[k] = None
    if not isinstance(roof, dict):
        return
    legs = {}
    for label, v in (result.get("rx_fm_variants") or {}).items():
        t = v.get("traffic") or {}
        legs[VARIANT_SHORT.get(label, label)] = {"bound": "hbm", "frac": v["frac_of_hbm_peak"], "MSample_per_s": v["value"], "steps": v["steps"],
                                                 "traffic_over_algorithmic": t.get("traffic_over_algorithmic"),
                                                 "parity_ok": (v.get("parity") or {}).get("parity_ok")}
    if pw is not None and pw is not result:
        r = pw["roofline"]
        legs["rx_power N=4096 (configs[2])"] = {"bound": r["bound"], "frac": r["frac"], "hbm_frac": r["hbm_frac"], "Mbins_per_s": pw["value"],
                                                "avg_launch_ms": r["avg_launch_ms"],
                                                "traffic_over_algorithmic": (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None,
                                                "parity_ok": pw.get("parity_ok")}
        for label, v in (pw.get("other_geometries") or {}).items():
            legs["rx_power " + label.split(",")[0][:60]] = {"bound": v.get("bound", "hbm"), "frac": v.get("frac", v["frac_of_hbm_peak"]), "hbm_frac": v["frac_of_hbm_peak"],
                                                            "traffic_frac": v.get("traffic_frac_of_hbm_peak"), "valu_frac": v.get("valu_frac"),
                                                            "Mbins_per_s": v["Mbins/s"], "N": v["N"], "parity_ok": (v.get("parity") or {}).get("parity_ok")}
    ch = result.get("channeliser")
    if ch:
        r = ch["roofline"]
        legs["channeliser 256 ch"] = {"bound": "valu", "frac": r["valu"]["frac"], "hbm_frac": r["frac"], "MSample_per_s": ch["value"],
                                      "traffic_over_algorithmic": (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None,
                                      "parity_ok": (ch.get("parity") or {}).get("parity_ok"), "parity_checker": (ch.get("parity") or {}).get("parity_checker")}
        for label, v in (ch.get("other_modes") or {}).items():
            legs["channeliser 256 ch, " + label] = {"bound": "valu", "frac": v.get("valu_frac"), "hbm_frac": v["frac_of_hbm_peak"], "MSample_per_s": v["value"],
                                                    "parity_ok": (v.get("parity") or {}).get("parity_ok")}
        if ch.get("nco_mode"):
            legs["channeliser 256 ch, NCO -> low_pass mode"] = {"bound": "valu", "MSample_per_s": ch["nco_mode"]["value"], "frac": ch["nco_mode"].get("valu_frac"),
                                                               "parity_ok": (ch["nco_mode"].get("parity") or {}).get("parity_ok")}
    for label, v in ((result.get("sdr_convert") or {}).get("legs") or {}).items():
        legs["rx_sdr " + label] = {"bound": "hbm", "frac": v["frac_of_hbm_peak"], "GBs": v["GB/s"], "frac_of_box_ceiling": v.get("frac_of_box_ceiling"),
                                   "parity_ok": (result["sdr_convert"]).get("parity_ok")}
    hf = (result.get("host_fed") or {}).get("legs") or {}
    for label, v in hf.items():
        legs["rx_fm host-fed " + label] = {"bound": "pcie", "frac": v["frac_of_pcie"], "GS_per_s": v["GS/s"],
                                           "parity_ok": ((result.get("host_fed") or {}).get("parity") or {}).get("parity_ok")}
    roof["legs"] = legs


LEG_SHORT = {"rx_fm ds=6 (-M wbfm default)": "fm_ds6", "rx_fm ds=5 (configs[0], 240 kHz)": "fm_ds5", "rx_fm -F ds=128": "fm_F_ds128", "rx_fm -M wbfm -F 9": "fm_wbfm_F9",
             "rx_power N=4096 (configs[2])": "pw_4096", "channeliser 256 ch": "chan256", "channeliser 256 ch, NCO -> low_pass mode": "chan256_nco",
             "rx_power full-scale input": "pw_4096_fullscale_hamming", "rx_power -f 100M:100.1M:10 -F 9: N=16384": "pw_16384_F9",
             "rx_power -f 100M:100.1M:10 (boxcar ds=28)": "pw_16384_boxcar", "rx_power -f 100M:100.2M:10 (boxcar ds=14)": "pw_32768_boxcar",
             "rx_power -f 100M:102.8M:20": "pw_262144",
             "channeliser 256 ch, -A std (NBFM default)": "chan256_std", "channeliser 256 ch, deemph + low_pass_real": "chan256_audio"}
COMPACT_LIMIT = 8192           # the driver kept the 4-16 KB lines of rounds 1-3 and dropped round 4's 25 KB one


def _num(x, digits=5):
    """a finite float rounded to `digits` significant figures, ints and bools as they are, anything else (NaN, inf, text) -> None"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def leg_key(label):
    if label in LEG_SHORT:
        return LEG_SHORT[label]
    k = label.replace("rx_power ", "pw ").replace("rx_sdr ", "sdr ").replace("rx_fm host-fed ", "hostfed ")
    k = re.sub(r"\(rxgpu_pin\)|one tune|input", "", k)
    return re.sub(r"[^A-Za-z0-9.=>+-]+", "_", k).strip("_")[:40]


def compact(result, full_path=None):
    """The ONE stdout line: the contract's scalars, `config`, `roofline` (with every other leg as {b: bound, f: fraction of it, t: HBM traffic over
    algorithmic bytes, ok: parity verdict}), `cpu_baseline`, `parity_ok` -- numbers and short names only, no nested object deeper than 3, under
    COMPACT_LIMIT bytes.  Everything else this run measured is the full record (`full`: a file under gpurun_out/, also on stderr)."""
    out = {k: (_num(result.get(k), 7) if k in ("value", "ms_per_step") else result.get(k))
           for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = result.get("config") or {}
    c = {"workload": cfg.get("workload_short") or str(cfg.get("workload", ""))[:120]}
    for k in ("blocks_per_step", "block_complex_samples", "bytes_per_step", "parallelism", "host_fixups_timed_loop", "settle_ms", "passes_per_step", "n_ranks", "rccl_ranks",
              "rccl_gathers_enqueued", "gather_impl", "rx_power_gather_is_product", "rccl_comm_error", "rccl_library", "gather_bytes_per_rank", "tunes_per_rank", "tunes_rank0", "rx_power_Mbins_per_s", "rx_power_ms_per_step",
              "rx_power_1gpu_same_run_Mbins_per_s", "rx_power_speedup_vs_1gpu", "scan_us_rank0", "gather_us_rank0", "rx_power_parity_ok", "rx_power_parity_tunes",
              "rx_power_parity_ranks", "rx_power_padding_rows_zero", "rx_power_cpu_baseline_Mbins_per_s_1core", "rx_fm_replicas_MSample_per_s",
              "rx_power_Mbins_per_s_torch_gather_fallback", "rx_power_speedup_vs_1gpu_torch_gather_fallback", "rx_power_projected_speedup"):
        if k in cfg and cfg[k] is not None:
            c[k] = _num(cfg[k], 6) if isinstance(cfg[k], float) else cfg[k]
    out["config"] = c
    roof = result.get("roofline") or {}
    r = {k: (_num(roof.get(k), 6) if not isinstance(roof.get(k), str) else roof.get(k))
         for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "frac_of_box_read_only",
                   "hbm_achieved", "hbm_frac") if k in roof}
    if isinstance(roof.get("box_ceilings"), dict):
        r["box_read_only_GBs"] = _num(roof["box_ceilings"].get("read_only_GBs"), 5)
    legs = {}
    for label, v in (roof.get("legs") or {}).items():
        t = v.get("traffic_over_algorithmic")
        if t is None and v.get("traffic_frac") and v.get("hbm_frac"):
            t = v["traffic_frac"] / v["hbm_frac"]
        e = {"b": v.get("bound"), "f": _num(v.get("frac"), 4), "t": _num(t, 4), "ok": v.get("parity_ok")}
        if v.get("bound") not in ("hbm", "pcie") and v.get("hbm_frac") is not None:
            e["hbm_f"] = _num(v["hbm_frac"], 4)
        legs[leg_key(label)] = e
    if legs:
        r["legs"] = legs
    out["roofline"] = r
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": _num(cb.get("value"), 6), "median": _num(cb.get("median"), 6), "unit": cb.get("unit"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "sample": cb.get("sample_short") or str(cb.get("sample", ""))[:160]}
        ac = cb.get("all_cores")
        if isinstance(ac, dict) and ac.get("value"):
            out["cpu_baseline"]["all_cores_value"] = _num(ac["value"], 6)
            out["cpu_baseline"]["all_cores"] = ac.get("cores")
    if "parity_ok" in result:
        out["parity_ok"] = result["parity_ok"]
    if result.get("parity_checked_samples") is not None:
        out["parity_checked_samples"] = result["parity_checked_samples"]
    if result.get("parity_checker"):
        out["parity_checker"] = str(result["parity_checker"]).split(" ")[0]
    if result.get("parity_all_legs"):
        out["parity_all_legs"] = result["parity_all_legs"]
    if full_path:
        out["full"] = full_path
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:                               # cannot happen with today's legs (~3 KB); if it ever does, the legs go first
        out["roofline"].pop("legs", None)
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    return line


def self_launch(argv, n):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
    bench.py <the same arguments>` on a free local port (one process per GPU; the launcher's own stdout carries rank 0's line)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: --gpus %d without a torchrun environment: exec %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=80.0, help="untimed milliseconds of the same step in front of the warm-up steps (0: none)")
    ap.add_argument("--blocks", type=int, default=16384, help="rx_fm blocks of 131072 complex samples per step (16384 = 8 GiB of cs16)")
    ap.add_argument("--passes", type=int, default=512, help="rx_power scanner() passes per step (one report interval)")
    ap.add_argument("--workload", default="both", choices=["both", "rx_fm", "rx_power", "chan", "sdr"])
    ap.add_argument("--cpu-seconds", type=float, default=9.0, help="CPU baseline budget per path (0 = skip)")
    ap.add_argument("--prof-level", type=int, default=1)
    ap.add_argument("--cpu-worker", default=None, choices=["fm", "power"], help=argparse.SUPPRESS)
    ap.add_argument("--cpu-lib", default="", help=argparse.SUPPRESS)
    ap.add_argument("--variants", default="all", choices=["all", "none"],
                    help="rx_fm side figures (ds=6, ds=5/240k, -F cascade, host-fed); `none` keeps the per-kernel averages of a "
                         "rocprofv3 run of this command to the headline launches (the ds=6 chain launches the same decimator kernel)")
    ap.add_argument("--require-librxgpu-gather", action="store_true",
                    help="N>1 only: FAIL if librxgpu's own RCCL communicator cannot be created on every rank (default: gather through torch.distributed "
                         "instead, say so in the line -- gather_impl, rccl_comm_error -- and set rx_power_gather_is_product false)")
    ap.add_argument("--allow-torch-gather", action="store_true", help=argparse.SUPPRESS)      # round 4's spelling of what is now the default
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size comparison with the CPU reference")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where rank 0 writes the full record (every leg, every parity dict); stdout carries the compact line only")
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker, args.cpu_seconds, args.cpu_lib)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_launch(sys.argv[1:], args.gpus)                 # does not return
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # stdout carries ONE line, rank 0's compact record: whatever else a library writes to file descriptor 1 on any rank (gloo's "[Gloo] Rank ..."
    # connection notes, an RCCL debug line, the reference's own printf in a checker) goes to stderr
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import rx_tools_amd as R
    from rx_tools_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # $RXGPU_BENCH_SHARE_GPU=1 (test hook, tests/test_gpu_power.py): every rank on device 0 and torch.distributed over gloo -- with
    # $RXGPU_RCCL_LIB pointing at tests/fake_rccl.c this runs the N > 1 path of this file on a box with ONE GPU (RCCL refuses two
    # ranks per device).  Not a measurement.
    share_gpu = os.environ.get("RXGPU_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    R.check(R.lib().rxgpu_init(local))
    L = R.lib()
    dev = torch.device("cuda", local)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def prof(name):
        ms, n = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def pmc_summary():
        f = newest_profile("pmc_summary.json")
        try:
            return json.load(open(f)), "profiles/" + os.path.basename(f)
        except (OSError, ValueError, TypeError):
            return {}, None

    def pmc_kernel(pmc, prefix, field):
        """the summary's entry for a kernel (template arguments included in its key), the one that has `field`"""
        for k, v in pmc.items():
            if k.startswith(prefix) and isinstance(v, dict) and v.get(field):
                return v
        raise KeyError(prefix)

    result = {}
    parity_all = {}            # leg -> verdict of its check against the CPU reference at bench size

    # What THIS box's HBM gives plain streams in the access shapes of the HBM-bound kernels, arithmetic taken out (rxgpu_diag_stream_rate:
    # non-temporal 16-byte pieces, grid-stride, two in flight per lane): the boxes of the pool differ by +-4 % on pure reads and by more on
    # mixed read/write traffic, so every HBM-bound leg is also printed as a fraction of the ceiling measured in the same process.
    def box_rate(mode, units=1 << 26, reps=5):
        g = C.c_double(0)
        R.check(L.rxgpu_diag_stream_rate(mode, units, reps, C.byref(g)))
        return g.value
    box = {"read_only_GBs": box_rate(0), "copy_1_1_GBs": box_rate(1), "expand_1_2_GBs": box_rate(2, 1 << 27), "shrink_2_1_GBs": box_rate(3),
           "read_only_grid_stride_loop_GBs": box_rate(4),
           "how": "rxgpu_diag_stream_rate: 1 GiB read per launch, 5 launches, hipEvents; read + written bytes"}

    # ------------------------------------------------------------------ rx_fm (headline)
    if args.workload in ("both", "rx_fm"):
        block_len = 2 * 131072
        n_blocks = args.blocks
        T = n_blocks * (block_len // 2)
        d_iq = device_capture(torch, dev, T, seed=12345 + rank)
        torch.cuda.synchronize()
        d_out = torch.zeros(T // 118 + 64, dtype=torch.int16, device=dev)
        hp = dict(downsample=118)
        s = R.FmStream(R.FmParams.wbfm(**hp), n_blocks, block_len)
        # Untimed, in front of the W warm-up steps: --settle-ms of the same step.  The first ~50 ms of this loop after the device has done other work
        # (the box probes, the capture generator) run 2-3 % slower than its steady state -- twelve alternating runs on one box: 0.790-0.817 of HBM
        # without, 0.805-0.832 with 60 ms of it, 0.831-0.833 with --warmup 50 (profiles/r06_settle_ab.txt); a read-only probe of the same length does
        # not do it.  `value` is a sustained rate; the channeliser leg below has had the same for two rounds.  config.settle_ms says what ran.
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            s.run_async(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), d_out.numel())
            s.wait()
        for _ in range(args.warmup):
            s.run_async(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), d_out.numel())
        s.wait()
        L.rxgpu_prof_reset()
        L.rxgpu_prof_enable(args.prof_level)
        barrier()
        t0 = time.perf_counter()
        # pipelined: run r+1's decimator (stream A) overlaps run r's audio stages (stream B); one wait at the end
        for _ in range(args.steps):
            s.run_async(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), d_out.numel())
        s.wait()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        L.rxgpu_prof_enable(0)
        ms, launches = prof("fm_decimate")
        fixups = s.host_fixups
        s.close()
        del d_out
        parity = {}
        # SURVEY 8(d) config 2 also names the -F variant; and the -M wbfm default decimation; BASELINE configs[0] is ds=5 at 240 kHz.
        # Same buffer, same pipelined loop, a few steps each; reported beside the headline, not part of `value`.
        variants, variant_kw = {}, {}
        if world == 1 and args.variants == "all":
            todo = [("-M wbfm default, downsample=6", dict(downsample=6), 6),
                    ("BASELINE configs[0] geometry: -s 240000, downsample=5, deemph_a=19", dict(downsample=5, rate_out=240000, deemph_a=19), 5),
                    ("-F cascade, downsample_passes=7 (ds=128)", dict(downsample_passes=7), 128),
                    ("-F 9 cascade as -M wbfm -F 9 sets it: downsample_passes=3 (ds=8) + droop FIR", dict(downsample_passes=3, comp_fir_size=9), 8)]
            for label, kw, per_in in todo:
                d_o = torch.zeros(T // per_in + 64, dtype=torch.int16, device=dev)
                sv = R.FmStream(R.FmParams.wbfm(**kw), n_blocks, block_len)
                for _ in range(2):
                    sv.run_async(d_iq.data_ptr(), n_blocks, block_len, d_o.data_ptr(), d_o.numel())
                sv.wait()
                k = max(20, args.steps)                                # the side figures get as many timed steps as the headline
                # two timed loops of k steps, the faster one reported (both kept in `loops_ms_per_step`): a side figure should not hang on one
                # transient of the box (round 6: a full run showed ds=6 at 0.42 and 0.50 minutes apart on unchanged code); `value` is never treated so
                loops = []
                for _ in range(2):
                    L.rxgpu_prof_reset()
                    L.rxgpu_prof_enable(2)
                    tv = time.perf_counter()
                    for _ in range(k):
                        sv.run_async(d_iq.data_ptr(), n_blocks, block_len, d_o.data_ptr(), d_o.numel())
                    sv.wait()
                    tv = time.perf_counter() - tv
                    L.rxgpu_prof_enable(0)
                    loops.append(tv)
                    if tv > min(loops):
                        continue                                       # the stage times below stay those of the faster loop
                    stages = {}
                    for nm in ("fm_decimate", "fm_fifth", "fm_fifth2", "fm_droop", "fm_disc", "fm_deemph", "fm_resample"):
                        sms, sn = prof(nm)
                        if sn:
                            stages[nm] = round(sms / sn * 1e3, 1)
                tv = min(loops)
                variant_kw[label] = kw
                variants[label] = {"value": T * k / tv / 1e6, "unit": "MSample/s", "ms_per_step": tv / k * 1e3, "steps": k,
                                   "loops_ms_per_step": [round(t / k * 1e3, 4) for t in loops], "timing": "the faster of two loops of %d pipelined steps" % k,
                                   "frac_of_hbm_peak": 4.0 * T * k / tv / 1e9 / HBM_PEAK_GBS, "stage_us_per_step": stages,
                                   "host_fixups": int(sv.host_fixups)}
                sv.close()
                del d_o
        # Every chain that was timed, in the launch shape that was timed (all n_blocks in one run, then a chained second run), against
        # the CPU reference over the WHOLE capture: output sample by sample and every carry.  The GPU sequences run first; the
        # reference legs (one forked checker per chain, tests/parity_at_size.py) then run side by side on the host cores.
        h_iq = None
        if rank == 0 and not args.no_parity:                      # N > 1: rank 0's replica is checked, the others wait at the next collective
            PA = parity_module()
            tail = max(1, n_blocks // 16)
            legs = {"headline": (hp, PA.fm_gpu_sequence(torch, R, d_iq, n_blocks, block_len, tail, hp))}
            for label, kw in variant_kw.items():
                legs[label] = (kw, PA.fm_gpu_sequence(torch, R, d_iq, n_blocks, block_len, tail, kw))
            h_iq = d_iq.cpu().numpy()
            verdicts = PA.fm_check_many(h_iq, n_blocks, block_len, legs)
            parity = dict(verdicts["headline"])
            parity["parity_sequence"] = ("2 pipelined runs (%d + %d blocks), carries chained on the device; output and every carry of the chain "
                                         "(now_r/now_j/prev_index or lp_*_hist/droop_*_hist, pre_r/pre_j, now_lpr/prev_lpr_index) compared" % (n_blocks, tail))
            parity["parity_legs"] = {lb: bool(v["parity_ok"]) for lb, v in verdicts.items()}
            for label in variant_kw:
                variants[label]["parity"] = verdicts[label]
            parity["parity_ok"] = all(v["parity_ok"] for v in verdicts.values())
            headline_got = legs["headline"][1]["got"]
            del legs
        # ---- host-fed leg (PCIe-inclusive, never `value`) and the drop-in's per-block latency
        host_fed = {}
        if world == 1 and args.variants == "all":
            hb = min(n_blocks, 4096)                                  # 2 GiB of capture from host memory
            h_iq = h_iq[: hb * block_len] if h_iq is not None else d_iq[: hb * block_len].cpu().numpy()
            h_out = np.zeros(hb * (block_len // 2) // 118 + 4096, np.int16)
            hf_same = None
            sh = R.FmStream(R.FmParams.wbfm(**hp), hb, block_len)
            legs = {}
            for label, pin in (("pinned (rxgpu_pin)", True), ("pageable", False)):
                if pin:
                    R.check(L.rxgpu_pin(h_iq.ctypes.data, h_iq.nbytes))
                try:
                    nh, _ = sh.run_host(h_iq.ctypes.data, hb, block_len, h_out.ctypes.data, h_out.size)  # staging allocated, pages touched
                    if parity and hf_same is None:
                        # the same chain from host memory, from zero carries: a prefix of what the device-resident run (checked above) produced
                        hf_same = bool(nh <= headline_got.size and np.array_equal(h_out[:nh], headline_got[:nh]))
                        hf_n = nh
                    t0 = time.perf_counter()
                    reps = 3
                    for _ in range(reps):
                        sh.run_host(h_iq.ctypes.data, hb, block_len, h_out.ctypes.data, h_out.size)
                    th = (time.perf_counter() - t0) / reps
                finally:
                    if pin:
                        R.check(L.rxgpu_unpin(h_iq.ctypes.data))
                gbs = h_iq.nbytes / th / 1e9
                legs[label] = {"GS/s": hb * (block_len // 2) / th / 1e9, "GB/s": gbs, "frac_of_pcie": gbs / PCIE_PEAK_GBS, "ms": th * 1e3}
            sh.close()
            host_fed = {"metric": "rxgpu_fm_stream_run_host: %d blocks (%.1f GiB) from host memory, 64 MiB chunks, H2D on a copy stream "
                                  "overlapped with the demodulation of the previous chunk, results copied back" % (hb, h_iq.nbytes / 2 ** 30),
                        "pcie_peak_GBs": PCIE_PEAK_GBS, "legs": legs}
            if hf_same is not None:
                host_fed["parity"] = {"parity_ok": hf_same, "parity_how": "output of the host-fed runs == the first %d int16 of the device-resident "
                                                                        "sequence that was checked against the reference" % hf_n}
                parity["parity_legs"]["host_fed"] = hf_same
                parity["parity_ok"] = parity["parity_ok"] and hf_same
            del h_iq
            # drop-in: rxgpu_callback + rxgpu_full_demod on the reference's own structs, block after block
            from rx_tools_amd.structs import DemodState, DongleState
            os.environ["RXGPU_DROPIN_TIMING"] = "1"                   # read once, at the first drop-in call of the process
            L.rxgpu_knobs_reload()                                    # ... from the library's knob snapshot
            d = DemodState()
            d.rate_in = d.rate_out = 170000
            d.rate_out2, d.custom_atan, d.deemph, d.deemph_a, d.downsample = 32000, 1, 1, 13, 118
            d.post_downsample, d.output_scale, d.squelch_hits, d.adc_block_const, d.rdc_block_const = 1, 1, 11, 9, 9
            libc = C.CDLL(None)
            libc.pthread_rwlock_init(C.byref(d, DemodState.rw.offset), None)
            libc.pthread_cond_init(C.byref(d, DemodState.ready.offset), None)
            libc.pthread_mutex_init(C.byref(d, DemodState.ready_m.offset), None)
            g = DongleState()
            g.demod_target = C.pointer(d)
            blk = d_iq[:block_len].cpu().numpy().copy()
            R.check(L.rxgpu_dropin_pin(C.addressof(d), C.addressof(g)))      # what the INTEGRATION.md patch does once at start-up
            R.check(L.rxgpu_pin(blk.ctypes.data, blk.nbytes))                # ... and for the dongle thread's read buffer (rtl_fm.c:873)
            for _ in range(5):
                L.rxgpu_callback(blk.ctypes.data, block_len, C.addressof(g))
                L.rxgpu_full_demod(C.addressof(d))
            nb = 200
            phases = (C.c_double * 7)()
            L.rxgpu_dropin_timing(phases, 7)                          # clear (the table fills only with $RXGPU_DROPIN_TIMING=1, set below)
            t0 = time.perf_counter()
            for _ in range(nb):
                L.rxgpu_callback(blk.ctypes.data, block_len, C.addressof(g))
            t_cb = (time.perf_counter() - t0) / nb
            L.rxgpu_dropin_timing(phases, 7)
            t0 = time.perf_counter()
            for _ in range(nb):
                L.rxgpu_callback(blk.ctypes.data, block_len, C.addressof(g))
                L.rxgpu_full_demod(C.addressof(d))
            t_both = (time.perf_counter() - t0) / nb
            L.rxgpu_dropin_timing(phases, 7)
            ph = list(phases)
            breakdown = None
            if ph[5] and ph[6]:
                breakdown = {"callback: H2D + pre-stage kernel + D2H (enqueue..sync)": ph[0] / ph[5], "callback: hand-off (rw lock, memcpy, signal)": ph[1] / ph[5],
                             "full_demod: set-up (params, side-car, carries in)": ph[2] / ph[6], "full_demod: run (kernels, carries back)": ph[3] / ph[6],
                             "full_demod: D2H of result[] and lowpassed[], struct fields": ph[4] / ph[6]}
            R.check(L.rxgpu_dropin_unpin(C.addressof(d), C.addressof(g)))
            L.rxgpu_unpin(blk.ctypes.data)
            L.rxgpu_dropin_release(C.addressof(d))
            host_fed["dropin_block_us"] = {"callback": t_cb * 1e6, "callback+full_demod": t_both * 1e6,
                                           "block_complex_samples": block_len // 2,
                                           "MSample/s": (block_len // 2) / t_both / 1e6, "phase_us": breakdown,
                                           "note": "one 1 MiB block per call pair: H2D raw, pre-stage kernel, D2H into buf16/lowpassed[]; "
                                                   "full_demod consumes the copy left in HBM: k_fm_block_dd + k_fm_row_audio, one copy back (header, result[], lowpassed[])"}
        del d_iq
        torch.cuda.empty_cache()
        value = world * T * args.steps / dt / 1e6
        traffic, traffic_src = None, None
        pmc, pmc_src = pmc_summary()
        # HBM bytes every chain moves per step (all its kernels: FETCH_SIZE x 2 + WRITE_SIZE from the counter passes in profiles/), beside the
        # 4 B per input sample the metric counts: what the chains below the roofline spend their time on
        chain_keys = {"-M wbfm default, downsample=6": "-M wbfm default, downsample=6",
                      "BASELINE configs[0] geometry: -s 240000, downsample=5, deemph_a=19": "configs[0] geometry: ds=5, 240 kHz",
                      "-F cascade, downsample_passes=7 (ds=128)": "-F cascade, 7 passes (ds=128)",
                      "-F 9 cascade as -M wbfm -F 9 sets it: downsample_passes=3 (ds=8) + droop FIR": "-M wbfm -F 9: 3 passes + droop FIR"}
        for label, v in variants.items():
            c = pmc.get("_chains", {}).get(chain_keys.get(label, ""), None)
            if c:
                scale = (4.0 * T) / c["algorithmic_bytes_per_step"]
                v["traffic"] = {"hbm_bytes_per_step": c["hbm_bytes_per_step"] * scale, "algorithmic_bytes_per_step": 4.0 * T,
                                "traffic_over_algorithmic": c["traffic_over_algorithmic"],
                                "achieved_GBs_of_traffic": c["hbm_bytes_per_step"] * scale / (v["ms_per_step"] * 1e-3) / 1e9,
                                "source": pmc_src + " (per-kernel table: " + pmc_src.replace("summary", "chains") + ")"}
        try:
            k = pmc_kernel(pmc, "k_fm_decimate<false, true, true", "hbm_bytes_per_launch")
            # measured on 2^30-sample launches (--blocks 8192); bytes scale with the launch
            traffic = k["hbm_bytes_per_launch"] * (T / float(k.get("launch_samples", 1 << 30)))
            traffic_src = pmc_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per gfx950 note)"
        except (KeyError, TypeError):
            pass
        achieved = (4.0 * T) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        result.update({
            "metric": "rx_fm full_demod complex IQ MSample/s (20 Msps WBFM geometry, ds=118)",
            "value": value, "unit": "MSample/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/int32", "dtype_note": "fp32 fma for the cs16 scale (equal on all 65536 inputs), fp64 for one atan2 per block",
            "data": "synthetic",
            "config": {"workload": "rx_fm WBFM 20.06 Msps -> 170 ksps -> 32 ksps: callback scale+rotate, low_pass ds=118, "
                                   "polar_disc_fast, deemph a=13, low_pass_real (BASELINE configs[1])",
                       "workload_short": "rx_fm WBFM 20.06 Msps, callback+full_demod, ds=118 -> 170 ksps -> 32 ksps (BASELINE configs[1])",
                       "blocks_per_step": n_blocks, "block_complex_samples": block_len // 2,
                       "bytes_per_step": 4 * T, "parallelism": "replicas x%d (rx_fm does not shard)" % world,
                       "capture": "non-repeating, generated on the device (seeded): FM carrier at -fs/4, 1 kHz tone, 75 kHz deviation, +-128 LSB noise",
                       "host_fixups_timed_loop": int(fixups), "settle_ms": args.settle_ms},
            "rx_fm_variants": variants,
            "host_fed": host_fed,
            "roofline": {"bound": "hbm", "kernel": "k_fm_decimate (F0+F1+F2)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "box_ceilings": box, "frac_of_box_read_only": achieved / box["read_only_GBs"],
                         "algorithmic_bytes_per_launch": 4 * T, "avg_launch_ms": (ms / launches) if launches else None},
        })
        result.update(parity)
        if parity:
            parity_all["rx_fm"] = bool(parity["parity_ok"])
        if rank == 0 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline_fm(block_len, args.cpu_seconds)
            result["cpu_baseline"]["reference_default_build_O0"] = cpu_subprocess("fm", min(3.0, args.cpu_seconds), "libref_fm_O0.so")
            result["cpu_baseline"]["all_cores"] = cpu_all_cores("fm", min(3.0, args.cpu_seconds))

    # ------------------------------------------------------------------ rx_power
    if args.workload in ("both", "rx_power"):
        plan = R.plan_range("24M:1.7G:1k", 0.0, 1)
        n = 1 << plan.bin_e
        total_tunes = plan.tune_count
        lo, mine, per = shard.tune_range(rank, world, total_tunes)   # contiguous tune ranges, SURVEY section 8(e)
        passes = args.passes
        wc, sw = R.window_coefs("rectangle", n), R.sine_table(plan.bin_e)
        ps = R.PowerScan(R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0),
                         per, wc, sw)
        g = torch.Generator(device=dev)
        g.manual_seed(777 + rank)
        d_in = torch.randint(-100, 101, (passes, max(mine, 1), plan.buf_len), dtype=torch.int16, device=dev, generator=g)
        # two report-interval buffers: the gather of interval k (behind the scan on the library's stream) never waits for the host
        d_avgs = [torch.zeros((per, n), dtype=torch.int64, device=dev) for _ in range(2)]   # padded to `per` rows for the gather
        d_smps = [torch.zeros(per, dtype=torch.int32, device=dev) for _ in range(2)]
        d_avg_all = torch.zeros((world, per, n), dtype=torch.int64, device=dev) if rank == 0 else None
        d_smp_all = torch.zeros((world, per), dtype=torch.int32, device=dev) if rank == 0 else None
        comm, gather_impl = None, "single process (no collective)"
        if world > 1:
            # the product's own communicator (librccl bound by librxgpu).  A scaling curve measured through torch.distributed.gather would not be the
            # product's: if the communicator cannot be created the run goes on through torch's gather and the line SAYS so (gather_impl,
            # rccl_comm_error, rx_power_gather_is_product false) -- a SCALE record with a labelled fallback beats none; --require-librxgpu-gather fails instead
            comm_err = None
            try:
                comm = shard.Comm.from_torch_distributed()
                gather_impl = "librxgpu rxgpu_power_gather: one ncclGroup {ncclGather(avg int64), ncclGather(samples int32)} from %s on the library's stream" % shard.Comm.library()
            except Exception as e:                               # noqa: BLE001
                comm_err = repr(e)
            ok = torch.tensor([1 if comm is not None else 0], device="cpu" if share_gpu else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:                              # all ranks or none
                if comm is not None:
                    comm.close()
                    comm = None
                if args.require_librxgpu_gather:
                    raise SystemExit("bench.py: librxgpu's RCCL communicator could not be created on every rank (%s); "
                                     "(--require-librxgpu-gather)" % (comm_err or "another rank failed"))
                gather_impl = "torch.distributed.gather (backend nccl = RCCL) -- NOT the product's gather: librxgpu's communicator failed: %s" % (comm_err or "on another rank")
                sys.stderr.write("bench.py: %s\n" % gather_impl)
        # the fallback gathers into the same [world][per][N] block the product's gather fills, so the sharded parity check below reads one place
        gbuf = [d_avg_all[r] for r in range(world)] if (world > 1 and comm is None and rank == 0) else None
        sbuf = [d_smp_all[r] for r in range(world)] if (world > 1 and comm is None and rank == 0) else None
        state = {"k": 0}

        def step():
            b = state["k"] & 1
            state["k"] += 1
            if comm is not None or world == 1:
                R.check(L.rxgpu_power_scan_run_sharded(ps._h, comm._h if comm is not None else None, d_in.data_ptr(), passes, total_tunes,
                                                       d_avgs[b].data_ptr(), d_smps[b].data_ptr(), n,
                                                       d_avg_all.data_ptr() if rank == 0 else None,
                                                       d_smp_all.data_ptr() if rank == 0 else None, 0))
            else:
                if mine:
                    ps.run(d_in.data_ptr(), passes, mine, d_avgs[b].data_ptr(), d_smps[b].data_ptr())
                L.rxgpu_sync()                                   # order torch's stream behind the library's
                shard.gather_rows(d_avgs[b], dst=0, out=gbuf)
                shard.gather_rows(d_smps[b], dst=0, out=sbuf)

        # (see the rx_fm leg.  A step holds a collective when N > 1: every rank must run the SAME number of them, so the count is fixed --
        # a step of the whole sweep is ~5 ms on one GPU -- not taken from each rank's own clock)
        for _ in range(int(args.settle_ms / 5.0 + 0.999) if args.settle_ms > 0 else 0):
            step()
            L.rxgpu_sync()
        for _ in range(args.warmup):
            step()
        L.rxgpu_sync()
        L.rxgpu_prof_reset()
        L.rxgpu_prof_enable(args.prof_level)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        L.rxgpu_sync()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        L.rxgpu_prof_enable(0)
        ms, launches = prof("pw_fft")
        gms, gl = prof("pw_gather")
        observed = comm.observed if comm is not None else (0, 1)
        # N > 1: the whole sweep on rank 0's GPU alone, same process, same clocks -- the 1-GPU point of the strong-scaling curve inside the same line
        one_gpu_rate = None
        if world > 1:
            if rank == 0:
                ps1 = R.PowerScan(R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0), total_tunes, wc, sw)
                d_in1 = torch.randint(-100, 101, (passes, total_tunes, plan.buf_len), dtype=torch.int16, device=dev, generator=g)
                d_a1 = torch.zeros((total_tunes, n), dtype=torch.int64, device=dev)
                d_s1 = torch.zeros(total_tunes, dtype=torch.int32, device=dev)
                for _ in range(max(2, args.warmup)):
                    ps1.run(d_in1.data_ptr(), passes, total_tunes, d_a1.data_ptr(), d_s1.data_ptr())
                L.rxgpu_sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    ps1.run(d_in1.data_ptr(), passes, total_tunes, d_a1.data_ptr(), d_s1.data_ptr())
                L.rxgpu_sync()
                one_gpu_rate = passes * total_tunes * (plan.buf_len // 2) * args.steps / (time.perf_counter() - t1) / 1e6
                ps1.close()
                del d_in1, d_a1, d_s1
            barrier()
        # PROJECTION, labelled as one (no multi-GPU node was ever offered; never part of `value`): what one rank of a world of W would do per interval,
        # timed here as rank-shaped launches on this device -- ceil(599 / W) tunes x the same passes through the sharded entry point -- beside the time the
        # ONE gather of avg + samples needs at the per-link xGMI rate (the root takes in W - 1 blocks over W - 1 links side by side; the gather runs on the
        # copy stream under the next interval's scan).  When a real SCALE record exists, a shortfall against this points at the collective, not at occupancy.
        projection = None
        if world == 1 and args.variants == "all":
            XGMI_LINK_GBS = 153.0
            t_full = dt / args.steps
            projection = {"what": "PROJECTED from 1-GPU measurements of rank-shaped launches, NOT a multi-GPU measurement", "xgmi_link_GBs_assumed": XGMI_LINK_GBS,
                          "one_gpu_ms_per_interval": t_full * 1e3, "by_world": {}}
            for W in (2, 4, 8):
                per_w = (total_tunes + W - 1) // W
                for _ in range(2):
                    R.check(L.rxgpu_power_scan_run_sharded(ps._h, None, d_in.data_ptr(), passes, per_w, d_avgs[0].data_ptr(), d_smps[0].data_ptr(), n,
                                                           d_avgs[0].data_ptr(), d_smps[0].data_ptr(), 0))
                L.rxgpu_sync()
                reps, ts = max(5, args.steps // 2), []
                for _ in range(reps):                              # launch by launch, the median: one slow launch of a 0.6 ms shard must not halve the figure
                    t1 = time.perf_counter()
                    R.check(L.rxgpu_power_scan_run_sharded(ps._h, None, d_in.data_ptr(), passes, per_w, d_avgs[0].data_ptr(), d_smps[0].data_ptr(), n,
                                                           d_avgs[0].data_ptr(), d_smps[0].data_ptr(), 0))
                    L.rxgpu_sync()
                    ts.append(time.perf_counter() - t1)
                t_shard = sorted(ts)[len(ts) // 2]
                t_gather = (per_w * n * 8 + per_w * 4) / (XGMI_LINK_GBS * 1e9)
                projection["by_world"][str(W)] = {
                    "tunes_per_rank": per_w, "shard_ms": t_shard * 1e3, "gather_ms_at_link_rate": t_gather * 1e3, "gather_bytes_per_rank": per_w * n * 8 + per_w * 4,
                    "speedup_gather_overlapped": t_full / max(t_shard, t_gather), "speedup_gather_serial": t_full / (t_shard + t_gather),
                    "ideal": total_tunes / float(per_w)}
        bins_per_step_all = passes * total_tunes * (plan.buf_len // 2)
        bins_local = passes * mine * (plan.buf_len // 2)
        hbm_achieved = (4.0 * bins_local) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        pmc, pmc_src = pmc_summary()
        valu, pw_traffic = None, None
        try:
            k = pmc_kernel(pmc, "k_pw_fft4096", "SQ_INSTS_VALU")
            # counted on a 512-pass, 599-tune launch; instructions and bytes scale with the bins of the launch
            instr = k["SQ_INSTS_VALU"] * (bins_local / float(512 * 599 * 8192))
            valu = instr / (ms / launches * 1e-3) / 1e9
            if k.get("hbm_bytes_per_launch"):
                pw_traffic = k["hbm_bytes_per_launch"] * (bins_local / float(512 * 599 * 8192))
        except (KeyError, TypeError, ZeroDivisionError):
            pass
        pw_peak, pw_peak_src = valu_ceiling("k_pw_fft4096")
        pw = {
            "metric": "rx_power FFT bins/s (scanner() chain, -f 24M:1.7G:1k geometry)",
            "value": bins_per_step_all * args.steps / dt / 1e6, "unit": "Mbins/s", "n_gpus": world,
            "ms_per_step": dt / args.steps * 1e3, "scaling": "strong", "dtype": "int16/int32/int64", "one_gpu_same_run_Mbins_per_s": one_gpu_rate,
            "projected_scaling": projection,
            "config": {"workload": "599 tunes x 16384 int16, N=4096, 2 FFT blocks/tune/pass, rectangle window (BASELINE configs[2]/[3])",
                       "workload_short": "rx_power -f 24M:1.7G:1k: 599 tunes x 16384 int16, N=4096 (BASELINE configs[2]/[3])",
                       "passes_per_step": passes, "tunes_this_rank": mine, "tunes_per_rank_padded": per,
                       "rccl_ranks": observed[1], "rccl_ranks_source": "ncclCommCount, checked by rxgpu_comm_create" if comm is not None else "no communicator",
                       "rccl_gathers_enqueued": comm.gathers if comm is not None else 0,
                       "scan_us_per_step_rank0": (ms / launches * 1e3) if launches else None,
                       "parallelism": "tunes sharded x%d, one gather of avg[] + samples to rank 0 per step" % world,
                       "gather": gather_impl, "gather_us_per_step_rank0": (gms / gl * 1e3) if gl else None,
                       "gather_is_product": bool(comm is not None or world == 1), "rccl_comm_error": (comm_err[:160] if (world > 1 and comm_err) else None),
                       "gather_bytes_per_rank": per * n * 8 + per * 4},
            "roofline": {"bound": "valu", "kernel": "k_pw_fft4096 (P4-P8)",
                         "achieved": valu, "peak": pw_peak, "unit": "G wave-instr/s",
                         "frac": (valu / pw_peak) if valu else None,
                         "valu_source": (pmc_src + " (rocprofv3 --pmc SQ_INSTS_VALU per launch) / live launch time") if valu else None,
                         "peak_source": pw_peak_src or "nominal: 1024 SIMDs x 2.4 GHz / 4 (one wave64 integer instruction per quad-cycle)",
                         "nominal_peak": VALU_PEAK_GINSTR, "frac_of_nominal": (valu / VALU_PEAK_GINSTR) if valu else None,
                         "hbm_achieved": hbm_achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": hbm_achieved / HBM_PEAK_GBS,
                         "traffic": pw_traffic, "algorithmic_bytes_per_launch": 4 * bins_local,
                         "avg_launch_ms": (ms / launches) if launches else None},
        }
        # the launch that was timed -- all passes, all tunes of this rank -- once more into zeroed integrators, against the
        # reference's own scanner() over every tune (tests/parity_at_size.py: forked checkers, tunes dealt to the host cores)
        pw_parity_ok = True
        if world == 1 and not args.no_parity:
            PA = parity_module()
            da = torch.zeros((per, n), dtype=torch.int64, device=dev)
            dsm = torch.zeros(per, dtype=torch.int32, device=dev)
            ps.run(d_in.data_ptr(), passes, mine, da.data_ptr(), dsm.data_ptr())
            R.check(L.rxgpu_sync())
            pw["parity"] = PA.power_check("24M:1.7G:1k", 0.0, "rectangle", 1, 0, 0, d_in.cpu().numpy(), da.cpu().numpy(), dsm.cpu().numpy())
            pw_parity_ok = pw["parity"]["parity_ok"]
            del da, dsm
        # N > 1: the SHARDED interval once more into zeroed integrators -- every rank scans its tune range, the one grouped gather lands in
        # rank 0's [world][per][N] block -- and rank 0 compares every gathered row with the reference's own scanner() (rtl_power.c:670-772)
        # run on the input of the rank that owns the tune (regenerated here from that rank's seed; a checksum from the owner proves the
        # regeneration), and checks that the padding rows of a short last rank arrived as zeros.  This is the "bit-exact CSV" of
        # BASELINE configs[3]: csv_dbm prints avg[]/samples and nothing else (rtl_power.c:1047-1050).
        if world > 1 and not args.no_parity:
            for b in range(2):
                d_avgs[b].zero_()
                d_smps[b].zero_()
            if rank == 0:
                d_avg_all.fill_(-1)                               # a padding row the gather does not deliver as zeros would show
                d_smp_all.fill_(-1)
            torch.cuda.synchronize()
            state["k"] = 0
            step()
            L.rxgpu_sync()
            barrier()
            pw["config"]["rccl_gathers_enqueued"] = comm.gathers if comm is not None else 0      # warm-up + timed steps + this interval
            mysum = int(d_in.view(torch.int32).sum(dtype=torch.int64).item())
            t_sum = torch.tensor([mysum], dtype=torch.int64, device="cpu" if share_gpu else dev)
            t_all = [torch.zeros_like(t_sum) for _ in range(world)]
            dist.all_gather(t_all, t_sum)                        # a plain tensor collective (gloo in the one-GPU test, RCCL otherwise)
            sums = [int(t.item()) for t in t_all]
            if rank == 0:
                PA = parity_module()
                got_all, smp_all = d_avg_all.cpu().numpy(), d_smp_all.cpu().numpy()
                h_in = np.empty((passes, total_tunes, plan.buf_len), np.int16)
                got_avg = np.empty((total_tunes, n), np.int64)
                got_smp = np.empty(total_tunes, np.int64)
                regen_ok, pad_ok = True, True
                for r in range(world):
                    lo_r, mine_r, _ = shard.tune_range(r, world, total_tunes)
                    g_r = torch.Generator(device=dev)
                    g_r.manual_seed(777 + r)
                    d_r = torch.randint(-100, 101, (passes, max(mine_r, 1), plan.buf_len), dtype=torch.int16, device=dev, generator=g_r)
                    regen_ok = regen_ok and int(d_r.view(torch.int32).sum(dtype=torch.int64).item()) == sums[r]
                    h_in[:, lo_r:lo_r + mine_r] = d_r[:, :mine_r].cpu().numpy()
                    got_avg[lo_r:lo_r + mine_r] = got_all[r, :mine_r]
                    got_smp[lo_r:lo_r + mine_r] = smp_all[r, :mine_r]
                    pad_ok = pad_ok and not got_all[r, mine_r:].any() and not smp_all[r, mine_r:].any()
                    del d_r
                v = PA.power_check("24M:1.7G:1k", 0.0, "rectangle", 1, 0, 0, h_in, got_avg, got_smp)
                v["parity_what"] = ("rank 0's gathered [world][per][N] avg block and samples after one sharded interval of %d passes: every tune of every rank "
                                    "against the reference's scanner() on that rank's input" % passes)
                v["parity_inputs_regenerated_match_owner_checksums"] = bool(regen_ok)
                v["parity_padding_rows_zero"] = bool(pad_ok)
                v["parity_ranks"] = world
                v["parity_ok"] = bool(v["parity_ok"] and regen_ok and pad_ok)
                pw["parity_sharded"] = v
                pw_parity_ok = v["parity_ok"]
                pw["parity_ok"] = bool(pw_parity_ok)
                parity_all["rx_power_sharded"] = bool(pw_parity_ok)
                del h_in, got_all, got_avg
        # the drop-in (rxgpu_scan on the reference's own struct tuning_state array, rtl_power.c:1040): per-sweep cost with the sums left
        # on the device, and the one download per report interval (rxgpu_scan_sync)
        if world == 1 and args.variants == "all":
            from rx_tools_amd.structs import TuningState
            sweeps = 20
            h_bufs = d_in[0, :total_tunes].cpu().numpy().copy()
            h_avgs = np.zeros((total_tunes, n), np.int64)
            arr = (TuningState * total_tunes)()
            for t in range(total_tunes):
                arr[t] = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, C.cast(h_avgs[t].ctypes.data, C.POINTER(C.c_int64)), 0,
                                     plan.downsample, plan.downsample_passes, plan.crop, C.cast(h_bufs[t].ctypes.data, C.POINTER(C.c_int16)), plan.buf_len)
            R.check(L.rxgpu_scan_deferred(1))                         # what the INTEGRATION.md patch switches on: one merge per report interval
            for _ in range(3):
                R.check(L.rxgpu_scan(arr, total_tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
            R.check(L.rxgpu_scan_sync(arr, total_tunes))
            h_avgs[:] = 0
            for t in range(total_tunes):
                arr[t].samples = 0
            t0 = time.perf_counter()
            for _ in range(sweeps):
                R.check(L.rxgpu_scan(arr, total_tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
            R.check(L.rxgpu_sync())
            t_scan = (time.perf_counter() - t0) / sweeps
            t0 = time.perf_counter()
            R.check(L.rxgpu_scan_sync(arr, total_tunes))
            t_sync = time.perf_counter() - t0
            R.check(L.rxgpu_scan_deferred(0))
            pw["dropin_scan_us"] = {"rxgpu_scan_per_sweep": t_scan * 1e6, "rxgpu_scan_sync_per_interval": t_sync * 1e6, "sweeps": sweeps,
                                    "Mbins/s": total_tunes * (plan.buf_len // 2) / t_scan / 1e6,
                                    "zero_copy_input": bool(L.rxgpu_scan_zero_copy()),
                                    "sync_in_place": bool(L.rxgpu_scan_sync_in_place()),
                                    "pcie_bound_us_per_sweep": total_tunes * plan.buf_len * 2 / 55.5e3, "pcie_bound_us_per_sync": total_tunes * n * 8 / 55.5e3,
                                    "note": "599 tunes x 16384 int16 = 19.6 MB per sweep from the caller's separate buffers, page-locked in place once: one launch on the copy stream "
                                            "reads them across PCIe into the scan's input, the call returns when they have been read; avg[] (19.6 MB of int64) is merged in place "
                                            "by one launch per interval (host rows += device accumulators across PCIe); bounds: one hipMemcpy of the same bytes at 55.5 GB/s"}
            if not args.no_parity:
                want1 = torch.zeros((per, n), dtype=torch.int64, device=dev)
                ws = torch.zeros(per, dtype=torch.int32, device=dev)
                ps.run(d_in.data_ptr(), 1, mine, want1.data_ptr(), ws.data_ptr())      # pass 0 alone: part of the launch checked above
                R.check(L.rxgpu_sync())
                same = bool(np.array_equal(h_avgs, sweeps * want1[:total_tunes].cpu().numpy()) and all(arr[t].samples == sweeps * int(ws[t]) for t in range(total_tunes)))
                pw["dropin_scan_us"]["parity_ok"] = same
            # the interval AFTER a report: csv_dbm has zeroed every row it printed (rtl_power.c:815-817) and said so (rxgpu_csv_dbm does by itself,
            # the drop-in around the reference's own csv_dbm through rxgpu_scan_rows_cleared): the merge writes, it does not read
            h_avgs[:] = 0
            for t in range(total_tunes):
                arr[t].samples = 0
            R.check(L.rxgpu_scan_deferred(1))
            assert L.rxgpu_scan_rows_cleared(arr, total_tunes) == total_tunes
            for _ in range(2):
                R.check(L.rxgpu_scan(arr, total_tunes, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
            R.check(L.rxgpu_sync())
            t0 = time.perf_counter()
            R.check(L.rxgpu_scan_sync(arr, total_tunes))
            pw["dropin_scan_us"]["rxgpu_scan_sync_per_interval_after_csv_dbm"] = (time.perf_counter() - t0) * 1e6
            R.check(L.rxgpu_scan_deferred(0))
            if not args.no_parity:
                pw["dropin_scan_us"]["parity_ok"] = bool(pw["dropin_scan_us"]["parity_ok"] and np.array_equal(h_avgs, 2 * want1[:total_tunes].cpu().numpy())
                                                         and all(arr[t].samples == 2 * int(ws[t]) for t in range(total_tunes)))
                pw_parity_ok = pw_parity_ok and same
                del want1, ws
            L.rxgpu_scan_release()                                    # h_bufs is page-locked in place by rxgpu_scan: released BEFORE the array dies (rxgpu.h, LIFETIME)
            del arr, h_bufs, h_avgs
        ps.close()
        del d_in
        # two more geometries of SURVEY 8(d) config 3, one launch shape each, rank 0 only (not part of `value`)
        more = {}
        try:
            pw_legs_pmc = json.load(open(newest_profile("pmc_power_legs.json")))
        except (OSError, ValueError, TypeError):
            pw_legs_pmc = {}
        fft_peak, _ = valu_ceiling("k_pwm_tail")
        if world == 1 and args.variants == "all":
            for label, rng, boxcar, window, amp, fir, npasses in (
                    ("full-scale input, hamming window (every int16 wrap of the window product and the butterflies)", "24M:1.7G:1k", 1, "hamming", 32767, 0, passes),
                    ("-f 100M:100.1M:10 -F 9: N=16384, fifth_order x4 (ds=16) + droop FIR, one tune", "100M:100.1M:10", 0, "rectangle", 2000, 9, 4096),
                    ("-f 100M:100.1M:10 (boxcar ds=28), N=16384, one tune", "100M:100.1M:10", 1, "rectangle", 2000, 0, 4096),
                    ("-f 100M:100.2M:10 (boxcar ds=14), N=32768, one tune", "100M:100.2M:10", 1, "rectangle", 2000, 0, 2048),
                    ("-f 100M:102.8M:20, N=262144 (radix-16 passes through HBM + register-blocked tail), one tune", "100M:102.8M:20", 1, "hamming", 20000, 0, 256)):
                pl = R.plan_range(rng, 0.0, boxcar)
                nn = 1 << pl.bin_e
                p2 = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, boxcar, fir, 0), pl.tune_count,
                                 R.window_coefs(window, nn), R.sine_table(pl.bin_e))
                di = torch.randint(-amp, amp + 1, (npasses, pl.tune_count, pl.buf_len), dtype=torch.int16, device=dev, generator=g)
                da = torch.zeros((pl.tune_count, nn), dtype=torch.int64, device=dev)
                dsm = torch.zeros(pl.tune_count, dtype=torch.int32, device=dev)
                for _ in range(2):
                    p2.run(di.data_ptr(), npasses, pl.tune_count, da.data_ptr(), dsm.data_ptr())
                L.rxgpu_sync()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    p2.run(di.data_ptr(), npasses, pl.tune_count, da.data_ptr(), dsm.data_ptr())
                L.rxgpu_sync()
                t2 = (time.perf_counter() - t0) / reps
                in_samples = npasses * pl.tune_count * (pl.buf_len // 2)
                more[label] = {"Mbins/s": in_samples / pl.downsample / t2 / 1e6, "input MSample/s": in_samples / t2 / 1e6,
                               "GB/s_in": 4.0 * in_samples / t2 / 1e9, "frac_of_hbm_peak": 4.0 * in_samples / t2 / 1e9 / HBM_PEAK_GBS,
                               "N": nn, "downsample": pl.downsample, "tunes": pl.tune_count, "passes": npasses, "ms": t2 * 1e3}
                # which roofline binds this leg: every kernel of the launch counted (rocprofv3 --pmc, profiles/rNN_pmc_power_legs.json), the live time
                # of the whole launch -- HBM traffic (fetched x2 + written) against 8 TB/s, VALU wave-instructions against the measured issue ceiling
                cnt = next((v for k, v in pw_legs_pmc.items() if label.startswith(k)), None)
                if cnt is None and nn == 4096 and pl.tune_count == total_tunes:
                    try:                                              # the configs[2] kernel on other data: the same instructions and bytes per launch
                        k4 = pmc_kernel(pmc, "k_pw_fft4096", "SQ_INSTS_VALU")
                        cnt = {"passes": 512, "valu_wave_instr_per_launch": k4["SQ_INSTS_VALU"], "hbm_bytes_per_launch": k4.get("hbm_bytes_per_launch") or 0.0}
                    except (KeyError, TypeError):
                        cnt = None
                if cnt and cnt.get("passes"):
                    sc = npasses / float(cnt["passes"])
                    tr, vi = cnt["hbm_bytes_per_launch"] * sc / t2 / 1e9, cnt["valu_wave_instr_per_launch"] * sc / t2 / 1e9
                    more[label].update({"traffic_GBs": tr, "traffic_frac_of_hbm_peak": tr / HBM_PEAK_GBS, "traffic_over_input_bytes": cnt["hbm_bytes_per_launch"] * sc / (4.0 * in_samples),
                                        "valu_G_wave_instr_per_s": vi, "valu_frac": vi / fft_peak,
                                        "bound": "hbm" if tr / HBM_PEAK_GBS >= vi / fft_peak else "valu", "frac": max(tr / HBM_PEAK_GBS, vi / fft_peak),
                                        "roofline_source": "profiles/rNN_pmc_power_legs.json or, N = 4096, rNN_pmc_summary.json of the newest round (all kernels of the launch) / live launch time"})
                if not args.no_parity:
                    da.zero_()
                    dsm.zero_()
                    p2.run(di.data_ptr(), npasses, pl.tune_count, da.data_ptr(), dsm.data_ptr())
                    R.check(L.rxgpu_sync())
                    more[label]["parity"] = PA.power_check(rng, 0.0, window, boxcar, fir, 0, di.cpu().numpy(), da.cpu().numpy(), dsm.cpu().numpy())
                    pw_parity_ok = pw_parity_ok and more[label]["parity"]["parity_ok"]
                p2.close()
                del di, da, dsm
        pw["other_geometries"] = more
        if world == 1 and not args.no_parity:
            pw["parity_ok"] = bool(pw_parity_ok)
            parity_all["rx_power"] = bool(pw_parity_ok)
        if rank == 0 and args.cpu_seconds > 0:
            pw["cpu_baseline"] = cpu_baseline_power(plan, args.cpu_seconds / 2)
            pw["cpu_baseline"]["reference_default_build_O0"] = cpu_subprocess("power", min(3.0, args.cpu_seconds / 2), "libref_power_O0.so")
            pw["cpu_baseline"]["all_cores"] = cpu_all_cores("power", min(3.0, args.cpu_seconds / 2))
        if comm is not None:
            comm.close()
        if args.workload == "rx_power":
            result.update(pw)
            result.update({"steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
                           "data": "synthetic"})
        else:
            result["rx_power"] = pw

    # ------------------------------------------------------------------ channeliser (extension, configs[4])
    if args.workload in ("both", "chan"):
        block_len, bin_e, n_ch = 2 * 131072, 10, 256
        n_blocks = max(8, args.blocks // 8)                       # 2048 blocks = 1 GiB per step by default
        T = n_blocks * (block_len // 2)
        d_iq = device_capture(torch, dev, T, seed=4242 + rank, amp=600.0)
        windows = T >> bin_e
        d_out = torch.zeros((n_ch, windows), dtype=torch.int16, device=dev)
        ch = R.Channeliser(R.ChanParams(bin_e, 384, n_ch, 1), n_blocks, block_len, R.sine_table(bin_e))
        steps = max(20, args.steps)
        # 0.6 ms per run: the clocks of an idle part take some ten milliseconds of load to settle (the first launches after a pause measured 10 % slower)
        for _ in range(max(40, args.warmup)):
            ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        L.rxgpu_prof_reset()
        L.rxgpu_prof_enable(args.prof_level)
        barrier()
        import gc
        gc.collect()
        gc.disable()                                              # a generation-2 collection over this process's heap costs milliseconds: not inside 20 runs of 0.6 ms
        # the call-by-call form first (every run waited for: what a caller that needs each run's carries on the host pays) ...
        step_ms = []
        for _ in range(steps):
            t1 = time.perf_counter()
            ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)      # synchronous
            step_ms.append((time.perf_counter() - t1) * 1e3)
        L.rxgpu_prof_reset()
        barrier()
        # ... then the timed loop: runs enqueued back to back (two in flight, carries chained on the device, the host's flag check of run r under
        # run r + 1), one wait at the end -- like the rx_fm headline's pipelined loop
        t0 = time.perf_counter()
        for _ in range(steps):
            ch.run_async(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        ch.wait()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        gc.enable()
        L.rxgpu_prof_enable(0)
        ms, launches = prof("ch_fft")
        chan_fix = ch.host_fixups
        chan_parity = None
        if rank == 0 and not args.no_parity:
            # one more run of the timed shape from zero carries: every window of every channel against the oracle's channeliser
            PA = parity_module()
            ch.set_carry(np.zeros(2 * n_ch, np.int32))
            ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
            chan_parity = PA.chan_check(d_iq.cpu().numpy(), n_blocks, block_len, bin_e, 384, n_ch, 1, R.sine_table(bin_e),
                                        d_out.cpu().numpy(), ch.get_carry())
            parity_all["channeliser"] = bool(chan_parity["parity_ok"])
        ch.close()
        achieved = (4.0 * T) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        pmc, pmc_src = pmc_summary()
        ch_traffic = ch_valu = None
        try:
            k = pmc_kernel(pmc, "k_ch_fft", "SQ_INSTS_VALU")
            ch_valu = k["SQ_INSTS_VALU"] * (T / float(2048 * 131072)) / (ms / launches * 1e-3) / 1e9
            if k.get("hbm_bytes_per_launch"):
                ch_traffic = k["hbm_bytes_per_launch"] * (T / float(2048 * 131072))
        except (KeyError, TypeError, ZeroDivisionError):
            pass
        ch_peak, ch_peak_src = valu_ceiling("k_ch_fftR")
        result["channeliser"] = {
            "metric": "256-channel NBFM channeliser, capture MSample/s (extension: fix_fft per 1024-sample window + fm_demod per channel)",
            "value": world * T * steps / dt / 1e6, "unit": "MSample/s", "n_gpus": world, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "call_by_call_step_ms_min_median_max": [min(step_ms), sorted(step_ms)[len(step_ms) // 2], max(step_ms)],
            "timed_loop": "rxgpu_chan_run_async x steps, one rxgpu_chan_wait", "dtype": "int16/int32",
            "config": {"workload": "BASELINE configs[4]: 256 channels x 19.5 kHz from one 20 Msps capture, N=1024, -A fast",
                       "blocks_per_step": n_blocks, "parallelism": "replicas x%d" % world, "host_fixups_timed_loop_tail": int(chan_fix)},
            "roofline": {"bound": "hbm", "kernel": "k_ch_fft", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": ch_traffic, "algorithmic_bytes_per_launch": 4 * T,
                         "avg_launch_ms": (ms / launches) if launches else None,
                         "valu": {"achieved": ch_valu, "peak": ch_peak, "unit": "G wave-instr/s", "frac": (ch_valu / ch_peak) if ch_valu else None,
                                  "peak_source": ch_peak_src, "valu_source": (pmc_src + " (SQ_INSTS_VALU per launch) / live launch time") if ch_valu else None},
                         "note": "integer-VALU bound like k_pw_fft (register-blocked radix-16 passes, packed butterfly): the binding roofline is `valu`"},
        }
        if chan_parity is not None:
            result["channeliser"]["parity"] = chan_parity
        # What NBFM defaults to (rtl_fm.c:1086-1099: custom_atan = 0 unless -A fast -> libm atan2 per demodulated sample, rtl_fm.c:476-483) and the
        # per-channel audio stages (deemph_filter + low_pass_real, every channel its own carried state): the same capture, the same launch shape
        if rank == 0 and args.variants == "all":
            for label, prm, audio in (("-A std (NBFM default)", R.ChanParams(bin_e, 384, n_ch, 0), False),
                                      ("deemph + low_pass_real", R.ChanParams(bin_e, 384, n_ch, 1, 1, 7, 19531, 8000, 0), True)):
                ch3 = R.Channeliser(prm, n_blocks, block_len, R.sine_table(bin_e))
                for _ in range(5):
                    ch3.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
                L.rxgpu_prof_reset()
                L.rxgpu_prof_enable(2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fix3 = 0
                for _ in range(10):
                    ch3.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
                    fix3 += ch3.host_fixups
                torch.cuda.synchronize()
                t3 = (time.perf_counter() - t0) / 10
                L.rxgpu_prof_enable(0)
                stage = {}
                for nm in ("ch_fft", "ch_demod", "ch_audio"):
                    sms, sn = prof(nm)
                    if sn:
                        stage[nm] = round(sms / sn * 1e3, 1)
                leg = {"value": T / t3 / 1e6, "unit": "MSample/s", "ms": t3 * 1e3, "stage_us": stage, "frac_of_hbm_peak": 4.0 * T / t3 / 1e9 / HBM_PEAK_GBS,
                       "host_fixups_per_run": fix3 / 10.0, "demodulated_samples_per_run": int(n_ch * windows)}
                leg["valu_frac"], leg["valu_frac_source"] = chan_mode_valu_frac("audio" if audio else "std", T, t3)
                if not args.no_parity:
                    PA = parity_module()
                    if not audio:
                        ch3.set_carry(np.zeros(2 * n_ch, np.int32))
                        ch3.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
                        leg["parity"] = PA.chan_check(d_iq.cpu().numpy(), n_blocks, block_len, bin_e, 384, n_ch, 0, R.sine_table(bin_e), d_out.cpu().numpy(), ch3.get_carry())
                    else:
                        # the audio carries run through the whole stream of a channel (no block ranges to deal out): the first 16 callback blocks of every
                        # channel, from zero carries, against the reference's own fix_fft + full_demod (deemph_filter, low_pass_real) per channel and block
                        nba = 16
                        ch3.set_carry(np.zeros(2 * n_ch, np.int32))
                        ch3.set_audio_carry(np.zeros(3 * n_ch, np.int32))
                        k3 = ch3.run(d_iq.data_ptr(), nba, block_len, d_out.data_ptr(), windows)
                        h3 = d_iq[: nba * block_len].cpu().numpy()
                        stream3 = PA.support.ref_chan_stream if PA.support.have_ref() else PA.support.oracle_chan_stream
                        want3, pre3, st3 = stream3(h3, block_len, bin_e, 384, n_ch, 1, 1, 7, 19531, 8000)
                        same3 = bool(k3 == want3.shape[1] and np.array_equal(d_out[:, :k3].cpu().numpy(), want3) and np.array_equal(ch3.get_carry(), pre3)
                                     and np.array_equal(ch3.get_audio_carry().reshape(n_ch, 3), np.asarray(st3).reshape(n_ch, 3)))
                        leg["parity"] = {"parity_ok": same3, "parity_checker": "reference" if PA.support.have_ref() else "port", "parity_blocks": nba,
                                         "parity_channels": int(n_ch), "parity_audio_samples_per_channel": int(k3)}
                    parity_all["channeliser " + label] = bool(leg["parity"]["parity_ok"])
                ch3.close()
                result["channeliser"].setdefault("other_modes", {})[label] = leg
        # SURVEY 8(f)2's literal definition as the second mode (rxgpu_chan_params.nco: callback scale -> integer NCO per channel -> low_pass at
        # downsample N): the same 256 channels in another fixed-point rounding, ~50 times the arithmetic of the bank -- timed on 1/8 of the capture
        if rank == 0 and args.variants == "all":
            nb2 = max(2, n_blocks // 8)
            T2 = nb2 * (block_len // 2)
            w2 = T2 >> bin_e
            d_o2 = torch.zeros((n_ch, w2), dtype=torch.int16, device=dev)
            ch2 = R.Channeliser(R.ChanParams(bin_e, 384, n_ch, 1, 0, 0, 0, -1, 1), nb2, block_len, R.sine_table(bin_e))
            ch2.run(d_iq.data_ptr(), nb2, block_len, d_o2.data_ptr(), w2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ch2.run(d_iq.data_ptr(), nb2, block_len, d_o2.data_ptr(), w2)
            torch.cuda.synchronize()
            t_nco = (time.perf_counter() - t0) / 3
            nco_f, nco_src = chan_mode_valu_frac("nco", T2, t_nco)
            nco = {"value": T2 / t_nco / 1e6, "unit": "MSample/s", "ms": t_nco * 1e3, "blocks": nb2, "valu_frac": nco_f, "valu_frac_source": nco_src,
                   "note": "rxgpu_chan_params.nco = 1; N^2-ish by definition (one multiply-accumulate per sample and channel), which is why the fix_fft bank is the default"}
            if not args.no_parity:
                PA = parity_module()
                nbc = 2                                            # two callback blocks of every channel against the checker (a block is 2.7e7 products per channel set)
                ch2.set_carry(np.zeros(2 * n_ch, np.int32))
                ch2.run(d_iq.data_ptr(), nbc, block_len, d_o2.data_ptr(), w2)
                h2 = d_iq[: nbc * block_len].cpu().numpy()
                wpb2 = block_len // 2 >> bin_e
                if PA.support.have_ref():
                    want2, pre2 = PA.support.ref_chan_nco_stream(h2, block_len, bin_e, 384, n_ch, 1)
                    kind2 = "reference (the reference's callback scale, Sinewave table, FIX_MPY, and full_demod at downsample N)"
                else:
                    want2, pre2 = PA.support.oracle_chan_nco_stream(h2, block_len, bin_e, 384, n_ch, 1)
                    kind2 = "port (rxo_chan_nco_block, pinned against the reference-built chain in tests/test_chan_oracle.py)"
                same2 = bool(np.array_equal(d_o2[:, : nbc * wpb2].cpu().numpy(), want2) and np.array_equal(ch2.get_carry(), pre2))
                nco["parity"] = {"parity_ok": same2, "parity_checker": kind2, "parity_windows_compared": int(nbc * wpb2), "parity_channels": int(n_ch)}
                parity_all["channeliser_nco"] = same2
            ch2.close()
            result["channeliser"]["nco_mode"] = nco
            del d_o2
        del d_iq, d_out

    # ------------------------------------------------------------------ rx_sdr -F conversions
    if args.workload in ("both", "sdr") and world == 1:
        n_elems = 1 << 28                                         # 1 GiB of CS16
        g = torch.Generator(device=dev).manual_seed(99)
        d16 = torch.randint(-32768, 32768, (2 * n_elems,), dtype=torch.int16, device=dev, generator=g)
        d12 = torch.randint(0, 256, (3 * n_elems,), dtype=torch.uint8, device=dev, generator=g)
        torch.cuda.synchronize()
        legs = {}
        sdr_same = True
        for fmt in ("CU8", "CS8", "CF32", "CS16"):
            conv = R.SDR_CONVERSIONS[fmt][0]
            src = d12 if fmt == "CS16" else d16
            out = R.sdr_convert(fmt, src)
            if not args.no_parity:
                # head and tail of the 2^28-element launch against the oracle's converters (every int16 value is covered by the tests)
                PA = parity_module()
                R.check(L.rxgpu_sync())
                m = 1 << 20
                in_per, out_per = (3, 2) if fmt == "CS16" else (2, 2)
                same = True
                for lo in (0, n_elems - m):
                    want = PA.support.oracle_sdr_convert(fmt, src[lo * in_per:(lo + m) * in_per].cpu().numpy())
                    have = out.view(-1)[lo * out_per:(lo + m) * out_per].cpu().numpy()
                    same = same and np.array_equal(have.view(np.uint8), want.view(np.uint8))
                sdr_same = sdr_same and same
            L.rxgpu_prof_reset()
            L.rxgpu_prof_enable(2)
            reps = 10
            for _ in range(reps):
                R.sdr_convert(fmt, src, out)
            R.check(L.rxgpu_sync())
            L.rxgpu_prof_enable(0)
            ms, launches = prof("sdr_convert")
            nbytes = L.rxgpu_sdr_in_bytes(conv, n_elems) + L.rxgpu_sdr_out_bytes(conv, n_elems)
            gbs = nbytes / (ms / launches * 1e-3) / 1e9 if launches else 0.0
            ceil_key = {"CU8": "shrink_2_1_GBs", "CS8": "shrink_2_1_GBs", "CF32": "expand_1_2_GBs", "CS16": "copy_1_1_GBs"}[fmt]
            legs[("CS12->" if fmt == "CS16" else "CS16->") + fmt] = {
                "MSample/s": n_elems / (ms / launches * 1e-3) / 1e6 if launches else 0.0,
                "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "bytes_per_element": nbytes / n_elems,
                "box_ceiling_GBs": box[ceil_key], "box_ceiling_shape": ceil_key, "frac_of_box_ceiling": gbs / box[ceil_key]}
            del out
        result["sdr_convert"] = {"metric": "rx_sdr -F output conversions, complex MSample/s and HBM GB/s (read + write), 2^28 elements per launch",
                                 "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "legs": legs, "box_ceilings": box}
        if not args.no_parity:
            result["sdr_convert"]["parity_ok"] = bool(sdr_same)
            parity_all["sdr_convert"] = bool(sdr_same)
        del d16, d12

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        hoist(result, parity_all, world)
        if parity_all:
            result["parity_all_legs"] = parity_all
            result["parity_ok"] = all(parity_all.values())
        full = json.dumps(result)
        full_path = None
        try:
            os.makedirs(os.path.dirname(args.full_out), exist_ok=True)
            with open(args.full_out, "w") as f:
                f.write(full + "\n")
            full_path = os.path.relpath(args.full_out, ROOT) if os.path.abspath(args.full_out).startswith(ROOT + os.sep) else os.path.abspath(args.full_out)
        except OSError as e:
            sys.stderr.write("bench.py: full record not written to %s: %s\n" % (args.full_out, e))
        sys.stderr.write("BENCH_FULL " + full + "\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(compact(result, full_path))                          # the ONE stdout line
        sys.stdout.flush()
        if result.get("parity_ok") is False:
            sys.stderr.write("bench.py: GPU output differs from the CPU reference at bench size: %s\n" % sorted(k for k, v in parity_all.items() if not v))
            sys.exit(3)


if __name__ == "__main__":
    main()
