#!/usr/bin/env python3
"""bench.py -- the rx_tools hot path on MI355X, measured the way BASELINE.json asks.

Headline (`value`): complex IQ MSample/s through the rx_fm callback pre-stage + full_demod()
chain at the "20 Msps" WBFM geometry of BASELINE config 2 (downsample=118 -> 170 ksps ->
32 ksps audio, -A fast, de-emphasis on), blocks of 131072 complex samples, input resident in
HBM.  One step = one rxgpu_fm_stream_run over --blocks blocks (default 16384 = 2^31 samples = 8 GiB of cs16).
The same JSON line carries, under "rx_power", FFT bins/s of the scanner() chain at the
config-3 geometry (-f 24M:1.7G:1k: 599 tunes x 16384 int16, N=4096), tunes sharded across the
ranks with one RCCL gather of the avg[] rows to rank 0 per step.

  python bench.py --gpus 1 --steps 100 --warmup 10      (the defaults; about 10 s incl. the CPU baselines)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

rx_fm does not shard (one stream, sequential carries): N>1 runs N independent replicas
("replicas only", weak scaling).  rx_power shards by tune (strong scaling of one sweep).
torch is used for device buffers and torch.distributed only; every sample is processed by
librxgpu.so through its C ABI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_fm(block_len, budget_s):
    """The reference's own callback + full_demod (oracle/_ref, gcc -O2) -- or, where that
    prebuilt object is absent, the oracle port -- on one host core, config-2 parameters."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    import rx_tools_amd as R
    n_buf = 16
    iq = R.synth.sig_fm(n_buf * block_len // 2, seed=12345)
    if support.have_ref():
        L = support.ref_fm()
        support.ref_fm_reset(L, downsample=118)
        scratch = np.zeros(block_len, np.int16)
        calls, t0 = 0, time.perf_counter()
        while True:
            L.ref_fm_run_blocks(support.ptr16(iq), n_buf, block_len, 64, support.ptr16(scratch), None, 0)
            calls += 64
            dt = time.perf_counter() - t0
            if dt >= budget_s:
                break
        kind = "reference"
    else:
        O = support.oracle()
        st = support.oracle_fm_state(downsample=118)
        out = np.zeros(n_buf * block_len // 2, np.int16)
        calls, t0 = 0, time.perf_counter()
        while True:
            O.rxo_fm_stream(C.byref(st), support.ptr16(iq), n_buf, block_len, support.ptr16(out), None)
            calls += n_buf
            dt = time.perf_counter() - t0
            if dt >= budget_s:
                break
        kind = "port"
    samples = calls * (block_len // 2)
    return {"value": samples / dt / 1e6, "unit": "MSample/s", "cores": 1, "kind": kind,
            "sample": "%d blocks of %d complex samples, rtlsdr_callback+full_demod, ds=118 wbfm, %.1f s on 1 thread (%s)"
                      % (calls, block_len // 2, dt, cpu_model())}


def cpu_baseline_power(plan, budget_s):
    """scanner()'s per-tune chain on one host core: the reference's own scanner() over all 599 tunes
    (oracle/_ref) where that prebuilt object exists, else the oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support
    import rx_tools_amd as R
    n = 1 << plan.bin_e
    if support.have_ref():
        P = support.ref_power()
        P.ref_power_set_flags(1, 0, 0)
        P.ref_power_scan_tuned.argtypes = [support.i16p, C.c_int]
        tunes = P.ref_power_setup(b"24M:1.7G:1k", 0.0, b"rectangle")
        data = R.synth.sig_noise(tunes * plan.buf_len, seed=777, amp=100)
        done, t0 = 0, time.perf_counter()
        while True:
            P.ref_power_scan_tuned(support.ptr16(data), 1)     # scanner() without retune()'s settle sleep
            done += tunes
            dt = time.perf_counter() - t0
            if dt >= budget_s:
                break
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
        done, t0 = 0, time.perf_counter()
        while True:
            for t in range(tunes):
                O.rxo_power_tune(C.byref(cfg), support.ptr16(data[t * plan.buf_len:(t + 1) * plan.buf_len]),
                                 support.ptr16(work), support.ptr64(avg), C.byref(smp))
            done += tunes
            dt = time.perf_counter() - t0
            if dt >= budget_s:
                break
        kind = "port"
    bins = done * (plan.buf_len // 2)
    return {"value": bins / dt / 1e6, "unit": "Mbins/s", "cores": 1, "kind": kind,
            "sample": "%d tune buffers of %d int16 (N=%d), scanner() per-tune chain, %.1f s on 1 thread (%s)"
                      % (done, plan.buf_len, n, dt, cpu_model())}


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
            total += json.loads(out.decode().strip().splitlines()[-1])["value"]
            ok += 1
        except (ValueError, IndexError, KeyError):
            pass
    return {"value": total, "cores": ok, "note": "%d concurrent single-thread replicas, %.0f s each" % (ok, budget_s)}


def cpu_worker(which, budget_s):
    if which == "fm":
        r = cpu_baseline_fm(2 * 131072, budget_s)
    else:
        import types
        r = cpu_baseline_power(types.SimpleNamespace(bin_e=12, buf_len=16384), budget_s)   # -f 24M:1.7G:1k geometry
    print(json.dumps(r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=16384, help="rx_fm blocks of 131072 complex samples per step (16384 = 8 GiB of cs16)")
    ap.add_argument("--passes", type=int, default=512, help="rx_power scanner() passes per step (one report interval)")
    ap.add_argument("--workload", default="both", choices=["both", "rx_fm", "rx_power", "chan", "sdr"])
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline budget per path (0 = skip)")
    ap.add_argument("--prof-level", type=int, default=1)
    ap.add_argument("--cpu-worker", default=None, choices=["fm", "power"], help=argparse.SUPPRESS)
    ap.add_argument("--variants", default="F", choices=["F", "all"], help="rx_fm side figures: the -F cascade (default) or also -M wbfm's ds=6")
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker, args.cpu_seconds)
        return

    import torch
    import torch.distributed as dist
    import rx_tools_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def prof(name):
        ms, n = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    result = {}

    # ------------------------------------------------------------------ rx_fm (headline)
    if args.workload in ("both", "rx_fm"):
        block_len = 2 * 131072
        n_blocks = args.blocks
        base = R.synth.sig_fm(8 * 131072, seed=12345 + rank)          # 8 blocks of signal (A)
        d_base = torch.from_numpy(base).to(dev)
        d_iq = d_base.repeat(n_blocks // 8 + 1)[: n_blocks * block_len].contiguous()
        del d_base
        T = n_blocks * (block_len // 2)
        d_out = torch.zeros(T // 118 + 64, dtype=torch.int16, device=dev)
        s = R.FmStream(R.FmParams.wbfm(downsample=118), n_blocks, block_len)
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
        # SURVEY 8(d) config 2 also names the -F variant; and the -M wbfm default decimation.  Same buffer, same
        # pipelined loop, a few steps each; reported beside the headline, not part of `value`.
        variants = {}
        if world == 1:
            # (the ds=6 variant launches the headline's own decimator kernel; it is off by default so that the rocprofv3
            # per-kernel averages of this command describe the headline launches only)
            todo = [("-F cascade, downsample_passes=7 (ds=128)", dict(downsample_passes=7), 128)]
            if args.variants == "all":
                todo.append(("-M wbfm default, downsample=6", dict(downsample=6), 6))
            for label, kw, per_in in todo:
                d_o = torch.zeros(T // per_in + 64, dtype=torch.int16, device=dev)
                sv = R.FmStream(R.FmParams.wbfm(**kw), n_blocks, block_len)
                for _ in range(3):
                    sv.run_async(d_iq.data_ptr(), n_blocks, block_len, d_o.data_ptr(), d_o.numel())
                sv.wait()
                k = max(5, args.steps // 5)
                tv = time.perf_counter()
                for _ in range(k):
                    sv.run_async(d_iq.data_ptr(), n_blocks, block_len, d_o.data_ptr(), d_o.numel())
                sv.wait()
                tv = time.perf_counter() - tv
                variants[label] = {"value": T * k / tv / 1e6, "unit": "MSample/s", "ms_per_step": tv / k * 1e3, "steps": k}
                sv.close()
                del d_o
        del d_iq
        torch.cuda.empty_cache()
        value = world * T * args.steps / dt / 1e6
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))["k_fm_decimate"]
            # measured on 2^30-sample launches (--blocks 8192); bytes scale with the launch
            traffic = pmc["hbm_bytes_per_launch"] * (T / float(1 << 30))
            traffic_src = "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per gfx950 note)"
        except (OSError, KeyError, ValueError):
            pass
        achieved = (4.0 * T) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        result.update({
            "metric": "rx_fm full_demod complex IQ MSample/s (20 Msps WBFM geometry, ds=118)",
            "value": value, "unit": "MSample/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/int32 (fp32 fma for the cs16 scale, fp64 for one atan2 per block)",
            "data": "synthetic",
            "config": {"workload": "rx_fm WBFM 20.06 Msps -> 170 ksps -> 32 ksps: callback scale+rotate, low_pass ds=118, "
                                   "polar_disc_fast, deemph a=13, low_pass_real (BASELINE configs[1])",
                       "blocks_per_step": n_blocks, "block_complex_samples": block_len // 2,
                       "bytes_per_step": 4 * T, "parallelism": "replicas x%d (rx_fm does not shard)" % world,
                       "host_fixups_last_step": fixups},
            "rx_fm_variants": variants,
            "roofline": {"bound": "hbm", "kernel": "k_fm_decimate (F0+F1+F2)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 4 * T, "avg_launch_ms": (ms / launches) if launches else None},
        })
        if rank == 0 and args.cpu_seconds > 0 and world == 1:
            result["cpu_baseline"] = cpu_baseline_fm(block_len, args.cpu_seconds)
            result["cpu_baseline"]["all_cores"] = cpu_all_cores("fm", min(4.0, args.cpu_seconds))

    # ------------------------------------------------------------------ rx_power
    if args.workload in ("both", "rx_power"):
        plan = R.plan_range("24M:1.7G:1k", 0.0, 1)
        n = 1 << plan.bin_e
        total_tunes = plan.tune_count
        from rx_tools_amd import shard
        lo, mine, per = shard.tune_range(rank, world, total_tunes)   # contiguous tune ranges, SURVEY section 8(e)
        passes = args.passes
        wc, sw = R.window_coefs("rectangle", n), R.sine_table(plan.bin_e)
        ps = R.PowerScan(R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0),
                         per, wc, sw)
        g = torch.Generator(device=dev)
        g.manual_seed(777 + rank)
        d_in = torch.randint(-100, 101, (passes, max(mine, 1), plan.buf_len), dtype=torch.int16, device=dev, generator=g)
        # two report-interval buffers: the gather of interval k (RCCL, torch's stream) overlaps the scan of interval k+1
        d_avgs = [torch.zeros((per, n), dtype=torch.int64, device=dev) for _ in range(2)]   # padded to `per` rows for the gather
        d_smp = torch.zeros(per, dtype=torch.int32, device=dev)
        gbuf = shard.gather_buffers(d_avgs[0], dst=0) if world > 1 else None
        gathered = [None, None]          # event after the gather that last read buffer b
        state = {"k": 0}

        def step():
            b = state["k"] & 1
            state["k"] += 1
            if gathered[b] is not None:
                gathered[b].synchronize()
            if mine:
                ps.run(d_in.data_ptr(), passes, mine, d_avgs[b].data_ptr(), d_smp.data_ptr())
            if world > 1:
                # order the gather (torch's stream) after the scan (librxgpu's stream)
                L.rxgpu_sync()
                shard.gather_rows(d_avgs[b], dst=0, out=gbuf)
                ev = torch.cuda.Event()
                ev.record()
                gathered[b] = ev

        for _ in range(args.warmup):
            step()
        L.rxgpu_prof_reset()
        L.rxgpu_prof_enable(args.prof_level)
        barrier()
        L.rxgpu_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        L.rxgpu_sync()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        L.rxgpu_prof_enable(0)
        ms, launches = prof("pw_fft")
        bins_per_step_all = passes * total_tunes * (plan.buf_len // 2)
        bins_local = passes * mine * (plan.buf_len // 2)
        achieved = (4.0 * bins_local) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        pw = {
            "metric": "rx_power FFT bins/s (scanner() chain, -f 24M:1.7G:1k geometry)",
            "value": bins_per_step_all * args.steps / dt / 1e6, "unit": "Mbins/s", "n_gpus": world,
            "ms_per_step": dt / args.steps * 1e3, "scaling": "strong", "dtype": "int16/int32/int64",
            "config": {"workload": "599 tunes x 16384 int16, N=4096, 2 FFT blocks/tune/pass, rectangle window (BASELINE configs[2]/[3])",
                       "passes_per_step": passes, "tunes_this_rank": mine,
                       "parallelism": "tunes sharded x%d, one RCCL gather of avg[] to rank 0 per step" % world},
            "roofline": {"bound": "hbm", "kernel": "k_pw_fft (P4-P8)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": 4 * bins_local,
                         "avg_launch_ms": (ms / launches) if launches else None,
                         "note": "integer-VALU/LDS bound on paper (SURVEY section 8d); HBM fraction reported as asked"},
        }
        if rank == 0 and args.cpu_seconds > 0 and world == 1:
            pw["cpu_baseline"] = cpu_baseline_power(plan, args.cpu_seconds / 2)
            pw["cpu_baseline"]["all_cores"] = cpu_all_cores("power", min(4.0, args.cpu_seconds / 2))
        ps.close()
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
        base = R.synth.sig_fm(8 * 131072, seed=4242 + rank, amp=600.0)
        d_iq = torch.from_numpy(base).to(dev).repeat(n_blocks // 8 + 1)[: n_blocks * block_len].contiguous()
        T = n_blocks * (block_len // 2)
        windows = T >> bin_e
        d_out = torch.zeros((n_ch, windows), dtype=torch.int16, device=dev)
        ch = R.Channeliser(R.ChanParams(bin_e, 384, n_ch, 1), n_blocks, block_len, R.sine_table(bin_e))
        steps = max(5, args.steps // 5)
        for _ in range(max(1, args.warmup // 3)):
            ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        L.rxgpu_prof_reset()
        L.rxgpu_prof_enable(args.prof_level)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        L.rxgpu_prof_enable(0)
        ms, launches = prof("ch_fft")
        ch.close()
        achieved = (4.0 * T) / (ms / launches * 1e-3) / 1e9 if launches else 0.0
        result["channeliser"] = {
            "metric": "256-channel NBFM channeliser, capture MSample/s (extension: fix_fft per 1024-sample window + fm_demod per channel)",
            "value": world * T * steps / dt / 1e6, "unit": "MSample/s", "n_gpus": world, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "dtype": "int16/int32",
            "config": {"workload": "BASELINE configs[4]: 256 channels x 19.5 kHz from one 20 Msps capture, N=1024, -A fast",
                       "blocks_per_step": n_blocks, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_ch_fft", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": 4 * T,
                         "avg_launch_ms": (ms / launches) if launches else None,
                         "note": "integer-VALU bound like k_pw_fft (register-blocked radix-16 passes, packed butterfly)"},
        }
        del d_iq, d_out

    # ------------------------------------------------------------------ rx_sdr -F conversions
    if args.workload in ("both", "sdr") and world == 1:
        n_elems = 1 << 28                                         # 1 GiB of CS16
        g = torch.Generator(device=dev).manual_seed(99)
        d16 = torch.randint(-32768, 32768, (2 * n_elems,), dtype=torch.int16, device=dev, generator=g)
        d12 = torch.randint(0, 256, (3 * n_elems,), dtype=torch.uint8, device=dev, generator=g)
        torch.cuda.synchronize()
        legs = {}
        for fmt in ("CU8", "CS8", "CF32", "CS16"):
            conv = R.SDR_CONVERSIONS[fmt][0]
            src = d12 if fmt == "CS16" else d16
            out = R.sdr_convert(fmt, src)
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
            legs[("CS12->" if fmt == "CS16" else "CS16->") + fmt] = {
                "MSample/s": n_elems / (ms / launches * 1e-3) / 1e6 if launches else 0.0,
                "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "bytes_per_element": nbytes / n_elems}
            del out
        result["sdr_convert"] = {"metric": "rx_sdr -F output conversions, complex MSample/s and HBM GB/s (read + write), 2^28 elements per launch",
                                 "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "legs": legs}
        del d16, d12

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
