"""CPU: host-side logic of librxgpu (range planner, tables, csv_dbm) against golden vectors
generated from the reference and against the oracle."""
import ast
import ctypes as C
import os

import numpy as np

import rx_tools_amd as R
from rx_tools_amd.structs import TuningState
from support import GOLDEN_DIR, oracle, ptr16, ptr32, ptr64


def test_plan_range_matches_reference_frequency_range():
    rows = np.load(os.path.join(GOLDEN_DIR, "plans.npz"))["rows"]
    assert len(rows) >= 8
    for r in rows:
        rng, crop, boxcar, n, bin_e, buf_len, ds, ds_p, rate, f0, df, crop_out = ast.literal_eval(str(r))
        p = R.plan_range(rng, crop, boxcar)
        assert (p.tune_count, p.bin_e, p.buf_len, p.downsample, p.downsample_passes, p.rate) == (n, bin_e, buf_len, ds, ds_p, rate), rng
        assert p.first_freq == f0 and (n == 1 or p.bw_seen == df) and p.crop == crop_out, rng


def test_config3_geometry():
    p = R.plan_range("24M:1.7G:1k")
    assert (p.tune_count, p.rate, p.bin_e, p.downsample, p.buf_len) == (599, 2797996, 12, 1, 16384)


def test_tables_match_oracle():
    O = oracle()
    for e in (0, 1, 3, 5, 12, 14, 17):
        n = 1 << e
        want = np.zeros(max(1, n * 3 // 4), np.int16)
        O.rxo_sine_table(e, ptr16(want))
        assert np.array_equal(R.sine_table(e)[:n * 3 // 4], want[:n * 3 // 4])
    for w in ("rectangle", "hamming", "blackman", "blackman-harris", "hann-poisson", "youssef", "kaiser", "bartlett"):
        for n in (2, 32, 4096):
            want = np.zeros(n, np.int32)
            O.rxo_window_coefs(w.encode(), n, ptr32(want))
            assert np.array_equal(R.window_coefs(w, n), want), (w, n)


def test_csv_dbm_matches_reference_rows(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from gen_golden import POWER_CASES
    z = np.load(os.path.join(GOLDEN_DIR, "power_cases.npz"))
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fclose.argtypes = [C.c_void_p]
    for name, *_ in POWER_CASES:
        bin_e, buf_len, ds, ds_p, rate, _n = map(int, z[name + "__meta"])
        avg = z[name + "__avg"].copy()
        path = str(tmp_path / (name + ".csv"))
        f = libc.fopen(path.encode(), b"wb")
        for t in range(avg.shape[0]):
            ts = TuningState(int(z[name + "__freqs"][t]), rate, bin_e, ptr64(avg[t]), int(z[name + "__samples"][t]), ds, ds_p,
                             float(z[name + "__crop"][0]), None, buf_len)
            R.lib().rxgpu_csv_dbm(C.byref(ts), f)
            assert ts.samples == 0 and not avg[t].any()
        libc.fclose(f)
        assert open(path).read().splitlines() == [str(r) for r in z[name + "__csv"]], name


def _atofs(s):
    mult = {"k": 1e3, "K": 1e3, "M": 1e6, "m": 1e6, "G": 1e9, "g": 1e9}.get(s[-1])
    return float(s[:-1]) * mult if mult else float(s)


def test_fm_plan_matches_reference_main():
    """rxgpu_fm_params_init + rxgpu_fm_plan_settings == the state rx_fm's own main() derives for the same flags
    (golden: the reference's globals after its main() ran, oracle/gen_golden.py fm_plans)"""
    import json
    rows = np.load(os.path.join(GOLDEN_DIR, "fm_plans.npz"))["rows"]
    assert len(rows) >= 12
    mode_fn = {R.fm.FmParams.for_mode(m)[0].mode: fn for m, fn in (("fm", 0), ("raw", 1), ("am", 2), ("usb", 3), ("lsb", 4))}
    for r in rows:
        row = json.loads(str(r))
        a, st = row["args"], row["state"]
        opt = {a[i]: [] for i in range(0, len(a), 2)}
        for i in range(0, len(a), 2):
            opt[a[i]].append(a[i + 1])
        mode = opt["-M"][0]
        kw = {}
        for e in opt.get("-E", []):
            kw.update({"rdc": dict(dc_block_raw=1), "deemp": dict(deemph=1), "offset": dict(offset_tuning=1), "edge": {},
                       "adc": dict(dc_block_audio=1)}[e])
        if "-o" in opt:
            kw["post_downsample"] = int(opt["-o"][0])
        if "-r" in opt:
            kw["rate_out2"] = int(_atofs(opt["-r"][0]))
        if "-A" in opt:
            kw["custom_atan"] = {"std": 0, "fast": 1, "lut": 2, "ale": 3}[opt["-A"][0]]
        tc = {"us": 75, "eu": 50}.get(opt.get("-c", ["us"])[0]) or int(float(opt["-c"][0]))
        freq = int(_atofs(opt["-f"][0])) + (16000 if mode in ("wbfm", "wfm") else 0)     # controller_thread_fn, rtl_fm.c:1006-1009
        p, plan = R.fm.FmParams.for_mode(mode, freq=freq, rate_in=int(_atofs(opt["-s"][0])) if "-s" in opt else None,
                                          fifth_order=int(opt["-F"][0]) if "-F" in opt else None,
                                          edge=1 if "edge" in opt.get("-E", []) else 0, time_constant_us=tc, **kw)
        got = dict(rate_in=plan.rate_in, rate_out=p.rate_out, rate_out2=p.rate_out2, downsample=p.downsample,
                   post_downsample=p.post_downsample, output_scale=p.output_scale, downsample_passes=p.downsample_passes,
                   comp_fir_size=p.comp_fir_size, custom_atan=p.custom_atan, deemph=p.deemph, deemph_a=p.deemph_a,
                   squelch_level=p.squelch_level, dc_block_audio=p.dc_block_audio, dc_block_raw=p.dc_block_raw,
                   adc_block_const=p.adc_block_const, rdc_block_const=p.rdc_block_const, mode_fn=mode_fn[p.mode],
                   capture_freq=plan.capture_freq, capture_rate=plan.capture_rate, offset_tuning=p.offset_tuning)
        assert got == st, (a, {k: (got[k], st[k]) for k in st if got[k] != st[k]})


def test_wbfm_defaults_are_derived():
    p = R.FmParams.wbfm()
    assert (p.downsample, p.downsample_passes, p.deemph, p.deemph_a, p.rate_out, p.rate_out2, p.custom_atan, p.output_scale,
            p.post_downsample, p.adc_block_const, p.rdc_block_const) == (6, 0, 1, 13, 170000, 32000, 1, 1, 1, 9, 9)
    assert R.FmParams.wbfm(downsample=118).downsample == 118


def test_shard_tunes_cover_every_tune_once():
    from rx_tools_amd import shard
    for total in (1, 7, 599, 600, 4096):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, cnt, per = shard.tune_range(r, world, total)
                assert per == -(-total // world) and cnt <= per
                seen += list(range(lo, lo + cnt))
            assert seen == list(range(total))


def test_profiles_hold_what_bench_reads():
    """bench.py takes the decimator's HBM traffic and rx_power's VALU instruction count from the committed PMC summary
    (profiles/rNN_pmc_summary.json, made by tools/collect_profiles.py): the kernels' template arguments are part of the
    keys, so a renamed instantiation would silently turn `roofline.traffic` into null"""
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pmc = json.load(open(os.path.join(root, "profiles", "r02_pmc_summary.json")))
    dec = [v for k, v in pmc.items() if k.startswith("k_fm_decimate<false, true, true") and isinstance(v, dict) and v.get("hbm_bytes_per_launch")]
    fft = [v for k, v in pmc.items() if k.startswith("k_pw_fft4096") and isinstance(v, dict) and v.get("SQ_INSTS_VALU")]
    assert dec and fft
    # 4 GiB launches: fetched + written bytes within 2 % of the 4 B per sample the path needs
    assert abs(dec[0]["hbm_bytes_per_launch"] / (4.0 * (1 << 30)) - 1.0) < 0.02


def test_newest_round_profiles_hold_what_bench_reads():
    """the tables behind bench.py's `roofline` objects, newest round first: the counter summary (the decimator's traffic, the FFT kernels'
    instruction counts, one traffic entry per timed rx_fm chain) and the measured VALU ceiling (per-opcode issue rates x the static
    opcode mix of the kernels tools/kernel_mix.py names) -- a renamed kernel instantiation must fail here, not turn a field into null"""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    pmc = json.load(open(bench.newest_profile("pmc_summary.json")))
    assert any(k.startswith("k_fm_decimate<false, true, true") and v.get("hbm_bytes_per_launch") for k, v in pmc.items() if isinstance(v, dict))
    assert any(k.startswith("k_pw_fft4096") and v.get("SQ_INSTS_VALU") for k, v in pmc.items() if isinstance(v, dict))
    assert any(k.startswith("k_ch_fft") and v.get("SQ_INSTS_VALU") for k, v in pmc.items() if isinstance(v, dict))
    chains = pmc["_chains"]
    assert len(chains) >= 7
    for name, c in chains.items():
        assert 1.0 <= c["traffic_over_algorithmic"] < 2.0, (name, c)
    # the whole-chain -F kernel took the 1/8-rate stream out of the `-M wbfm -F 9` chain
    assert [c for n, c in chains.items() if "-F 9" in n][0]["traffic_over_algorithmic"] < 1.35
    for sub in ("k_pw_fft4096", "k_ch_fftR", "k_pwm_tail"):
        peak, src = bench.valu_ceiling(sub)
        assert src is not None and 0.20 * bench.SIMDS * bench.CLOCK_GHZ < peak < 0.30 * bench.SIMDS * bench.CLOCK_GHZ
    sys.path.insert(0, os.path.join(root, "tools"))
    import kernel_mix
    mix = json.load(open(bench.newest_profile("kernel_mix.json")))
    for names in kernel_mix.WANT.values():
        for n in names:
            assert any(n in k for k in mix), n
    if bench.newest_profile("pmc_power_legs.json"):
        legs = json.load(open(bench.newest_profile("pmc_power_legs.json")))
        assert len(legs) == 4 and all(v["valu_wave_instr_per_launch"] > 0 and v["hbm_bytes_per_launch"] > 0 for v in legs.values())


def test_every_tuning_knob_the_sources_read_is_in_the_snapshot_table():
    """rxgpu_knob() aborts on a name the snapshot table (rxgpu_rt.c, g_knob_names) does not hold -- on the launch path of whoever asks: every
    name the library's sources pass to it has to be registered, and INTEGRATION.md has to list the ones a user may set"""
    import glob
    import re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(ROOT, "rx_tools_amd", "csrc")
    used = set()
    for f in glob.glob(os.path.join(csrc, "*")):
        if f.endswith((".c", ".hip", ".h", ".inc")):
            used |= set(re.findall(r'rxgpu_knob\("(\w+)"\)', open(f).read()))
    src = open(os.path.join(csrc, "rxgpu_rt.c")).read()
    tab = src[src.index("g_knob_names[] = {"):]
    table = set(re.findall(r'"(RXGPU_\w+)"', tab[:tab.index("};")]))
    assert used and used <= table, sorted(used - table)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    undocumented = sorted(k for k in used if k not in doc and not re.match(r"RXGPU_EXP\d$", k))
    assert not undocumented, undocumented


def test_bench_compact_line_stays_small_and_strict():
    """What bench.py prints on stdout is ONE compact line: the driver kept the 4-16 KB lines of rounds 1-3 and dropped round 4's 25 KB one
    (BENCH_r04.json: parsed null).  compact() over the largest full records committed so far must stay under 8 KB, be strict JSON (no NaN /
    Infinity, no object nested deeper than 3), and carry the fields the contract names."""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    def depth(o):
        return 1 + max((depth(v) for v in o.values()), default=0) if isinstance(o, dict) else 0

    for name in ("r04_bench_n1.json", "r04_bench_n1_b.json", "r03_bench_n1.json"):
        full = json.load(open(os.path.join(root, "profiles", name)))
        full["roofline"]["poison"] = float("nan")                 # a NaN anywhere in the full record must not reach the line
        full["value"] = float(full["value"])
        line = bench.compact(full, "gpurun_out/bench_full.json")
        assert "\n" not in line and len(line) < bench.COMPACT_LIMIT and len(line) < 8192, (name, len(line))
        assert "NaN" not in line and "Infinity" not in line
        back = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
        assert depth(back) <= 4                                    # line -> roofline -> legs -> one leg
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline", "cpu_baseline"):
            assert k in back, (name, k)
        assert back["config"]["workload"] and back["value"] == float("%.7g" % full["value"])
        r = back["roofline"]
        assert r["bound"] == "hbm" and abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["peak"] == 8000.0 and r["traffic"] > 0
        assert back["cpu_baseline"]["value"] > 0 and back["cpu_baseline"]["cores"] == 1 and back["cpu_baseline"]["kind"] == "reference"
        if "legs" in full["roofline"]:
            assert len(r["legs"]) == len(full["roofline"]["legs"]) and r["legs"]["fm_ds6"]["ok"] is True and 0.3 < r["legs"]["fm_ds6"]["f"] < 1.0
            assert all(set(v) <= {"b", "f", "t", "ok", "hbm_f"} for v in r["legs"].values())
            assert back["parity_ok"] is True and back["config"]["rccl_ranks"] == 1 and back["config"]["rx_power_Mbins_per_s"] > 0
    # an absurdly large leg table is dropped rather than allowed to push the line over the limit
    full["roofline"]["legs"] = {"leg %d" % i: {"bound": "hbm", "frac": 0.5, "parity_ok": True} for i in range(400)}
    line = bench.compact(full)
    assert len(line) < bench.COMPACT_LIMIT and "legs" not in json.loads(line)["roofline"]


def test_bench_self_launches_when_gpus_gt_1_without_torchrun(tmp_path):
    """`python bench.py --gpus N` (the shape of the driver's N=1 command) must not die on launch shape at N > 1: without WORLD_SIZE it re-executes
    itself under torch.distributed.run with one process per GPU.  Checked here by pointing the re-exec at a stand-in interpreter that records its argv."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "fakepython"
    fake.write_text("#!/bin/sh\nprintf '%s\\n' \"$@\" > " + str(tmp_path / "argv") + "\n")
    fake.chmod(0o755)
    code = ("import sys, os\nsys.path.insert(0, %r)\nimport bench\nsys.executable = %r\nsys.argv = ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1']\n"
            "os.environ.pop('WORLD_SIZE', None)\nbench.main()\n" % (root, str(fake)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    argv = (tmp_path / "argv").read_text().split("\n")
    assert argv[:3] == ["-m", "torch.distributed.run", "--nnodes=1"] and argv[3:5] == ["--nproc-per-node", "4"]
    assert "--master-addr" in argv and argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.join(root, "bench.py"))
    assert argv[i + 1:i + 7] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # under a launcher whose world size disagrees with --gpus it refuses instead of looping
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b"WORLD_SIZE=2" in out.stderr
