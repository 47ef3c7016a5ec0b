"""-m gpu: the rx_power HIP path against the oracle, through the C ABI (librxgpu.so)."""
import ctypes as C

import numpy as np
import pytest

import rx_tools_amd as R
from support import oracle, sig_noise, PowerCfg, ptr16, ptr32, ptr64, have_ref, ref_power_scan_first

pytestmark = pytest.mark.gpu


def oracle_scan(data, passes, tunes, plan, window, sinewave, boxcar, comp_fir, peak_hold):
    O = oracle()
    n = 1 << plan.bin_e
    wc = np.ascontiguousarray(window, np.int32)
    sw = np.ascontiguousarray(sinewave, np.int16)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, boxcar, comp_fir, peak_hold,
                   ptr32(wc), ptr16(sw))
    avg = np.zeros((tunes, n), np.int64)
    samples = np.zeros(tunes, np.int32)
    work = np.zeros(plan.buf_len, np.int16)
    d3 = data.reshape(passes, tunes, plan.buf_len)
    for p in range(passes):
        for t in range(tunes):
            s = C.c_int(int(samples[t]))
            O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(d3[p, t])), ptr16(work), ptr64(avg[t]), C.byref(s))
            samples[t] = s.value
    return avg, samples


def gpu_scan(data, passes, tunes, plan, window, sinewave, boxcar, comp_fir, peak_hold, avg0=None):
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    n = 1 << plan.bin_e
    p = R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, boxcar, comp_fir, peak_hold)
    s = R.PowerScan(p, tunes, window, sinewave)
    d_in = to_dev(data)
    d_avg = torch.zeros((tunes, n), dtype=torch.int64, device="cuda") if avg0 is None else to_dev(avg0)
    d_samples = torch.zeros(tunes, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                  # torch's fills run on ITS stream: they must have landed before the library's stream adds to these arrays
    s.run(d_in.data_ptr(), passes, tunes, d_avg.data_ptr(), d_samples.data_ptr())
    R.check(R.lib().rxgpu_sync())
    out = d_avg.cpu().numpy(), d_samples.cpu().numpy()
    s.close()
    return out


CASES = [
    # range, crop, window, (boxcar, comp_fir, peak_hold), amplitude, passes, max tunes
    ("24M:1.7G:1k", 0.0, "rectangle", (1, 0, 0), 100, 3, 7),        # config 3 geometry, no window wrap
    ("24M:1.7G:1k", 0.0, "rectangle", (1, 0, 0), 32768, 2, 5),      # full scale: wraps everywhere
    ("24M:1.7G:1k", 0.0, "hamming", (1, 0, 1), 3000, 3, 4),         # peak hold
    ("24M:1.7G:1k", 0.0, "hamming", (1, 0, 0), 3000, 2, 70),        # more than 64 tunes: straight atomics on avg[] (fewer: per-group partial spectra)
    ("24M:1.7G:1k", 0.0, "rectangle", (1, 0, 1), 20000, 40, 2),     # many passes of two tunes: several groups per tune through the partial buffer, peak hold
    ("88M:108M:125k", 0.0, "blackman-harris", (1, 0, 0), 5000, 2, 8),   # N=32
    ("100M:101M:1k", 0.2, "bartlett", (1, 0, 0), 500, 2, 1),        # N=1024
    ("100M:100.1M:10", 0.0, "rectangle", (1, 0, 0), 2000, 1, 1),    # N=16384, boxcar ds=28
    ("100M:100.1M:10", 0.0, "youssef", (0, 0, 0), 2000, 2, 1),      # fifth_order ds=16
    ("100M:100.1M:10", 0.0, "hann-poisson", (0, 9, 0), 30000, 1, 1),    # + droop FIR
    ("100M:100.1M:10", 0.0, "blackman", (1, 0, 1), 32768, 3, 1),    # N=16384 (two-launch register-blocked path), full scale, peak hold
    ("100M:100.2M:10", 0.0, "hamming", (1, 0, 0), 32768, 3, 1),     # N=32768, full scale
    ("100M:100.2M:10", 0.0, "rectangle", (1, 0, 1), 9000, 2, 1),    # N=32768, peak hold
    ("100M:100.3M:100", 0.0, "rectangle", (1, 0, 0), 9000, 2, 1),   # boxcar, odd ds
    ("100M:102M:5k", 0.0, "hamming", (1, 0, 0), 20000, 2, 1),       # N=512
    ("100M:102M:10k", 0.0, "rectangle", (1, 0, 1), 9000, 3, 1),     # N=256
    ("100M:102.5M:600", 0.0, "blackman", (1, 0, 0), 32768, 2, 1),   # N=8192
    ("100M:102M:2k", 0.0, "bartlett", (1, 0, 0), 12000, 2, 1),      # N=1024
    ("100M:102M:1500", 0.0, "rectangle", (1, 0, 0), 12000, 2, 1),   # N=2048
    # N > 2^15: radix-16 passes through a scratch copy in HBM until a sub-transform fits a workgroup (one head pass up to 2^16, two
    # up to 2^20, three for 2^21), then the register-blocked tail
    ("100M:102M:40", 0.0, "hamming", (1, 0, 0), 20000, 2, 1),       # N=2^16
    ("100M:102M:20", 0.0, "blackman", (1, 0, 0), 32768, 3, 1),      # N=2^17, full scale, several passes (the per-group partial spectra)
    ("100M:102.8M:20", 0.0, "rectangle", (1, 0, 1), 32768, 2, 1),   # N=2^18, peak hold, full scale
    ("100M:102.8M:10", 0.0, "hamming", (1, 0, 0), 12000, 2, 1),     # N=2^19
    ("100M:102.8M:5", 0.0, "rectangle", (1, 0, 1), 32768, 2, 1),    # N=2^20, peak hold
    ("100M:102.8M:2", 0.0, "blackman", (1, 0, 0), 3000, 1, 1),      # N=2^21, the reference's largest
    ("100M:110M:1M", 0.0, "rectangle", (1, 0, 0), 5000, 3, 10),     # rms_power path
    ("100M:110M:1M", 0.0, "rectangle", (1, 0, 1), 5000, 3, 10),
]


@pytest.mark.parametrize("rng,crop,window,flags,amp,passes,max_tunes", CASES)
def test_scan_bit_exact(rng, crop, window, flags, amp, passes, max_tunes):
    plan = R.plan_range(rng, crop, flags[0])
    tunes = min(plan.tune_count, max_tunes)
    n = 1 << plan.bin_e
    wc = R.window_coefs(window, n)
    sw = R.sine_table(plan.bin_e)
    data = sig_noise(passes * tunes * plan.buf_len, seed=777, amp=amp)
    want_avg, want_samples = oracle_scan(data, passes, tunes, plan, wc, sw, *flags)
    if have_ref() and passes * plan.tune_count <= 5000:          # (the reference runs every tune of the sweep, the ones past max_tunes on zeros)
        # oracle/_ref travelled with the tree: the expected values are held to the reference's own scanner() on this input first
        ref_avg, ref_samples = ref_power_scan_first(rng, crop, window, flags, data, passes, tunes)
        assert np.array_equal(ref_avg, want_avg) and np.array_equal(ref_samples, want_samples), "oracle != reference"
    got_avg, got_samples = gpu_scan(data, passes, tunes, plan, wc, sw, *flags)
    assert np.array_equal(got_samples, want_samples)
    bad = np.argwhere(got_avg != want_avg)
    assert bad.size == 0, "first mismatch at %s: got %d want %d (%d bad)" % (
        bad[0], got_avg[tuple(bad[0])], want_avg[tuple(bad[0])], len(bad))


def test_scan_linearity_in_passes_full_geometry():
    """size-independent property at config-3 size: avg over 2P identical passes == 2 x avg over P,
    all 599 tunes"""
    plan = R.plan_range("24M:1.7G:1k", 0.0, 1)
    n = 1 << plan.bin_e
    wc = R.window_coefs("rectangle", n)
    sw = R.sine_table(plan.bin_e)
    one = sig_noise(plan.tune_count * plan.buf_len, seed=31, amp=800)
    a1, s1 = gpu_scan(one, 1, plan.tune_count, plan, wc, sw, 1, 0, 0)
    a4, s4 = gpu_scan(np.tile(one, 4), 4, plan.tune_count, plan, wc, sw, 1, 0, 0)
    assert np.array_equal(a4, 4 * a1)
    assert np.array_equal(s4, 4 * s1)
    # and a few tunes against the oracle
    want, _ = oracle_scan(one[:3 * plan.buf_len], 1, 3, plan, wc, sw, 1, 0, 0)
    assert np.array_equal(a1[:3], want)


def _run_rccl_world(world, tmp_path, env=None):
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "rccl_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, **(env or {})))
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert (tmp_path / "ok").exists()
    return (tmp_path / "ok").read_text()


@pytest.mark.timeout(900)
def test_sharded_sweep_through_librxgpu_rccl_world1(tmp_path):
    """product scan -> ncclGather issued by librxgpu itself (rxgpu_power_scan_run_sharded, librccl bound at run time)
    -> rxgpu_csv_dbm on the root == the oracle's CSV.  One rank: what a 1-GPU box can run."""
    assert "world=1" in _run_rccl_world(1, tmp_path)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_sweep_through_librxgpu_rccl_multi_gpu(world, tmp_path):
    """the same with one process per GPU over xGMI; needs that many GPUs"""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    assert ("world=%d" % world) in _run_rccl_world(world, tmp_path)


def _fake_rccl():
    """tests/fake_rccl.c built on demand: a file-based transport that lets several ranks share the one GPU of the test box"""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "libfake_rccl.so"), os.path.join(here, "fake_rccl.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so,
                               "-L/opt/rocm/lib", "-lamdhip64"])
    return so


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_sweep_multi_rank_on_one_gpu(world, tmp_path):
    """The product's multi-rank path with 2, 4, 8 PROCESSES on this box's one GPU: every rank creates its communicator through
    rxgpu_comm_create, scans its contiguous tune range with rxgpu_power_scan_run_sharded (padding rows zeroed, grouped gather),
    the root merges the rows and prints the CSV == the oracle's single-process sweep.  RCCL itself refuses two ranks per device,
    so $RXGPU_RCCL_LIB points librxgpu at tests/fake_rccl.c (a file transport); everything above the transport is the product."""
    env = {"RXGPU_RCCL_LIB": _fake_rccl(), "FAKE_RCCL_DIR": str(tmp_path), "RCCL_WORKER_ONE_GPU": "1", "RCCL_WORKER_DIRTY_PADDING": "1"}
    ok = _run_rccl_world(world, tmp_path, env)
    assert ("world=%d" % world) in ok and "libfake_rccl" in ok


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,form", [(2, "torchrun"), (4, "plain")])
def test_bench_multi_rank_path_on_one_gpu(world, form, tmp_path):
    """bench.py --gpus N as the driver launches it -- under torch.distributed.run ("torchrun"), and as the bare `python bench.py --gpus N` of
    its N=1 command ("plain": bench.py re-executes itself under the launcher) -- on this box's one GPU: every rank on device 0,
    torch.distributed over gloo and librxgpu's communicator over the file transport ($RXGPU_BENCH_SHARE_GPU is the hook).  The N > 1 code of
    bench.py -- rx_fm replicas with a max over ranks, rx_power tunes sharded through rxgpu_power_scan_run_sharded with the grouped gather,
    ONE compact JSON line from rank 0 on stdout, the full record in a file -- has then run before the driver's 8-GPU node runs it."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RXGPU_RCCL_LIB=_fake_rccl(), FAKE_RCCL_DIR=str(tmp_path), RXGPU_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    full_path = str(tmp_path / "full.json")
    # (--settle-ms 10: two untimed steps of the sharded sweep in front of the warm-up, each with its gather)
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--settle-ms", "10", "--blocks", "256", "--passes", "8", "--cpu-seconds", "1.5", "--full-out", full_path]
    if form == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    stdout_lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(stdout_lines) == 1 and len(stdout_lines[0]) < 8192            # ONE line, of a size the driver has been seen to keep
    compact = json.loads(stdout_lines[0])
    line = json.load(open(full_path))
    assert line["n_gpus"] == world and line["value"] > 0 and line["config"]["parallelism"].startswith("replicas x%d" % world)
    pw = line["rx_power"]
    # settle + warm-up + steps + the one parity interval
    assert pw["n_gpus"] == world and pw["config"]["rccl_ranks"] == world and pw["config"]["rccl_gathers_enqueued"] == 6
    assert "rxgpu_power_gather" in pw["config"]["gather"] and "libfake_rccl" in pw["config"]["gather"]
    assert pw["config"]["tunes_per_rank_padded"] == -(-599 // world) and pw["value"] > 0
    # ... and the gathered rows of the sharded interval were compared with the CPU checker over every tune of every rank
    v = pw["parity_sharded"]
    assert v["parity_ok"] and v["parity_ranks"] == world and v["parity_tunes_compared"] == 599 and v["parity_passes"] == 8
    assert v["parity_padding_rows_zero"] and v["parity_inputs_regenerated_match_owner_checksums"]
    assert line["parity_ok"] is True and line["parity_checked_samples"] > 0
    # the compact line alone answers the scaling questions: ranks seen by the communicator, sharded bins/s beside the 1-GPU rate of the same
    # run, the gathered rows bit-exact against the reference's scanner()
    cfg = compact["config"]
    assert compact["n_gpus"] == world and compact["steps"] == 2 and compact["warmup"] == 1 and compact["value"] == float("%.7g" % line["value"])
    assert cfg["n_ranks"] == world and cfg["rccl_ranks"] == world and cfg["rccl_gathers_enqueued"] == 6 and cfg["rccl_library"].startswith("libfake_rccl")
    assert abs(cfg["rx_power_Mbins_per_s"] / pw["value"] - 1) < 1e-4 and cfg["rx_power_ms_per_step"] > 0 and cfg["tunes_per_rank"] == -(-599 // world)
    assert cfg["rx_power_1gpu_same_run_Mbins_per_s"] > 0 and cfg["rx_power_speedup_vs_1gpu"] > 0
    assert cfg["scan_us_rank0"] > 0 and cfg["gather_us_rank0"] > 0 and "ncclGather" in cfg["gather_impl"]
    assert cfg["rx_power_parity_ok"] is True and cfg["rx_power_parity_tunes"] == 599 and cfg["rx_power_parity_ranks"] == world
    assert cfg["rx_power_padding_rows_zero"] is True
    assert compact["parity_ok"] is True and compact["parity_all_legs"]["rx_power_sharded"] is True and compact["parity_all_legs"]["rx_fm"] is True
    assert compact["roofline"]["legs"]["pw_4096"]["ok"] is True and "chan256" in compact["roofline"]["legs"]
    assert compact["roofline"]["frac"] > 0 and compact["roofline"]["avg_launch_ms"] > 0
    assert compact["cpu_baseline"]["value"] > 0 and compact["cpu_baseline"]["cores"] == 1 and cfg["rx_power_cpu_baseline_Mbins_per_s_1core"] > 0


def test_comm_rejects_a_rank_the_communicator_does_not_report(tmp_path):
    """rxgpu_comm_adopt checks rank/world against ncclCommUserRank/ncclCommCount"""
    import ctypes as C
    import os
    env_before = os.environ.get("RXGPU_RCCL_LIB")
    import subprocess
    import sys
    code = (
        "import ctypes as C, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import rx_tools_amd as R\n"
        "from rx_tools_amd import shard\n"
        "L = R.lib(); R.check(L.rxgpu_init(0))\n"
        "c = shard.Comm(shard.Comm.unique_id(), 0, 1)\n"
        "assert c.observed == (0, 1)\n"
        "fk = C.CDLL(os.environ['RXGPU_RCCL_LIB'])\n"
        "h = C.c_void_p()\n"
        "# the same ncclComm_t adopted under a wrong world size must be refused\n"
        "inner = C.cast(c._h, C.POINTER(C.c_void_p))[0]\n"
        "rc = L.rxgpu_comm_adopt(C.byref(h), inner, 0, 2)\n"
        "assert rc == -2, rc\n"
        "rc = L.rxgpu_comm_adopt(C.byref(h), inner, 0, 1)\n"
        "assert rc == 0, rc\n"
        "L.rxgpu_comm_destroy(h)\n"
        "print('adopt-ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RXGPU_RCCL_LIB=_fake_rccl(), FAKE_RCCL_DIR=str(tmp_path)),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b"adopt-ok" in out.stdout, out.stdout.decode()[-2000:]
    assert os.environ.get("RXGPU_RCCL_LIB") == env_before


@pytest.mark.parametrize("bin_e,blocks,fir,amp,tunes,passes", [
    (14, 1, 9, 32768, 1, 3),       # the bench leg's shape (-f 100M:100.1M:10 -F 9), full scale: every 32-bit tap sum, the FIR's wrap
    (14, 1, 0, 32768, 2, 2),       # no FIR: the cascade's samples themselves are the transform's input
    (14, 2, 9, 9000, 3, 2),        # two transforms per buffer, several tunes: the sums of each (pass, tune)
    (12, 2, 9, 32768, 3, 2),       # N = 4096 behind the same cascade: the transform kernel takes the dc from its registers, no sums wanted
    (10, 4, 0, 20000, 2, 1),       # N = 1024
    (15, 1, 9, 32768, 1, 2),       # N = 2^15
])
def test_four_stateless_passes_in_registers(bin_e, blocks, fir, amp, tunes, passes):
    """-F with downsample_passes = 4 (ds = 16): k_pw_fifth_regn (four eased-in fifth_order passes + droop FIR + remove_dc's sums in one launch, rtl_power.c:582-654,
    734-745) + k_pw_fifth_fix (each buffer's first samples, literally) == the oracle's scanner(): buffers of one and two transforms, with and without the FIR,
    full-scale noise and constant full-scale input (the saturated tap sums), N from 2^10 to 2^15"""
    import types
    n = 1 << bin_e
    plan = types.SimpleNamespace(bin_e=bin_e, buf_len=2 * n * 16 * blocks, downsample=16, downsample_passes=4)
    wc, sw = R.window_coefs("hamming", n), R.sine_table(bin_e)
    for data in (sig_noise(passes * tunes * plan.buf_len, seed=21 + bin_e, amp=amp),
                 np.full(passes * tunes * plan.buf_len, 32767 if fir else -32768, np.int16)):
        want, ws = oracle_scan(data, passes, tunes, plan, wc, sw, 0, fir, 0)
        got, gs = gpu_scan(data, passes, tunes, plan, wc, sw, 0, fir, 0)
        bad = np.argwhere(got != want)
        assert bad.size == 0 and np.array_equal(gs, ws), "first mismatch at %s (%d bad)" % (bad[0] if bad.size else None, len(bad))


@pytest.mark.parametrize("bin_e,ds,blocks,tunes,passes,peak", [
    (14, 4, 1, 3, 2, 0),      # four spans per buffer, six buffers: every span's wave sums and seam output land in their own buffer's pair
    (14, 8, 2, 2, 3, 1),      # two transforms per buffer (remove_dc runs over the whole buffer, rtl_power.c:744-745), peak hold
    (15, 6, 1, 2, 2, 0),      # ds = 6: windows straddle the span seams (16384 % 6 != 0), the seam kernel's outputs carry their share
    (14, 5, 1, 1, 2, 0),      # buf_len / 2 = 81920 = five spans
])
def test_boxcar_with_remove_dc_sums_riding_in_the_decimator(bin_e, ds, blocks, tunes, passes, peak):
    """rx_power's boxcar in front of a large transform where the buffers are whole 16384-sample spans: k_fm_decimate<.., DCS> leaves every wave's share
    of remove_dc's sums (rtl_power.c:609-624) and k_pw_boxcar_seams adds a span's four, its seam output, and does the buffer's atomics -- the transform
    runs no dc pass of its own.  Several tunes and passes (buffer boundaries at span boundaries), full-scale noise and a constant: == the oracle's scanner()"""
    import types
    n = 1 << bin_e
    plan = types.SimpleNamespace(bin_e=bin_e, buf_len=2 * n * ds * blocks, downsample=ds, downsample_passes=0)
    assert (plan.buf_len // 2) % 16384 == 0
    wc, sw = R.window_coefs("hamming", n), R.sine_table(bin_e)
    for data in (sig_noise(passes * tunes * plan.buf_len, seed=40 + ds, amp=32768),
                 np.full(passes * tunes * plan.buf_len, -32768, np.int16)):
        want, ws = oracle_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        got, gs = gpu_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        bad = np.argwhere(got != want)
        assert bad.size == 0 and np.array_equal(gs, ws), "first mismatch at %s (%d bad)" % (bad[0] if bad.size else None, len(bad))


@pytest.mark.parametrize("bin_e,buf_len,tunes,passes,peak,window", [
    (8, 8192, 3, 5, 0, "hamming"),       # one group of 16 side-by-side transforms per buffer
    (8, 16384, 70, 2, 0, "rectangle"),   # two groups; more than 64 tunes: the accumulators go to avg[] by atomics, not through the partial buffer
    (9, 32768, 2, 3, 1, "blackman"),     # four groups, peak hold
    (10, 16384, 5, 4, 0, "hamming"),     # N = 1024: a transform's 64 threads are one wave (wave-level ordering in the exchanges)
    (11, 8192, 2, 3, 0, "bartlett"),     # N = 2048, one group of two transforms
    (11, 32768, 1, 2, 1, "rectangle"),   # N = 2048, four groups
    (10, 4096, 2, 2, 0, "hamming"),      # HALF a group per buffer (two of four side-by-side transforms): the two-pass form keeps this shape
    (13, 16384, 3, 3, 0, "hamming"),     # N = 8192: one transform of 512 threads per buffer, one transpose area
    (13, 32768, 2, 2, 1, "rectangle"),   # ... two per buffer, peak hold
    (13, 65536, 1, 2, 0, "blackman"),    # ... four
    (5, 16384, 8, 3, 0, "hamming"),      # N = 32 (two threads per transform, 128 side by side): the coarse-bin sweeps
    (5, 8192, 70, 2, 1, "rectangle"),    # ... one group, many tunes, peak hold
    (6, 16384, 3, 2, 0, "blackman"),     # N = 64
    (7, 32768, 2, 2, 0, "hamming"),      # N = 128, four groups
    (7, 16384, 5, 3, 1, "bartlett"),     # N = 128, peak hold
])
def test_small_transforms_with_the_buffer_in_registers(bin_e, buf_len, tunes, passes, peak, window):
    """N = 32 ... 2048 and 8192 (k_pw_fftR2: a thread holds all its samples of the pass -- one, two or four groups of side-by-side transforms --, remove_dc from
    the registers, the next pass on its way, rtl_power.c:744-768) on full-scale noise and on a constant, with and without peak hold, few and many tunes
    == the oracle's scanner(); a buffer that is no whole number of groups stays with k_pw_fftR"""
    import types
    n = 1 << bin_e
    plan = types.SimpleNamespace(bin_e=bin_e, buf_len=buf_len, downsample=1, downsample_passes=0)
    wc, sw = R.window_coefs(window, n), R.sine_table(bin_e)
    for data in (sig_noise(passes * tunes * buf_len, seed=60 + bin_e, amp=32768), np.full(passes * tunes * buf_len, 32767, np.int16)):
        want, ws = oracle_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        got, gs = gpu_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        bad = np.argwhere(got != want)
        assert bad.size == 0 and np.array_equal(gs, ws), "first mismatch at %s (%d bad)" % (bad[0] if bad.size else None, len(bad))


@pytest.mark.parametrize("buf_len,peak", [(16384, 0), (16384, 1), (16390, 0), (4098, 0)])
def test_rms_power_path_full_scale(buf_len, peak):
    """bin_e == 0 (bins of 1 MHz and more: rms_power, rtl_power.c:403-429, 710-713): sum and sum of squares of a buffer's int16 -- 16-byte loads, the pair
    of squares of a dword as one unsigned 32-bit value (2 x 32768^2 = 2^31 at full scale) -- on full-scale noise and on a constant -32768; buffers
    whose length leaves every second one misaligned take the element-wise form.  == the oracle's scanner(), fp64 dc term included"""
    import types
    plan = types.SimpleNamespace(bin_e=0, buf_len=buf_len, downsample=1, downsample_passes=0)
    wc, sw = R.window_coefs("rectangle", 1), R.sine_table(0)
    tunes, passes = 5, 3
    for data in (sig_noise(passes * tunes * buf_len, seed=9, amp=32768), np.full(passes * tunes * buf_len, -32768, np.int16)):
        want, ws = oracle_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        got, gs = gpu_scan(data, passes, tunes, plan, wc, sw, 1, 0, peak)
        assert np.array_equal(got, want) and np.array_equal(gs, ws)


@pytest.mark.parametrize("bin_e,ds_p,fir,tunes,passes", [
    (10, 5, 9, 2, 2), (10, 6, 0, 1, 2), (10, 7, 9, 2, 1), (12, 5, 0, 3, 2), (14, 5, 9, 1, 2), (8, 8, 9, 1, 2),
    (10, 2, 0, 3, 2), (12, 2, 9, 2, 2), (14, 2, 9, 1, 2), (11, 3, 9, 2, 3), (13, 3, 0, 2, 2), (15, 3, 9, 1, 2), (14, 3, 0, 2, 1),
])
def test_stateless_passes_other_than_four(bin_e, ds_p, fir, tunes, passes):
    """-F with five to eight passes (ds = 32 ... 256, rtl_power.c:734-745): the first four in the register kernel (k_pw_fifth_regn without FIR and sums,
    k_pw_fifth_fix for every buffer's first level-4 samples), the others on the 1/16-rate buffers, then the droop FIR and remove_dc as before; with two
    or three passes (ds = 4, 8): the whole cascade, the FIR and -- in front of a large transform -- the sums in the register kernel (k_pw_fifth_regn<2>,
    <3>) -- on full-scale noise and on a constant == the oracle's scanner()"""
    import types
    n = 1 << bin_e
    plan = types.SimpleNamespace(bin_e=bin_e, buf_len=2 * n << ds_p, downsample=1 << ds_p, downsample_passes=ds_p)
    wc, sw = R.window_coefs("hamming", n), R.sine_table(bin_e)
    for data in (sig_noise(passes * tunes * plan.buf_len, seed=70 + ds_p, amp=32768), np.full(passes * tunes * plan.buf_len, 32767, np.int16)):
        want, ws = oracle_scan(data, passes, tunes, plan, wc, sw, 0, fir, 0)
        got, gs = gpu_scan(data, passes, tunes, plan, wc, sw, 0, fir, 0)
        bad = np.argwhere(got != want)
        assert bad.size == 0 and np.array_equal(gs, ws), "first mismatch at %s (%d bad)" % (bad[0] if bad.size else None, len(bad))


def test_buffer_that_is_no_whole_number_of_large_transforms():
    """N = 2^16 with 1.5 transforms per decimated buffer (boxcar ds = 2 on 3 * 2^17 int16: the second transform is half samples, half the zeros
    the boxcar leaves behind, rtl_power.c:723-733) -- a geometry the reference's planner never makes and rxgpu_power_scan_create accepts: it takes
    the one-launch-per-radix-2-stage network (k_pwb_*) instead of the radix-16 passes, and gives the oracle's sums"""
    import types
    plan = types.SimpleNamespace(bin_e=16, buf_len=3 << 17, downsample=2, downsample_passes=0)
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    data = sig_noise(2 * plan.buf_len, seed=5, amp=12000)
    want, ws = oracle_scan(data, 2, 1, plan, wc, sw, 1, 0, 0)
    got, gs = gpu_scan(data, 2, 1, plan, wc, sw, 1, 0, 0)
    assert np.array_equal(got, want) and np.array_equal(gs, ws)


@pytest.mark.parametrize("rng", ["100M:102M:20", "100M:102.8M:20", "100M:102.8M:2"])
def test_two_head_passes_in_one_launch_or_two(rng):
    """N = 2^17, 2^18, 2^21: the first two radix-16 passes fused (k_pwm_head2: 8-byte loads) and -- what an input that is not 8-byte aligned
    takes -- as a launch each through the scratch copy (k_pwm_head + k_pwm_head_mid) are the same fix_fft: identical avg[] (the aligned form equals
    the oracle in test_scan_bit_exact)"""
    from gpu_support import to_dev, torch_cuda
    torch = torch_cuda()
    plan = R.plan_range(rng, 0.0, 1)
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs("blackman", n), R.sine_table(plan.bin_e)
    data = sig_noise(2 * plan.buf_len, seed=11, amp=32768)
    a, sa = gpu_scan(data, 2, 1, plan, wc, sw, 1, 0, 0)
    s = R.PowerScan(R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0), 1, wc, sw)
    d_in = to_dev(np.concatenate([np.zeros(2, np.int16), data]))               # the capture starts 4 bytes into the allocation
    d_avg = torch.zeros((1, n), dtype=torch.int64, device="cuda")
    d_samples = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    s.run(d_in.data_ptr() + 4, 2, 1, d_avg.data_ptr(), d_samples.data_ptr())
    R.check(R.lib().rxgpu_sync())
    b, sb = d_avg.cpu().numpy(), d_samples.cpu().numpy()
    s.close()
    assert np.array_equal(a, b) and np.array_equal(sa, sb)


def test_gather_of_no_tunes_is_a_noop():
    L = R.lib()
    R.check(L.rxgpu_init(0))
    R.check(L.rxgpu_power_gather(None, None, None, 0, 4096, None, None, 0))
    from rx_tools_amd import shard
    first, count, per = shard.tune_range(0, 4, 0)
    assert (first, count, per) == (0, 0, 0)


def test_gather_without_a_communicator_is_a_copy():
    from gpu_support import torch_cuda
    torch = torch_cuda()
    L = R.lib()
    R.check(L.rxgpu_init(0))
    a = torch.arange(3 * 8, dtype=torch.int64, device="cuda").reshape(3, 8)
    s = torch.arange(3, dtype=torch.int32, device="cuda")
    a2, s2 = torch.zeros_like(a), torch.zeros_like(s)
    torch.cuda.synchronize()
    R.check(L.rxgpu_power_gather(None, a.data_ptr(), s.data_ptr(), 3, 8, a2.data_ptr(), s2.data_ptr(), 0))
    R.check(L.rxgpu_sync())
    assert torch.equal(a, a2) and torch.equal(s, s2)
