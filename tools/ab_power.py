"""tools/ab_power.py [VAR=VAL ...] -- the configs[2] rx_power launch (599 tunes x 512 passes, N=4096) under several environment settings, alternating in one process"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
settings = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
allvars = sorted({k for st in settings for k in st})
pl = R.plan_range("24M:1.7G:1k", 0.0, 1)
nn, passes = 1 << pl.bin_e, 512
g = torch.Generator(device="cuda"); g.manual_seed(3)
di = torch.randint(-100, 101, (passes, pl.tune_count, pl.buf_len), dtype=torch.int16, device="cuda", generator=g)
da = torch.zeros((pl.tune_count, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(pl.tune_count, dtype=torch.int32, device="cuda")
ref = None
for rep in range(3):
    for st in settings:
        for v in allvars: os.environ.pop(v, None)
        os.environ.update(st)
        p2 = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, 1, 0, 0), pl.tune_count, R.window_coefs("rectangle", nn), R.sine_table(pl.bin_e))
        da.zero_(); dsm.zero_()
        p2.run(di.data_ptr(), passes, pl.tune_count, da.data_ptr(), dsm.data_ptr()); L.rxgpu_sync()
        if ref is None: ref = da.clone()
        same = bool(torch.equal(ref, da))
        t0 = time.perf_counter()
        for _ in range(10): p2.run(di.data_ptr(), passes, pl.tune_count, da.data_ptr(), dsm.data_ptr())
        L.rxgpu_sync(); dt = (time.perf_counter() - t0) / 10
        print((",".join("%s=%s" % kv for kv in st.items()) or "default").ljust(28), "ms", round(dt * 1e3, 3), "Gbins/s", round(passes * pl.tune_count * 8192 / dt / 1e9, 1), "same avg[] as the first setting:", same, flush=True)
        p2.close()
