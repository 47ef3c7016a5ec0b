"""tools/pw_big_time.py -- G bins/s of the large-N rx_power geometries (one tune, the bench legs' shapes and N = 2^16 .. 2^21), best of five timed launches each;
run it under two values of $RXGPU_LIB_FLAVOUR (a scratch build librxgpu_<name>.so beside the product) for an A/B of compile-time choices on one box"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
g = torch.Generator(device="cuda"); g.manual_seed(3)
out = []
for rng, passes, boxcar, fir in (("100M:100.1M:10", 4096, 0, 9), ("100M:100.1M:10", 4096, 1, 0), ("100M:100.2M:10", 2048, 1, 0), ("100M:102M:40", 512, 1, 0), ("100M:102M:20", 256, 1, 0),
                                 ("100M:102.8M:20", 256, 1, 0), ("100M:102.8M:5", 64, 1, 0), ("100M:102.8M:2", 32, 1, 0)):
    pl = R.plan_range(rng, 0.0, boxcar)
    nn = 1 << pl.bin_e
    di = torch.randint(-2000, 2001, (passes, 1, pl.buf_len), dtype=torch.int16, device="cuda", generator=g)
    ps = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, boxcar, fir, 0), 1, R.window_coefs("hamming", nn), R.sine_table(pl.bin_e))
    da = torch.zeros((1, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(3): ps.run(di.data_ptr(), passes, 1, da.data_ptr(), dsm.data_ptr())
    L.rxgpu_sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ps.run(di.data_ptr(), passes, 1, da.data_ptr(), dsm.data_ptr()); L.rxgpu_sync(); best = min(best, time.perf_counter() - t0)
    bins = passes * (pl.buf_len // 2) // pl.downsample
    out.append("N=2^%-2d ds=%-2d %s%s  %7.1f us  %6.1f G bins/s" % (pl.bin_e, pl.downsample, "boxcar" if boxcar else "fifth ", " +fir" if fir else "     ", best * 1e6, bins / best / 1e9))
    ps.close(); del di, da
print(os.environ.get("RXGPU_LIB_FLAVOUR", "product"), "\n  " + "\n  ".join(out))
