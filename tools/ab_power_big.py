"""tools/ab_power_big.py -- one-tune sweeps at N = 2^16 .. 2^21: the radix-16 path (the first two passes fused, or -- $RXGPU_FFT_HEAD2=0 -- a launch
each) against the one-launch-per-stage network ($RXGPU_FFT_STAGEWISE)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
g = torch.Generator(device="cuda"); g.manual_seed(3)
for rng, passes in (("100M:102M:40", 512), ("100M:102M:20", 256), ("100M:102.8M:20", 256), ("100M:102.8M:5", 64), ("100M:102.8M:2", 32)):
    pl = R.plan_range(rng, 0.0, 1)
    nn = 1 << pl.bin_e
    di = torch.randint(-2000, 2001, (passes, 1, pl.buf_len), dtype=torch.int16, device="cuda", generator=g)
    res = {}
    for mode in ("radix16", "head2=0", "radix16", "head2=0", "stagewise"):
        os.environ.pop("RXGPU_FFT_STAGEWISE", None); os.environ.pop("RXGPU_FFT_HEAD2", None)
        if mode == "stagewise": os.environ["RXGPU_FFT_STAGEWISE"] = "1"
        if mode == "head2=0": os.environ["RXGPU_FFT_HEAD2"] = "0"
        p2 = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, 1, 0, 0), 1, R.window_coefs("hamming", nn), R.sine_table(pl.bin_e))
        da = torch.zeros((1, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(1, dtype=torch.int32, device="cuda")
        p2.run(di.data_ptr(), passes, 1, da.data_ptr(), dsm.data_ptr()); L.rxgpu_sync()
        res[mode] = da.clone()
        t0 = time.perf_counter()
        for _ in range(3): p2.run(di.data_ptr(), passes, 1, da.data_ptr(), dsm.data_ptr())
        L.rxgpu_sync(); dt = (time.perf_counter() - t0) / 3
        print("N=2^%d" % pl.bin_e, mode.ljust(9), "ms", round(dt * 1e3, 3), "Gbins/s", round(passes * (pl.buf_len // 2) / dt / 1e9, 2), flush=True)
        p2.close()
    print("   same avg[]:", bool(torch.equal(res["radix16"], res["stagewise"]) and torch.equal(res["radix16"], res["head2=0"])))
