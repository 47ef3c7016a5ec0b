"""rx_power geometries: time per launch with prof level 2 (pw_downsample / pw_fft), with and without the N=2^14 register-blocked kernel"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
def prof(n):
    ms, k = C.c_double(0), C.c_long(0); L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k)); return round(ms.value / max(1, k.value) * 1e3, 1)
g = torch.Generator(device="cuda"); g.manual_seed(3)
for label, rng, boxcar, fir, npasses in (("N=16384 -F 9 ds=16", "100M:100.1M:10", 0, 9, 4096), ("N=16384 boxcar ds=28", "100M:100.1M:10", 1, 0, 4096),
                                         ("N=8192", "100M:102.5M:600", 1, 0, 2048), ("N=32768", "100M:100.2M:10", 1, 0, 1024), ("N=4096 config3", "24M:1.7G:1k", 1, 0, 512)):
    pl = R.plan_range(rng, 0.0, boxcar)
    nn = 1 << pl.bin_e
    for r14 in (1, 0):
        if r14: os.environ.pop("RXGPU_FFT_NO_R14", None)
        else: os.environ["RXGPU_FFT_NO_R14"] = "1"
        if pl.bin_e != 14 and not r14: continue
        p2 = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, boxcar, fir, 0), pl.tune_count, R.window_coefs("hamming", nn), R.sine_table(pl.bin_e))
        di = torch.randint(-2000, 2001, (npasses, pl.tune_count, pl.buf_len), dtype=torch.int16, device="cuda", generator=g)
        da = torch.zeros((pl.tune_count, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(pl.tune_count, dtype=torch.int32, device="cuda")
        for _ in range(2): p2.run(di.data_ptr(), npasses, pl.tune_count, da.data_ptr(), dsm.data_ptr())
        L.rxgpu_sync(); L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
        t0 = time.perf_counter()
        for _ in range(5): p2.run(di.data_ptr(), npasses, pl.tune_count, da.data_ptr(), dsm.data_ptr())
        L.rxgpu_sync(); dt = (time.perf_counter() - t0) / 5; L.rxgpu_prof_enable(0)
        ins = npasses * pl.tune_count * (pl.buf_len // 2)
        print(label, "N", nn, "ds", pl.downsample, "tunes", pl.tune_count, "buf_len", pl.buf_len, "r14" if r14 else "radix2", "ms", round(dt * 1e3, 3), "downsample us", prof("pw_downsample"), "fft us", prof("pw_fft"),
              "in GS/s", round(ins / dt / 1e9, 1), "Gbins/s", round(ins / pl.downsample / dt / 1e9, 2), flush=True)
        p2.close(); del di, da, dsm
