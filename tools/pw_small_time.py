"""tools/pw_small_time.py -- G bins/s of rx_power at the configs[2] buffer shape (599 tunes x 16384 int16, 512 passes resident) for N = 256 ... 8192 (k_pw_fftR<M>;
N = 4096: k_pw_fft4096), best of five launches each: how the register-blocked transform holds up beside the bench's N = 4096"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
g = torch.Generator(device="cuda"); g.manual_seed(3)
tunes, passes, buf_len = int(os.environ.get("PW_TUNES", "599")), int(os.environ.get("PW_PASSES", "256")), 16384
di = torch.randint(-100, 101, (passes, tunes, buf_len), dtype=torch.int16, device="cuda", generator=g)
for bin_e in (int(v) for v in (sys.argv[1:] or "8 9 10 11 12 13".split())):
    nn = 1 << bin_e
    ps = R.PowerScan(R.PowerParams(bin_e, buf_len, 1, 0, 1, 0, 0), tunes, R.window_coefs("rectangle", nn), R.sine_table(bin_e))
    da = torch.zeros((tunes, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(tunes, dtype=torch.int32, device="cuda")
    for _ in range(3): ps.run(di.data_ptr(), passes, tunes, da.data_ptr(), dsm.data_ptr())
    L.rxgpu_sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ps.run(di.data_ptr(), passes, tunes, da.data_ptr(), dsm.data_ptr()); L.rxgpu_sync(); best = min(best, time.perf_counter() - t0)
    bins = passes * tunes * (buf_len // 2)
    print("N=2^%-2d  %8.1f us  %6.1f G bins/s  (%.2f TB/s of input)" % (bin_e, best * 1e6, bins / best / 1e9, bins * 4 / best / 1e12), flush=True)
    ps.close(); del da
