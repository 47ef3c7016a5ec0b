"""Stage times of the small-decimation chain (ds=6 / ds=5): tiled vs LDS-staged audio path, pipelined and not."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bl = 2 * 131072
g = torch.Generator(device="cuda"); g.manual_seed(1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_capture
d_iq = device_capture(torch, torch.device("cuda"), blocks * 131072, seed=5)
def dump(names):
    out = {}
    for n in names:
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k))
        if k.value: out[n] = round(ms.value / k.value * 1e3, 1)
    return out
for name, kw, per in [("ds=6", dict(downsample=6), 6), ("ds=5/240k", dict(downsample=5, rate_out=240000, deemph_a=19), 5), ("ds=118", dict(downsample=118), 118)]:
    for tiled in (1, 0):
        if tiled: os.environ.pop("RXGPU_NO_TILED", None)
        else: os.environ["RXGPU_NO_TILED"] = "1"
        d_out = torch.zeros(blocks * 131072 // per + 64, dtype=torch.int16, device="cuda")
        s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)
        for mode in ("serial", "pipelined"):
            for _ in range(2): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
            L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            k = 6
            for _ in range(k):
                if mode == "serial": s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
                else: s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
            s.wait()
            dt = (time.perf_counter() - t0) / k
            L.rxgpu_prof_enable(0)
            print(name, "tiled" if tiled else "staged", mode, "us/step", round(dt * 1e6, 1), dump(["fm_decimate", "fm_disc", "fm_deemph", "fm_resample"]),
                  "TS/s", round(blocks * 131072 / dt / 1e12, 3), flush=True)
        s.close(); del d_out
