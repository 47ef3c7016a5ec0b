"""Stage times of the -F chain for 1..7 passes (which fused group costs what)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bl = 2 * 131072
d_iq = device_capture(torch, torch.device("cuda"), blocks * 131072, seed=5)
def dump(names):
    out = {}
    for n in names:
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k))
        if k.value: out[n] = round(ms.value / k.value * 1e3, 1)
    return out
for passes in (1, 2, 3, 6, 7):
    d_out = torch.zeros((blocks * 131072 >> passes) + 64, dtype=torch.int16, device="cuda")
    s = R.FmStream(R.FmParams.wbfm(downsample_passes=passes), blocks, bl)
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
        print("passes", passes, mode.ljust(9), "us/step", round(dt * 1e6, 1), dump(["fm_fifth", "fm_fifth2", "fm_droop", "fm_disc", "fm_deemph", "fm_resample"]),
              "TS/s", round(blocks * 131072 / dt / 1e12, 3), flush=True)
    s.close(); del d_out
