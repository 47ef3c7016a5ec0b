"""A/B of launch-time environment switches on the headline chain (ds=118) inside one process: same box, alternating."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
switch = sys.argv[2] if len(sys.argv) > 2 else "RXGPU_DEC_NARROW"
ds = int(sys.argv[3]) if len(sys.argv) > 3 else 118
bl = 2 * 131072
d_iq = device_capture(torch, torch.device("cuda"), blocks * 131072, seed=5)
d_out = torch.zeros(blocks * 131072 // (ds if ds > 0 else 8) + 64, dtype=torch.int16, device="cuda")
s = R.FmStream(R.FmParams.wbfm(downsample=ds) if ds > 0 else (R.FmParams.wbfm(downsample_passes=-ds) if ds > -10 else R.FmParams.wbfm(downsample_passes=(-ds) // 10, comp_fir_size=9)), blocks, bl)   # ds: 118 | -7 (passes) | -39 (3 passes + droop FIR)
def dump(names):
    out = {}
    for n in names:
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k))
        if k.value: out[n] = round(ms.value / k.value * 1e3, 1)
    return out
for rep in range(4):
    for on in (1, 0):
        if on: os.environ[switch] = sys.argv[4] if len(sys.argv) > 4 else "1"
        else: os.environ.pop(switch, None)
        L.rxgpu_knobs_reload()       # kernel-variant knobs are read from a snapshot; plan knobs (RXGPU_FUSE_A, ...) need a new stream: tools/ab_env.py
        for _ in range(3): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
        L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 30
        for _ in range(k): s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
        s.wait()
        dt = (time.perf_counter() - t0) / k
        L.rxgpu_prof_enable(0)
        print((switch + "=" + (sys.argv[4] if len(sys.argv) > 4 else "1") if on else "default").ljust(22), "us/step", round(dt * 1e6, 1), dump(["fm_decimate", "fm_fifth", "fm_fifth2", "fm_droop", "fm_disc", "fm_deemph", "fm_resample"]),
              "TS/s", round(blocks * 131072 / dt / 1e12, 3), flush=True)
