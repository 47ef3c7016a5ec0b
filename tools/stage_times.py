"""Print per-stage device times (hipEvents, prof level 2) for one rx_fm step and one rx_power step."""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
def dump(names):
    out = {}
    for n in names:
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k))
        if k.value: out[n] = round(ms.value / k.value * 1e3, 1)
    return out
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bl = 2 * 131072
base = torch.from_numpy(R.synth.sig_fm(8 * 131072)).cuda()
d_iq = base.repeat(blocks // 8)[: blocks * bl].contiguous()
d_out = torch.zeros(blocks * 131072 // 118 + 64, dtype=torch.int16, device="cuda")
for name, kw in [("low_pass ds=118", dict(downsample=118)), ("fifth_order x7", dict(downsample_passes=7)), ("low_pass ds=6", dict(downsample=6))]:
    s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)
    if kw.get("downsample") == 6:
        d_out = torch.zeros(blocks * 131072 // 6 + 64, dtype=torch.int16, device="cuda")
    for _ in range(2): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
    L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
    dt = (time.perf_counter() - t0) / 5
    L.rxgpu_prof_enable(0)
    st = dump(["fm_decimate", "fm_decimate_generic", "fm_fifth", "fm_droop", "fm_disc", "fm_deemph", "fm_resample"])
    print(name, "wall us/step", round(dt * 1e6, 1), "stages us:", st, "GS/s", round(blocks * 131072 / dt / 1e9, 1))
    s.close()
