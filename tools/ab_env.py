"""tools/ab_env.py BLOCKS DS VAR=VAL[,VAR=VAL...] [VAR=VAL...] -- pipelined step time of one rx_fm chain under several environment settings, alternating
inside one process (same box).  DS: 118 | 6 | -7 (passes) | -39 (3 passes + droop FIR).  'default' (no variable) is always measured too.
AB_SYNC=1 in the environment: runs one at a time (no overlap between consecutive runs)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
blocks, ds = int(sys.argv[1]), int(sys.argv[2])
settings = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[3:]]
allvars = sorted({k for st in settings for k in st})
bl = 2 * 131072
d_iq = device_capture(torch, torch.device("cuda"), blocks * 131072, seed=5)
d_out = torch.zeros(blocks * 131072 // (ds if ds > 0 else 8) + 64, dtype=torch.int16, device="cuda")
kw = dict(downsample=ds) if ds > 0 else (dict(downsample_passes=-ds) if ds > -10 else dict(downsample_passes=(-ds) // 10, comp_fir_size=9))
if ds == 5: kw.update(rate_out=240000, deemph_a=19)
if os.environ.get("AB_KW"):                       # e.g. AB_KW=deemph=0 (no audio stages behind the discriminator: the linear pcm layout)
    kw.update({k: int(v) for k, v in (kv.split("=") for kv in os.environ["AB_KW"].split(","))})
def dump(names):
    out = {}
    for n in names:
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(n.encode(), C.byref(ms), C.byref(k))
        if k.value: out[n] = round(ms.value / k.value * 1e3, 1)
    return out
for rep in range(3):
    for st in settings:
        for v in allvars: os.environ.pop(v, None)
        os.environ.update(st)
        s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)          # a stream keeps the knobs it was created with (rxgpu_knob snapshot)
        for _ in range(3): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
        L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 20
        if os.environ.get("AB_SYNC"):                 # every run alone on the device: what each kernel takes without the other stream beside it
            for _ in range(k): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
        else:
            for _ in range(k): s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
            s.wait()
        dt = (time.perf_counter() - t0) / k
        L.rxgpu_prof_enable(0)
        s.close()
        print((",".join("%s=%s" % kv for kv in st.items()) or "default").ljust(34), "us/step", round(dt * 1e6, 1),
              dump(["fm_decimate", "fm_fifth", "fm_fifth2", "fm_droop", "fm_disc", "fm_deemph", "fm_resample"]), "TS/s", round(blocks * 131072 / dt / 1e12, 3), flush=True)
