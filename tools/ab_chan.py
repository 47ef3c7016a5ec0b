"""tools/ab_chan.py [VAR=VAL ...] -- the 256-channel channeliser run of bench.py (1 GiB capture, N=1024) under environment settings, alternating in one process"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
settings = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
allvars = sorted({k for st in settings for k in st})
block_len, bin_e, n_ch, n_blocks = 2 * 131072, 10, 256, 2048
T = n_blocks * (block_len // 2)
d_iq = device_capture(torch, torch.device("cuda"), T, seed=4242, amp=600.0)
windows = T >> bin_e
d_out = torch.zeros((n_ch, windows), dtype=torch.int16, device="cuda")
ref = None
for rep in range(3):
    for st in settings:
        for v in allvars: os.environ.pop(v, None)
        os.environ.update(st)
        ch = R.Channeliser(R.ChanParams(bin_e, 384, n_ch, 1), n_blocks, block_len, R.sine_table(bin_e))
        ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        if ref is None: ref = d_out.clone()
        same = bool(torch.equal(ref, d_out))
        for _ in range(3): ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        L.rxgpu_prof_enable(0)
        ks = {}
        for nme in ("ch_fft", "ch_demod"):
            ms, k = C.c_double(0), C.c_long(0)
            L.rxgpu_prof_get(nme.encode(), C.byref(ms), C.byref(k))
            if k.value: ks[nme] = round(ms.value / k.value * 1e3, 1)
        print((",".join("%s=%s" % kv for kv in st.items()) or "default").ljust(24), "ms", round(dt * 1e3, 3), "GS/s", round(T / dt / 1e9, 1), ks, "same output:", same, flush=True)
        ch.close()
