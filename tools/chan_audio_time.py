"""tools/chan_audio_time.py -- the channeliser's bench shape with the per-channel audio stages on (deemph_filter a=7 at 19.5 kHz, low_pass_real to 8 kHz): ms per 1 GiB run"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
block_len, bin_e, n_ch, n_blocks = 2 * 131072, 10, 256, 2048
T = n_blocks * (block_len // 2)
d_iq = device_capture(torch, torch.device("cuda"), T, seed=4242, amp=600.0)
windows = T >> bin_e
d_out = torch.zeros((n_ch, windows), dtype=torch.int16, device="cuda")
for label, p in (("no audio stages", R.ChanParams(bin_e, 384, n_ch, 1, 0, 0, 0, -1, 0)), ("deemph a=7 + low_pass_real 19531 -> 8000", R.ChanParams(bin_e, 384, n_ch, 1, 1, 7, 19531, 8000, 0)),
                 ("deemph a=7 only", R.ChanParams(bin_e, 384, n_ch, 1, 1, 7, 19531, -1, 0))):
    ch = R.Channeliser(p, n_blocks, block_len, R.sine_table(bin_e))
    for _ in range(4): ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
    L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    L.rxgpu_prof_enable(0)
    ks = {}
    for nme in ("ch_fft", "ch_demod", "ch_audio"):
        ms, k = C.c_double(0), C.c_long(0)
        L.rxgpu_prof_get(nme.encode(), C.byref(ms), C.byref(k))
        if k.value: ks[nme] = round(ms.value / k.value * 1e3, 1)
    print(label.ljust(44), "ms", round(dt * 1e3, 3), "GS/s", round(T / dt / 1e9, 1), ks, flush=True)
    ch.close()
