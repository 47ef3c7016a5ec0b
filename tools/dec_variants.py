"""Decimator kernel time alone for a few parameter variants (diagnostic): which part of k_fm_decimate costs bandwidth."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bl = 2 * 131072
base = torch.from_numpy(R.synth.sig_fm(8 * 131072)).cuda()
d_iq = base.repeat(blocks // 8)[: blocks * bl].contiguous()
d_out = torch.zeros(blocks * 131072 // 6 + 64, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for name, kw in [("ds=118", dict(downsample=118)), ("ds=118 prescaled", dict(downsample=118, prescaled=1)),
                 ("ds=118 offset_tuning", dict(downsample=118, offset_tuning=1)), ("ds=118 -A std (no fused disc)", dict(downsample=118, custom_atan=0)),
                 ("ds=6", dict(downsample=6)), ("ds=1000", dict(downsample=1000))]:
    s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)
    for _ in range(3): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
    L.rxgpu_prof_reset(); L.rxgpu_prof_enable(1)
    for _ in range(10): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
    L.rxgpu_prof_enable(0)
    ms, k = C.c_double(0), C.c_long(0)
    L.rxgpu_prof_get(b"fm_decimate", C.byref(ms), C.byref(k))
    us = ms.value / max(1, k.value) * 1e3
    print("%-34s %8.1f us  %7.1f GB/s" % (name, us, blocks * bl * 2 / (us * 1e-6) / 1e9))
    s.close()
