"""Time k_fm_decimate from an experimental build of the library (diagnostic): python tools/dec_exp.py <lib.so>"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd._lib as LL
LL.LIB_PATH = os.path.abspath(sys.argv[1])
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
blocks, bl = 8192, 2 * 131072
base = torch.from_numpy(R.synth.sig_fm(8 * 131072)).cuda()
d_iq = base.repeat(blocks // 8)[: blocks * bl].contiguous()
d_out = torch.zeros(blocks * 131072 // 100 + 64, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
s = R.FmStream(R.FmParams.wbfm(downsample=118), blocks, bl)
for _ in range(3): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
L.rxgpu_prof_reset(); L.rxgpu_prof_enable(1)
for _ in range(10): s.run(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
L.rxgpu_prof_enable(0)
ms, k = C.c_double(0), C.c_long(0)
L.rxgpu_prof_get(b"fm_decimate", C.byref(ms), C.byref(k))
us = ms.value / max(1, k.value) * 1e3
print("%-40s %8.1f us  %7.1f GB/s" % (os.path.basename(sys.argv[1]), us, blocks * bl * 2 / (us * 1e-6) / 1e9))
