"""PCIe-inclusive rate of rxgpu_fm_stream_run_host (pageable host buffer in, host audio out)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rx_tools_amd as R
R.check(R.lib().rxgpu_init(0))
blocks, bl = 1024, 2 * 131072
iq = np.tile(R.synth.sig_fm(8 * 131072), blocks // 8)
for pinned in (False, True):
    buf = iq
    if pinned:
        t = torch.from_numpy(iq).pin_memory(); buf = t.numpy()
    out = np.zeros(blocks * 131072 // 118 + 64, np.int16)
    s = R.FmStream(R.FmParams.wbfm(downsample=118), blocks, bl)
    s.run_host(buf.ctypes.data, blocks, bl, out.ctypes.data, out.size)
    t0 = time.perf_counter()
    for _ in range(5): s.run_host(buf.ctypes.data, blocks, bl, out.ctypes.data, out.size)
    dt = (time.perf_counter() - t0) / 5
    print("host-fed (%s): %.1f GS/s = %.1f GB/s over PCIe" % ("pinned" if pinned else "pageable", blocks * 131072 / dt / 1e9, blocks * bl * 2 / dt / 1e9))
    s.close()
