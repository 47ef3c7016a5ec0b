"""Whole-job pipelined rx_fm rate for another parameter set than bench.py's headline (diagnostic):
   python tools/pipelined_variant.py downsample=6     |  downsample_passes=7  |  downsample=118 custom_atan=0 ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
kw = dict(a.split("=") for a in sys.argv[1:])
kw = {k: int(v) for k, v in kw.items()}
blocks, bl = kw.pop("blocks", 8192), 2 * 131072
base = torch.from_numpy(R.synth.sig_fm(8 * 131072)).cuda()
d_iq = base.repeat(blocks // 8)[: blocks * bl].contiguous()
d_out = torch.zeros(blocks * 131072 // max(1, min(kw.get("downsample", 6), 6)) + 64, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)
for _ in range(5):
    s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
s.wait()
steps = 40
t0 = time.perf_counter()
for _ in range(steps):
    s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
s.wait()
dt = time.perf_counter() - t0
print(kw, "ms/step %.3f  %.1f GSample/s" % (dt / steps * 1e3, blocks * 131072 * steps / dt / 1e9))
s.close()
