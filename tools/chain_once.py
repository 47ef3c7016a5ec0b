"""tools/chain_once.py BLOCKS DS [RUNS] -- a few pipelined runs of one rx_fm chain and nothing else (what rocprofv3 wraps for per-kernel counters).
DS: 118 | 6 | 5 | -7 (passes) | -39 (3 passes + droop FIR)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
blocks, ds = int(sys.argv[1]), int(sys.argv[2])
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
bl = 2 * 131072
d_iq = device_capture(torch, torch.device("cuda"), blocks * 131072, seed=5)
d_out = torch.zeros(blocks * 131072 // (ds if ds > 0 else 8) + 64, dtype=torch.int16, device="cuda")
kw = dict(downsample=ds) if ds > 0 else (dict(downsample_passes=-ds) if ds > -10 else dict(downsample_passes=(-ds) // 10, comp_fir_size=9))
if ds == 5: kw.update(rate_out=240000, deemph_a=19)
s = R.FmStream(R.FmParams.wbfm(**kw), blocks, bl)
for _ in range(runs):
    s.run_async(d_iq.data_ptr(), blocks, bl, d_out.data_ptr(), d_out.numel())
s.wait()
s.close()
