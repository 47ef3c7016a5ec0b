"""tools/chan_once.py -- six runs of the bench line's channeliser shape (256 channels of a 1024-bin bank, 1 GiB of capture per run) and nothing else: what the
`chan` counter passes of tools/profile_round.sh profile (bench.py's own leg warms up for 40 runs and times the NCO mode too)"""
import os, sys
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
ch = R.Channeliser(R.ChanParams(bin_e, 384, n_ch, 1), n_blocks, block_len, R.sine_table(bin_e))
for _ in range(6):
    ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
torch.cuda.synchronize()
ch.close()
print("ok")
