"""tools/chan_once.py [MODE] -- six runs of the bench line's channeliser shape (256 channels of a 1024-bin bank, 1 GiB of capture per run) and nothing else:
what the `chan` counter passes of tools/profile_round.sh profile (bench.py's own leg warms up for 40 runs and times the other modes too).
MODE: fast (default: -A fast fused into the bank) | std (libm atan2 per sample, k_ch_demod: what NBFM defaults to) | audio (de-emphasis + low_pass_real per
channel) | nco (SURVEY's literal NCO -> low_pass definition, on 1/8 of the capture like the bench leg)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
from bench import device_capture
L = R.lib(); R.check(L.rxgpu_init(0))
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
block_len, bin_e, n_ch, n_blocks = 2 * 131072, 10, 256, 2048
T = n_blocks * (block_len // 2)
d_iq = device_capture(torch, torch.device("cuda"), T, seed=4242, amp=600.0)
if mode == "nco":
    n_blocks //= 8
    T //= 8
windows = T >> bin_e
d_out = torch.zeros((n_ch, windows), dtype=torch.int16, device="cuda")
prm = {"fast": R.ChanParams(bin_e, 384, n_ch, 1), "std": R.ChanParams(bin_e, 384, n_ch, 0),
       "audio": R.ChanParams(bin_e, 384, n_ch, 1, 1, 7, 19531, 8000, 0), "nco": R.ChanParams(bin_e, 384, n_ch, 1, 0, 0, 0, -1, 1)}[mode]
ch = R.Channeliser(prm, n_blocks, block_len, R.sine_table(bin_e))
for _ in range(6):
    ch.run(d_iq.data_ptr(), n_blocks, block_len, d_out.data_ptr(), windows)
torch.cuda.synchronize()
ch.close()
print("ok", mode, T)
