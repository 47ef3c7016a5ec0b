"""tools/pw_big_once.py RANGE PASSES [REPS] -- one rx_power geometry and nothing else (what rocprofv3 wraps for the large-N legs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rx_tools_amd as R
L = R.lib(); R.check(L.rxgpu_init(0))
rng, passes = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
boxcar = int(os.environ.get("PW_BOXCAR", "1")); fir = int(os.environ.get("PW_FIR", "0"))
pl = R.plan_range(rng, 0.0, boxcar)
nn = 1 << pl.bin_e
g = torch.Generator(device="cuda"); g.manual_seed(3)
di = torch.randint(-2000, 2001, (passes, pl.tune_count, pl.buf_len), dtype=torch.int16, device="cuda", generator=g)
p2 = R.PowerScan(R.PowerParams(pl.bin_e, pl.buf_len, pl.downsample, pl.downsample_passes, boxcar, fir, 0), pl.tune_count, R.window_coefs("hamming", nn), R.sine_table(pl.bin_e))
da = torch.zeros((pl.tune_count, nn), dtype=torch.int64, device="cuda"); dsm = torch.zeros(pl.tune_count, dtype=torch.int32, device="cuda")
for _ in range(reps):
    p2.run(di.data_ptr(), passes, pl.tune_count, da.data_ptr(), dsm.data_ptr())
L.rxgpu_sync()
print("N=2^%d ds=%d tunes=%d passes=%d buf_len=%d" % (pl.bin_e, pl.downsample, pl.tune_count, passes, pl.buf_len))
