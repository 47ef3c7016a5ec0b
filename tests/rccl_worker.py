"""One rank of the multi-GPU rx_power path, everything through librxgpu: rxgpu_power_scan_run on this rank's tunes,
ncclGather from librccl (rxgpu_power_gather: on the library's stream; the sharded entry point puts it on the copy stream behind the scan), rank 0 feeds rxgpu_csv_dbm and compares the CSV
with the oracle's single-process sweep.  No torch.distributed: the ncclUniqueId travels through a file, the way a C
rx_power launched once per GPU would pass it.

usage: python rccl_worker.py <rank> <world> <tmpdir> [passes]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TOTAL_TUNES = 11         # not a multiple of 2, 4 or 8
RANGE = "24M:60M:1k"     # N = 4096: the register-blocked kernel, 16384 int16 per tune


def main():
    rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    import torch
    import rx_tools_amd as R
    from rx_tools_amd import shard
    from rx_tools_amd.structs import TuningState
    from support import oracle, sig_noise, PowerCfg, ptr16, ptr32, ptr64
    L = R.lib()
    # RCCL_WORKER_ONE_GPU: every rank on device 0 (only with a transport that allows it: tests/fake_rccl.c)
    dev = 0 if os.environ.get("RCCL_WORKER_ONE_GPU") else rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    R.check(L.rxgpu_init(dev))
    id_path = os.path.join(tmp, "nccl_id.bin")
    if rank == 0:
        uid = shard.Comm.unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            assert time.time() - t0 < 120, "rank 0 never published the ncclUniqueId"
            time.sleep(0.05)
        uid = open(id_path, "rb").read()
    comm = shard.Comm(uid, rank, world)

    plan = R.plan_range(RANGE)
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    data = sig_noise(passes * TOTAL_TUNES * plan.buf_len, seed=4321, amp=6000).reshape(passes, TOTAL_TUNES, plan.buf_len)
    first, count, per = shard.tune_range(rank, world, TOTAL_TUNES)
    ps = R.PowerScan(R.PowerParams(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0), per, wc, sw)
    d_in = torch.from_numpy(np.ascontiguousarray(data[:, first:first + count])).cuda() if count else torch.zeros(1, dtype=torch.int16, device="cuda")
    d_avg = torch.zeros((per, n), dtype=torch.int64, device="cuda")
    d_smp = torch.zeros(per, dtype=torch.int32, device="cuda")
    if os.environ.get("RCCL_WORKER_DIRTY_PADDING") and count < per:
        d_avg[count:] = 123456789                # the sharded entry point has to hand the root zeros for rows no tune owns
        d_smp[count:] = 77
    d_avg_all = torch.zeros((world, per, n), dtype=torch.int64, device="cuda") if rank == 0 else None
    d_smp_all = torch.zeros((world, per), dtype=torch.int32, device="cuda") if rank == 0 else None
    torch.cuda.synchronize()                  # torch's fills are on its stream, the library adds to these arrays on its own
    torch.cuda.synchronize()
    # the sharded entry point: scan of this rank's tunes on the library's stream + the gather behind it on the copy stream, no host sync between (rxgpu_sync covers both)
    R.check(L.rxgpu_power_scan_run_sharded(ps._h, comm._h, d_in.data_ptr(), passes, TOTAL_TUNES, d_avg.data_ptr(), d_smp.data_ptr(), n,
                                           d_avg_all.data_ptr() if rank == 0 else None, d_smp_all.data_ptr() if rank == 0 else None, 0))
    R.check(L.rxgpu_sync())
    if rank == 0:
        merged = d_avg_all.reshape(world * per, n)[:TOTAL_TUNES].cpu().numpy().copy()
        msmp = d_smp_all.reshape(world * per)[:TOTAL_TUNES].cpu().numpy().copy()
        O = oracle()
        cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0, ptr32(wc), ptr16(sw))
        want_avg = np.zeros((TOTAL_TUNES, n), np.int64)
        want_smp = np.zeros(TOTAL_TUNES, np.int32)
        work = np.zeros(plan.buf_len, np.int16)
        for p in range(passes):
            for t in range(TOTAL_TUNES):
                s = C.c_int(int(want_smp[t]))
                O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[p, t])), ptr16(work), ptr64(want_avg[t]), C.byref(s))
                want_smp[t] = s.value
        assert np.array_equal(merged, want_avg), "gathered avg[] rows differ from the single-process sweep"
        assert np.array_equal(msmp, want_smp)
        assert not d_avg_all.reshape(world * per, n)[TOTAL_TUNES:].any() and not d_smp_all.reshape(world * per)[TOTAL_TUNES:].any(), "padding rows not zero"
        assert comm.observed == (rank, world) and comm.gathers == 1
        # rank 0 prints the CSV rows in tune order (rtl_power.c:1047-1050) -- through the product's csv writer
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fclose.argtypes = [C.c_void_p]
        path = os.path.join(tmp, "sharded.csv")
        f = libc.fopen(path.encode(), b"wb")
        buf = C.create_string_buffer(1 << 20)
        want_rows = []
        for t in range(TOTAL_TUNES):
            s = C.c_int(int(want_smp[t]))
            O.rxo_csv_row(buf, len(buf), plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, plan.downsample, plan.crop,
                          ptr64(want_avg[t]), C.byref(s))
            want_rows.append(buf.value.decode())
            ts = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, ptr64(merged[t]), int(msmp[t]),
                             plan.downsample, plan.downsample_passes, plan.crop, None, plan.buf_len)
            L.rxgpu_csv_dbm(C.byref(ts), f)
        libc.fclose(f)
        assert open(path).read() == "".join(want_rows)
        open(os.path.join(tmp, "ok"), "w").write("world=%d rccl=%s" % (world, shard.Comm.library()))
    ps.close()
    comm.close()


if __name__ == "__main__":
    main()
