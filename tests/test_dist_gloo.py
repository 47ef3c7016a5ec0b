"""CPU, 2 processes over gloo: the multi-GPU rx_power path (tune sharding + one gather + rank-0
csv_dbm) gives the same CSV rows as one process scanning every tune.  The per-rank avg[] rows
come from the oracle here (no GPU in this container); on the GPU box bench.py runs the same
shard/gather code with librxgpu filling the rows and backend "nccl" (= RCCL)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import rx_tools_amd as R
from rx_tools_amd import shard
from rx_tools_amd.structs import TuningState
from support import oracle, sig_noise, PowerCfg, ptr16, ptr32, ptr64

TOTAL_TUNES = 7          # deliberately not a multiple of the world size
RANGE = "88M:108M:125k"  # N = 32, 16384 int16 per tune


def scan_rows(first, count, per, plan, data):
    O = oracle()
    n = 1 << plan.bin_e
    wc, sw = R.window_coefs("hamming", n), R.sine_table(plan.bin_e)
    cfg = PowerCfg(plan.bin_e, plan.buf_len, plan.downsample, plan.downsample_passes, 1, 0, 0, ptr32(wc), ptr16(sw))
    avg = np.zeros((per, n), np.int64)
    samples = np.zeros(per, np.int32)
    work = np.zeros(plan.buf_len, np.int16)
    for i in range(count):
        t = first + i
        s = C.c_int(0)
        O.rxo_power_tune(C.byref(cfg), ptr16(np.ascontiguousarray(data[t])), ptr16(work), ptr64(avg[i]), C.byref(s))
        samples[i] = s.value
    return avg, samples


def csv_rows(avg, samples, plan, path):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fclose.argtypes = [C.c_void_p]
    f = libc.fopen(path.encode(), b"wb")
    for t in range(avg.shape[0]):
        ts = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, ptr64(avg[t]), int(samples[t]),
                         plan.downsample, plan.downsample_passes, plan.crop, None, plan.buf_len)
        R.lib().rxgpu_csv_dbm(C.byref(ts), f)
    libc.fclose(f)
    return open(path).read()


def worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = R.plan_range(RANGE)
    data = sig_noise(TOTAL_TUNES * plan.buf_len, seed=99, amp=3000).reshape(TOTAL_TUNES, plan.buf_len)
    first, count, per = shard.tune_range(rank, world, TOTAL_TUNES)
    avg, samples = scan_rows(first, count, per, plan, data)
    blocks = shard.gather_rows(torch.from_numpy(avg))
    sblocks = shard.gather_rows(torch.from_numpy(samples.astype(np.int64)).unsqueeze(1))
    if rank == 0:
        merged = shard.merge_rows(blocks, TOTAL_TUNES).numpy().copy()
        msamples = shard.merge_rows(sblocks, TOTAL_TUNES).numpy()[:, 0]
        got = csv_rows(merged, msamples, plan, os.path.join(tmp, "sharded.csv"))
        full_avg, full_samples = scan_rows(0, TOTAL_TUNES, TOTAL_TUNES, plan, data)
        want = csv_rows(full_avg, full_samples, plan, os.path.join(tmp, "single.csv"))
        assert got == want and got.count("\n") == TOTAL_TUNES
        open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_tune_ranges_cover_everything():
    for total in (1, 7, 599, 600):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, cnt, per = shard.tune_range(r, world, total)
                assert cnt <= per
                got += list(range(lo, lo + cnt))
            assert got == list(range(total))
    assert [shard.tune_range(r, 8, 599)[1] for r in range(8)] == [75] * 7 + [74]


@pytest.mark.timeout(300)
def test_sharded_scan_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
