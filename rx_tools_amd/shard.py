"""Multi-GPU rx_power: contiguous tune ranges per rank and the one gather that merges them.

The product path is librxgpu's own (rxgpu_comm_*, rxgpu_power_gather in rxgpu_comm.c: one grouped ncclGather launch from
librccl on the library's stream, right behind the scan kernels); `Comm` binds it.  The torch.distributed helpers below it
are what the CPU tests (gloo) use, and bench.py only where librxgpu's own communicator cannot be created (labelled in its line).

scanner()'s tunes are independent units (rtl_power.c:679-771: own buf16, avg, samples); nothing
crosses tunes until csv_dbm prints rows in tune order (1047-1050).  So rank r scans tunes
[r*per, min(T,(r+1)*per)) with per = ceil(T/W), and once per report interval every rank's
[per][N] int64 avg block (padded to `per` rows so the collective is fixed-size) goes to rank 0
in a single gather -- RCCL over xGMI with backend "nccl", gloo in the CPU tests.  Disjoint rows:
no reduction, no ring.  rx_fm does not shard (one stream, sequential carries).
"""


import ctypes as C

from ._lib import lib, check


def tune_range(rank, world, total):
    """(first, count, per) of the tunes rank `rank` of `world` owns (rxgpu_shard_tunes)."""
    first, count, per = C.c_int(0), C.c_int(0), C.c_int(0)
    check(lib().rxgpu_shard_tunes(rank, world, total, C.byref(first), C.byref(count), C.byref(per)))
    return first.value, count.value, per.value


class Comm:
    """rxgpu_comm: an RCCL communicator owned by librxgpu (one rank per process / GPU)."""

    def __init__(self, unique_id, rank, world):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().rxgpu_comm_create(C.byref(self._h), buf, rank, world))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().rxgpu_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls):
        """Rank 0 makes the ncclUniqueId, torch.distributed (any backend) hands it round, every rank joins."""
        import torch.distributed as dist
        from ._lib import RxGpuError
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except RxGpuError as e:                     # the others are waiting in the broadcast: tell them
                box[0] = "rank 0: %s" % e
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], bytes):
            raise RxGpuError("no ncclUniqueId (%s)" % box[0])
        return cls(box[0], rank, world)

    @staticmethod
    def library():
        p = lib().rxgpu_comm_library()
        return p.decode() if p else None

    @property
    def observed(self):
        """(rank, world) as the communicator reports them (checked against ncclCommUserRank / ncclCommCount at creation)"""
        return lib().rxgpu_comm_rank(self._h), lib().rxgpu_comm_world(self._h)

    @property
    def gathers(self):
        return lib().rxgpu_comm_gathers(self._h)

    def gather(self, d_avg_local, d_samples_local, per, n_bins, d_avg_all=0, d_samples_all=0, root=0):
        """device addresses; asynchronous on the library's stream"""
        check(lib().rxgpu_power_gather(self._h, d_avg_local, d_samples_local, per, n_bins, d_avg_all or None,
                                       d_samples_all or None, root))

    def close(self):
        if self._h:
            lib().rxgpu_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_buffers(local, dst=0):
    """Receive buffers for gather_rows on dst (allocate once, reuse every report interval)."""
    import torch
    import torch.distributed as dist
    if dist.get_rank() != dst:
        return None
    return [torch.zeros_like(local) for _ in range(dist.get_world_size())]


def gather_rows(local, dst=0, out=None):
    """local: [per, N] tensor (any device).  Returns the list of per-rank blocks on dst, else None."""
    import torch.distributed as dist
    if out is None:
        out = gather_buffers(local, dst)
    dist.gather(local, out if dist.get_rank() == dst else None, dst=dst)
    return out


def merge_rows(blocks, total):
    """Concatenate the gathered [per, N] blocks and drop the padding rows."""
    import torch
    return torch.cat(blocks, dim=0)[:total]
