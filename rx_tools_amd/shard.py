"""Multi-GPU rx_power: contiguous tune ranges per rank and the one gather that merges them.

scanner()'s tunes are independent units (rtl_power.c:679-771: own buf16, avg, samples); nothing
crosses tunes until csv_dbm prints rows in tune order (1047-1050).  So rank r scans tunes
[r*per, min(T,(r+1)*per)) with per = ceil(T/W), and once per report interval every rank's
[per][N] int64 avg block (padded to `per` rows so the collective is fixed-size) goes to rank 0
in a single gather -- RCCL over xGMI with backend "nccl", gloo in the CPU tests.  Disjoint rows:
no reduction, no ring.  rx_fm does not shard (one stream, sequential carries).
"""


def tune_range(rank, world, total):
    """(first, count, per) of the tunes rank `rank` of `world` owns."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    hi = min(total, lo + per)
    return lo, hi - lo, per


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
