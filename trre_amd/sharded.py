"""Line-sharded scan over several GPUs of one node (one process per GPU).

The reference's scan path has no exchange step — a line's output depends on that
line and the read-only program (trre_nft.c:776-790) — so the input is cut at
'\\n' boundaries into one contiguous shard per rank, every rank scans its shard
on its own GPU, and the outputs are concatenated in rank order.  No data-path
collective: the only communication is an all_gather of the shard output SIZES
(one int64 per rank) so that each rank knows its offset in the global output.
"""
import torch
import torch.distributed as dist

from .api import shard_bounds


def shard_of(data, rank, world):
    """This rank's contiguous byte range of `data` (cut just past a '\\n')."""
    b = shard_bounds(data, world)
    return b[rank], b[rank + 1]


def scan_sharded(data, scan_fn, group=None):
    """Scan `data` (bytes, identical on all ranks) line-sharded over the process
    group.  `scan_fn(bytes) -> bytes` runs one shard (normally Program.scan /
    Program.scan_tensor on this rank's GPU).  Returns (offset, shard_output,
    total_size): where this rank's output sits in the global output."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_of(data, rank, world)
    out = scan_fn(data[lo:hi])
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    mine = torch.tensor([len(out)], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        sizes = [s.to(dev) for s in sizes]
        mine = mine.to(dev)
    dist.all_gather(sizes, mine, group=group)
    sizes = [int(s.item()) for s in sizes]
    return sum(sizes[:rank]), out, sum(sizes)
