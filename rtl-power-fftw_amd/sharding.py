"""Multi-GPU partitioning of an acquisition.

`pwr` is a plain sum over frames (/root/reference/src/datastore.cxx:83-85) and
every hop has its own `pwr` (acquisition.cxx:252), so the path shards with no
data-path collective: work is split hop-major into contiguous, frame-aligned
ranges, each rank runs the fused kernel on its range, and the per-bin double
accumulators meet in ONE reduce per hop (RCCL over xGMI when the tensors are on
GPUs, gloo in the CPU tests).  The message is hops*N doubles -- latency-bound, so
it is issued asynchronously and overlapped with the next hop's kernel.
"""


def shard_frames(total_frames, world_size, rank):
    """Contiguous frame range [first, first+count) of `rank`: the cut between rank
    r-1 and rank r is at frame ceil(total*r/G), so frame k goes to rank
    floor(k*G/total) and cuts fall only on multiples of 2N bytes."""
    first = (total_frames * rank + world_size - 1) // world_size
    end = (total_frames * (rank + 1) + world_size - 1) // world_size
    return first, end - first


def shard_hops(n_hops, frames_per_hop, world_size, rank):
    """Hop-major partition: list of (hop, first_frame, count) for `rank`.  With
    world_size dividing n_hops every rank owns whole hops (no arithmetic reduce
    needed for them); otherwise hops are cut at frame boundaries."""
    total = n_hops * frames_per_hop
    first, count = shard_frames(total, world_size, rank)
    out = []
    pos, end = first, first + count
    while pos < end:
        hop = pos // frames_per_hop
        in_hop = pos - hop * frames_per_hop
        n = min(frames_per_hop - in_hop, end - pos)
        out.append((hop, in_hop, n))
        pos += n
    return out


def reduce_power(pwr, dst=0, group=None, async_op=False):
    """Sum per-bin accumulators onto rank `dst` (torch tensor, any backend)."""
    import torch.distributed as dist
    return dist.reduce(pwr, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
