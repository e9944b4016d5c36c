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


class ScanRing:
    """The exchange step of a sharded scan: a ring of blocks of `rows` x N
    double accumulators.  A rank fills the rows it owns of block k (its kernels write them),
    `submit(k)` starts ONE asynchronous reduce of the whole block onto rank `dst` -- fewer,
    larger collectives -- and the ring lets that reduce overlap the next scans' kernels;
    `begin(k)` is called before a block is written again: it waits for the block's previous
    reduce and, on `dst`, clears the block -- after a reduce `dst` holds the SUM, and the rows
    it does not own would otherwise be counted again next time round.

    Works on any torch device / backend (RCCL on GPUs in bench.py, gloo on CPU in the tests)."""

    def __init__(self, rows, N, device, nring=4, dst=0, enabled=True, clear_on_reuse=True):
        import torch
        self.blocks = [torch.zeros(rows, N, dtype=torch.float64, device=device) for _ in range(nring)]
        self.pending = [None] * nring
        self.used = [False] * nring
        self.dst = dst
        self.enabled = enabled
        self.clear_on_reuse = clear_on_reuse
        self.wait_seconds = 0.0        # host time spent waiting for a block's previous reduce (bench.py reports it)
        self.waits = 0

    def __len__(self):
        return len(self.blocks)

    def begin(self, k):
        """Block k is about to be written for a new scan."""
        if self.pending[k] is not None:
            import time
            t0 = time.perf_counter()
            self.pending[k].wait()
            self.wait_seconds += time.perf_counter() - t0
            self.waits += 1
            self.pending[k] = None
        if self.enabled and self.clear_on_reuse and self.used[k]:
            import torch.distributed as dist
            if dist.get_rank() == self.dst:
                self.blocks[k].zero_()
        self.used[k] = True
        return self.blocks[k]

    def submit(self, k, async_op=True):
        """All of this rank's rows of block k are written (or enqueued on the current stream)."""
        if not self.enabled:
            return None
        w = reduce_power(self.blocks[k], dst=self.dst, async_op=async_op)
        self.pending[k] = w if async_op else None
        return w

    def drain(self):
        for k in range(len(self.blocks)):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None
