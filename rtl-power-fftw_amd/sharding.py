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


class _StagedWork:
    """What `ScanRing.pending` holds for a host-staged reduce: `.wait()` like a torch Work."""

    def __init__(self):
        import threading
        self._done = threading.Event()
        self.error = None

    def wait(self):
        self._done.wait()
        if self.error is not None:
            raise self.error
        return True


class ScanRing:
    """The exchange step of a sharded scan: a ring of blocks of `rows` x N
    double accumulators.  A rank fills the rows it owns of block k (its kernels write them),
    `submit(k)` starts ONE asynchronous reduce of the whole block onto rank `dst` -- fewer,
    larger collectives -- and the ring lets that reduce overlap the next scans' kernels;
    `begin(k)` is called before a block is written again: it waits for the block's previous
    reduce and, on `dst`, clears the block -- after a reduce `dst` holds the SUM, and the rows
    it does not own would otherwise be counted again next time round.

    Works on any torch device / backend (RCCL on GPUs in bench.py, gloo on CPU in the tests).

    `host_staged=True` is the rehearsal form for boxes with fewer GPUs than ranks (gloo has no
    reduce for device tensors): a worker thread waits for the block's kernels (an event on the
    submitting stream), copies the block to a pinned host twin, reduces the twin over a process
    group of its own -- every reduce of that group is issued by this one thread, in submit order,
    so the main thread's barriers and all-reduces cannot interleave with them differently on
    different ranks -- and, on `dst`, copies the sum back into the device block."""

    def __init__(self, rows, N, device, nring=4, dst=0, enabled=True, clear_on_reuse=True, host_staged=False):
        import torch
        self.blocks = [torch.zeros(rows, N, dtype=torch.float64, device=device) for _ in range(nring)]
        self.pending = [None] * nring
        self.used = [False] * nring
        self.dst = dst
        self.enabled = enabled
        self.clear_on_reuse = clear_on_reuse
        self.wait_seconds = 0.0        # host time spent waiting for a block's previous reduce (bench.py reports it)
        self.waits = 0
        self.host_staged = bool(host_staged and enabled)
        if self.host_staged:
            import queue
            import threading
            import torch.distributed as dist
            self._on_gpu = torch.device(device).type == "cuda"
            self._host = [torch.zeros(rows, N, dtype=torch.float64) for _ in range(nring)]
            if self._on_gpu:
                self._host = [h.pin_memory() for h in self._host]
                self._side = torch.cuda.Stream(device=device)
            self._group = dist.new_group(backend="gloo")
            self._jobs = queue.Queue()
            self._thread = threading.Thread(target=self._staging_loop, name="scan-ring-staging", daemon=True)
            self._thread.start()

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
        if self.host_staged:
            w = _StagedWork()
            ready = None
            if self._on_gpu:
                import torch
                ready = torch.cuda.Event()
                ready.record()                      # after this rank's kernels of block k on the current stream
            self._jobs.put((k, ready, w))
            if not async_op:
                w.wait()
        else:
            w = reduce_power(self.blocks[k], dst=self.dst, async_op=async_op)
        self.pending[k] = w if async_op else None
        return w

    def _staging_loop(self):
        import torch
        import torch.distributed as dist
        if self._on_gpu:
            torch.cuda.set_device(self.blocks[0].device)      # a new thread starts on device 0
        while True:
            job = self._jobs.get()
            if job is None:
                return
            k, ready, w = job
            try:
                if self._on_gpu:
                    with torch.cuda.stream(self._side):
                        self._side.wait_event(ready)
                        self._host[k].copy_(self.blocks[k], non_blocking=True)
                    self._side.synchronize()
                else:
                    self._host[k].copy_(self.blocks[k])
                dist.reduce(self._host[k], dst=self.dst, op=dist.ReduceOp.SUM, group=self._group)
                if dist.get_rank() == self.dst:
                    if self._on_gpu:
                        with torch.cuda.stream(self._side):
                            self.blocks[k].copy_(self._host[k], non_blocking=True)
                        self._side.synchronize()
                    else:
                        self.blocks[k].copy_(self._host[k])
            except Exception as exc:            # surfaces in wait(): a failed reduce must not look like a finished one
                w.error = exc
            w._done.set()

    def drain(self):
        for k in range(len(self.blocks)):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def close(self):
        """Stop the staging thread (host-staged form); the ring must be drained first."""
        if self.host_staged and self._thread is not None:
            self._jobs.put(None)
            self._thread.join()
            self._thread = None
