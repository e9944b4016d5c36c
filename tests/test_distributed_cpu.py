"""N>1 path on CPU: world_size-2 gloo processes shard an 8-hop scan hop-major at
frame boundaries, accumulate their shards (with the oracle standing in for the
kernel -- there is no GPU here) and meet in one reduce per hop, exactly the
structure bench.py / the engine use with RCCL.  (-m "not gpu")"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rtl_power_fftw_amd as rpf
from helpers import max_rel, oracle_accumulate


def test_shard_frames_partitions_exactly():
    for total in (0, 1, 7, 10000, 40000):
        for world in (1, 2, 3, 4, 8):
            got = [rpf.sharding.shard_frames(total, world, r) for r in range(world)]
            assert sum(c for _, c in got) == total
            pos = 0
            for first, count in got:
                assert first == pos
                pos += count
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def test_shard_hops_is_hop_major_and_frame_aligned():
    # config C5: 8 hops x 5000 frames; 8 ranks own one hop each, 2 ranks own 4 each
    for world in (1, 2, 4, 8):
        pieces = [rpf.sharding.shard_hops(8, 5000, world, r) for r in range(world)]
        flat = [p for ps in pieces for p in ps]
        assert sum(n for _, _, n in flat) == 40000
        if world == 8:
            assert pieces == [[(h, 0, 5000)] for h in range(8)]
    # a world size that does not divide the hops cuts inside a hop, at a frame boundary
    pieces = [rpf.sharding.shard_hops(2, 10, 3, r) for r in range(3)]
    assert pieces == [[(0, 0, 7)], [(0, 7, 3), (1, 0, 4)], [(1, 4, 6)]]


N, HOPS, FRAMES = 256, 4, 30


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = torch.zeros(HOPS, N, dtype=torch.float64)
    for hop, first, count in rpf.sharding.shard_hops(HOPS, FRAMES, world, rank):
        stream = rpf.synth.noise_tones_iq(50 + hop, N * FRAMES)
        part, done = oracle_accumulate(N, stream[2 * N * first: 2 * N * (first + count)], count)
        assert done == count
        acc[hop] += torch.from_numpy(part)
    works = [rpf.sharding.reduce_power(acc[h], dst=0, async_op=True) for h in range(HOPS)]
    for w in works:
        w.wait()
    if rank == 0:
        np.save(out, acc.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_scan_equals_single_process(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    for hop in range(HOPS):
        stream = rpf.synth.noise_tones_iq(50 + hop, N * FRAMES)
        want, _ = oracle_accumulate(N, stream, FRAMES)
        # double addition is not associative: the sharded sum differs at 1e-16 level
        assert max_rel(got[hop], want) < 1e-13


def _ring_worker(rank, world, port, out, host_staged=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hops, n, scans = 8, 16, 11                          # more scans than ring slots: blocks are reused
    ring = rpf.sharding.ScanRing(hops, n, "cpu", nring=3, dst=0, host_staged=host_staged)
    mine = rpf.sharding.shard_hops(hops, 10, world, rank)
    for s in range(scans):
        k = s % len(ring)
        blk = ring.begin(k)
        for hop, first, count in mine:                  # this rank's share of hop `hop`: count/10 of its power
            blk[hop] = float(s + 1) * (hop + 1) * count / 10.0
        ring.submit(k)
        if host_staged and s % 4 == 1:                  # the main thread's own collectives, between the staging
            t = torch.ones(1)                           # thread's reduces (bench.py: flags, region times, barriers)
            dist.all_reduce(t)
            assert t.item() == world
        if s >= 2:                                      # read a scan that is two submits old (still in the ring)
            j = (s - 2) % len(ring)
            ring.pending[j] and ring.pending[j].wait()
            if rank == 0:
                want = float(s - 1) * (torch.arange(hops, dtype=torch.float64) + 1)
                assert torch.allclose(ring.blocks[j][:, 0], want, rtol=1e-13), (s, ring.blocks[j][:, 0], want)
    ring.drain()
    ring.close()
    if rank == 0:
        np.save(out, ring.blocks[(scans - 1) % len(ring)].numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("host_staged", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_scan_ring_reuse_does_not_count_rows_twice(tmp_path, world, host_staged):
    """bench.py's exchange (sharding.ScanRing): after a reduce rank 0 holds the SUM; the rows it
    does not own must be cleared before the block is filled again, or every reuse of a ring
    slot would add the previous scan's spectra once more."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ring.npy")
    # host_staged: bench.py's one-GPU rehearsal form (--dist-backend gloo --share-device): a staging thread
    # issues the reduces on a process group of its own while the main thread keeps using the default one
    mp.spawn(_ring_worker, args=(world, port, out, host_staged), nprocs=world, join=True)
    got = np.load(out)
    assert np.allclose(got[:, 0], 11.0 * (np.arange(8) + 1), rtol=1e-13)
