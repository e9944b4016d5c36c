#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X power-spectrum engine.

Metric (BASELINE.json): complex IQ samples/s through the fused
unpack + FFT + |X|^2-accumulate path, inputs resident in HBM.

A "step" is one acquisition of config C2 (N=4096 bins, 10000 repeats,
rectangular window, synthetic receiver-like u8 IQ): the fused
kernel K1 over the 81 920 000-byte stream plus the partial-spectrum reduce K3.
Successive steps walk a ring of distinct replay buffers whose total size exceeds
the 256 MiB Infinity Cache, so every step reads its bytes from HBM.

N>1 (launched by torch.distributed.run, one rank per GPU): every rank owns an
independent shard of the frames (weak scaling, per-GPU work fixed) and the
per-bin accumulators of 8 consecutive steps (the hops of one scan) are summed onto
rank 0 with ONE asynchronous RCCL reduce of 8 x 4096 doubles that overlaps the
following steps' kernels.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BINS = 4096
REPEATS = 10000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
CPU_BASELINE_SECONDS = 10.0


def load_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "librpf_oracle.so"))
    u8p = ctypes.POINTER(ctypes.c_uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    fp = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.rpf_oracle_accumulate.argtypes = [ctypes.c_int, fp, ctypes.c_int, u8p, ctypes.c_size_t,
                                          ctypes.c_int64, dp, i64p]
    lib.rpf_oracle_accumulate_mt.argtypes = [ctypes.c_int, fp, u8p, ctypes.c_size_t, ctypes.c_int64,
                                             ctypes.c_int, dp, i64p]
    return lib


def cpu_baseline(stream, pwr_gpu, window=None):
    """Time the CPU restatement (oracle, kind 'port') on this box's host cores on
    a bounded sample of the same workload, and check the GPU result against it."""
    lib = load_oracle()
    u8p = ctypes.POINTER(ctypes.c_uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    wp = window.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if window is not None else None
    pwr = np.zeros(N_BINS)
    done = ctypes.c_int64()
    passes = 0
    t0 = time.perf_counter()
    while True:
        rc = lib.rpf_oracle_accumulate(N_BINS, wp, 32, stream.ctypes.data_as(u8p), stream.size,
                                       REPEATS, pwr.ctypes.data_as(dp), ctypes.byref(done))
        assert rc == 0 and done.value == REPEATS
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= CPU_BASELINE_SECONDS:
            break
    one = N_BINS * REPEATS * passes / dt
    rel = float(np.max(np.abs(pwr_gpu - pwr) / pwr))
    # all host cores (disjoint frame ranges; not the reference's structure)
    cores = os.cpu_count() or 1
    pwr_mt = np.zeros(N_BINS)
    t0 = time.perf_counter()
    p2 = 0
    while True:
        rc = lib.rpf_oracle_accumulate_mt(N_BINS, wp, stream.ctypes.data_as(u8p), stream.size,
                                          REPEATS, cores, pwr_mt.ctypes.data_as(dp), ctypes.byref(done))
        assert rc == 0
        p2 += 1
        dt2 = time.perf_counter() - t0
        if dt2 >= CPU_BASELINE_SECONDS / 2:
            break
    allc = N_BINS * REPEATS * p2 / dt2
    return {
        "value": one, "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "C2 stream (%d frames x %d bins) replayed %d times through oracle/rpf_oracle.c, "
                  "1 thread like the reference's single FFT thread" % (REPEATS, N_BINS, passes),
        "all_cores_value": allc, "all_cores": cores,
        "gpu_vs_cpu_max_rel_err": rel,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--replay-buffers", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the per-step reduce even with one rank")
    ap.add_argument("--workload", choices=["C2", "C3"], default="C2",
                    help="C2 (default, the metric's configuration): rectangular window; C3: periodic Hann window")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket the fused kernel with HIP events on every k-th timed step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rtl_power_fftw_amd as rpf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank if use_dist else 0)
    torch.cuda.set_device(dev)

    # ---- workload: this rank's shard of the frames (seeded per rank) ----------
    stream_bytes = 2 * N_BINS * REPEATS
    base = rpf.synth.noise_tones_iq(2 + 1000 * rank, N_BINS * REPEATS)
    d_base = torch.from_numpy(base).to(dev)
    nb = max(1, args.replay_buffers)
    # further replay buffers = the same stream rotated by whole frames (distinct
    # addresses and byte order, same statistics)
    bufs = [d_base] + [torch.roll(d_base, shifts=2 * N_BINS * (37 * i)) for i in range(1, nb)]

    window = rpf.synth.hann_window(N_BINS) if args.workload == "C3" else None
    ds = rpf.Datastore(rpf.Params(N=N_BINS, window=window is not None, repeats=REPEATS), window,
                       device=dev.index or 0)
    # Multi-GPU exchange: the spectra of HOPS consecutive steps (= the
    # hops of one scan, config C5 has 8) meet in ONE reduce of HOPS*N doubles --
    # fewer, larger collectives -- issued asynchronously on a ring of blocks so that
    # it overlaps the following steps' kernels.
    HOPS = 8
    nring = 4
    d_pwr = [torch.zeros(HOPS, N_BINS, dtype=torch.float64, device=dev) for _ in range(nring)]
    s = torch.cuda.current_stream().cuda_stream
    pending = [None] * nring

    def step(i, ev=None):
        blk, hop = (i // HOPS) % nring, i % HOPS
        if hop == 0 and pending[blk] is not None:
            pending[blk].wait()
            pending[blk] = None
        if ev is not None:
            ev[0].record()
        ds.device_fused(bufs[i % nb].data_ptr(), stream_bytes, REPEATS, s)
        if ev is not None:
            ev[1].record()
        ds.device_reduce(d_pwr[blk][hop].data_ptr(), s)
        if use_dist and hop == HOPS - 1:
            pending[blk] = dist.reduce(d_pwr[blk], dst=0, op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for k in range(nring):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def fence(last_step=None):
        # a scan cut short by the step count still owes its (partial) reduce
        if use_dist and last_step is not None and last_step % HOPS != HOPS - 1:
            blk = (last_step // HOPS) % nring
            dist.reduce(d_pwr[blk], dst=0, op=dist.ReduceOp.SUM)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # this rank's own spectrum of buffer 0 (checked against the CPU oracle below)
    d_chk = torch.zeros(N_BINS, dtype=torch.float64, device=dev)
    ds.accumulate_device(bufs[0].data_ptr(), stream_bytes, REPEATS, d_chk.data_ptr(), s)
    torch.cuda.synchronize()
    pwr_first = d_chk.cpu().numpy().copy()

    for i in range(args.warmup):
        step(i)
    fence(args.warmup - 1 if args.warmup else None)

    events = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev = None
        if args.event_every > 0 and i % args.event_every == 0:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            events.append(ev)
        step(i, ev)
    fence(args.steps - 1)
    elapsed = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * N_BINS * REPEATS * args.steps / elapsed
        k1_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else None
        alg_bytes = 2 * N_BINS * REPEATS + 8 * N_BINS + (4 * N_BINS if window is not None else 0)
        info = ds.launch_info()
        roof = None
        if k1_ms:
            achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
            traffic = measured_peak = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("fft_accum_%s_hbm_bytes_per_launch" % args.workload.lower())
                    measured_peak = tj.get("measured_read_only_GBps")   # tools/hbm_read_bench.hip, same box type
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "fft_accum_kernel<N=4096,P=16>", "kernel_ms": k1_ms,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "samples_per_s_kernel_only": N_BINS * REPEATS / (k1_ms * 1e-3)}
            if measured_peak:
                roof["measured_read_only_peak"] = measured_peak
                roof["frac_of_measured_read_only"] = achieved / measured_peak
        out = {
            "metric": "IQ samples/s through FFT+|X|^2-accumulate",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: N=4096 bins x 10000 repeats per step per GPU, %s window, " % (
                args.workload, "periodic Hann" if window is not None else "rectangular") +
                                   "u8 IQ resident in HBM (%d replay buffers of %d B)" % (nb, stream_bytes),
                       "launch": info, "reduce": "one async RCCL reduce of 8 x 4096 f64 bins per 8 steps" if use_dist else "none"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(base, pwr_first, window)
        print(json.dumps(out), flush=True)

    ds.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
