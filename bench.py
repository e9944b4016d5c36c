#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X power-spectrum engine.

Metric (BASELINE.json): complex IQ samples/s through the fused
unpack + FFT + |X|^2-accumulate path, inputs resident in HBM.

Workloads (--workload, default "auto" = C2 on one GPU, C5 on several):
  C2  one step = one acquisition, N=4096 bins x 10000 repeats, rectangular window
      (BASELINE.json's metric configuration; the N=1 headline)
  C3  the same stream under a periodic Hann window
  C4  one step = one acquisition, N=262144 bins x 1000 repeats (four-step kernels)
  C5  one step = one 8-hop scan, N=4096 x 5000 repeats per hop, hop h = seed 50+h.
      STRONG scaling: the 8 x 5000 frames of a scan are dealt hop-major to the ranks
      (sharding.shard_hops), every rank runs ONE launch of the fused kernel over its hops' frame
      ranges (rpf_accumulate_device_hops: accumulators handed over and zeroed at hop boundaries) and the
      per-bin accumulators meet in ONE asynchronous RCCL reduce of 8 x 4096 doubles per
      scan, overlapped with the next scan's kernels.  Rank 0 checks the reduced spectra
      against the committed float64 fixtures (tests/golden/c5_hop*.npz) before printing.
With C2/C3/C4 on several GPUs every rank runs its own acquisitions (weak scaling).

Successive steps walk a ring of distinct replay buffers whose total size exceeds the
256 MiB Infinity Cache, so every step reads its bytes from HBM.

Timing: an untimed pre-warm of >= 0.2 s (clocks, caches, lazy HIP state), then W warm-up
steps, then EXACTLY K steps between barrier + synchronize; when one such region is shorter
than 0.5 s it is repeated and the MEDIAN region is reported (regions listed in
"timing"), so a 20-step run reports what a 2000-step run reports.

Ranks: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N
ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
arguments>`, one rank per device) and passes rank 0's line through; under torchrun (the driver's
multi-GPU form) the environment's ranks are used as they are.  Fewer devices than ranks is an
error (exit code 2), never a silent one-GPU measurement -- unless `--dist-backend gloo
--share-device` asks for the REHEARSAL: every rank on device 0, the per-scan reduce staged through
the host (sharding.ScanRing, host_staged), so that launcher, shards, ring reuse, per-rank
reporting and the fixture check all run on the real kernels of a one-GPU box; its `value` is N
ranks time-slicing one GPU and says nothing about scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector FP32 peak (FMA counted as 2)
CLOCK_GHZ = 2.4                # peak shader clock used for the issue-slot fractions
N_CUS = 256
CPU_BASELINE_SECONDS = 10.0
PREWARM_SECONDS = 0.2
MIN_REGION_SECONDS = 0.5
MAX_REGIONS = 25

WORKLOADS = {
    "C2": dict(N=4096, R=10000, window=False, seed=2),
    "C3": dict(N=4096, R=10000, window=True, seed=2),
    "C4": dict(N=262144, R=1000, window=False, seed=4),
    "C5": dict(N=4096, R=5000, window=False, seed=50, hops=8),
}


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


def cpu_baseline(N, R, stream, pwr_gpu, window=None):
    """Time the CPU restatement (oracle, kind 'port') on this box's host cores on
    a bounded sample of the same workload, check the GPU result against it, and --
    if this box has the reference's real FFT provider -- real FFTW next to both."""
    lib = load_oracle()
    u8p = ctypes.POINTER(ctypes.c_uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    wp = window.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if window is not None else None
    pwr = np.zeros(N)
    done = ctypes.c_int64()
    passes = 0
    t0 = time.perf_counter()
    while True:
        rc = lib.rpf_oracle_accumulate(N, wp, 32, stream.ctypes.data_as(u8p), stream.size,
                                       R, pwr.ctypes.data_as(dp), ctypes.byref(done))
        assert rc == 0 and done.value == R
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= CPU_BASELINE_SECONDS:
            break
    one = N * R * passes / dt
    rel = float(np.max(np.abs(pwr_gpu - pwr) / pwr))
    # all host cores (disjoint frame ranges; not the reference's structure): one persistent worker (plan)
    # per thread, every thread walks its frames `loops` times -- thread start-up and planning are not what
    # is being compared with the GPU
    # (the cores this process may run on: affinity mask and cgroup CPU quota -- a container's share, not the
    # machine's nameplate; the GPU boxes show 256 CPUs and grant 16)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    lib.rpf_oracle_accumulate_mt_loops.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float), u8p, ctypes.c_size_t,
                                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int, dp,
                                                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    pwr_mt = np.zeros(N)
    secs = ctypes.c_double()
    loops = 1
    while True:
        rc = lib.rpf_oracle_accumulate_mt_loops(N, wp, stream.ctypes.data_as(u8p), stream.size, R, cores, loops,
                                                pwr_mt.ctypes.data_as(dp), ctypes.byref(done), ctypes.byref(secs))
        assert rc == 0
        if secs.value >= CPU_BASELINE_SECONDS / 4 or loops >= 1 << 14:
            break
        loops = int(min(1 << 14, max(2 * loops, loops * CPU_BASELINE_SECONDS / 3 / max(secs.value, 1e-4))))
    allc = N * R * loops / secs.value
    assert float(np.max(np.abs(pwr_mt - pwr) / pwr)) < 1e-12      # the same frames, another grouping of the sums
    out = {
        "value": one, "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "the step's stream (%d frames x %d bins) replayed %d times through oracle/rpf_oracle.c, "
                  "1 thread like the reference's single FFT thread" % (R, N, passes),
        "all_cores_value": allc, "all_cores": cores, "machine_cpus": os.cpu_count(),
        "all_cores_sample": "%d threads, one persistent plan each, %d walks over the step's stream" % (cores, loops),
        "gpu_vs_cpu_max_rel_err": rel,
    }
    if rel >= 1e-6:
        # C4's stream: a bin of the 16 deterministic lines, where the CPU path's float32 last pass is itself 1.59e-6 from
        # float64 truth and the GPU (last pass in double, exact twiddles before it) 1.3e-7 -- tests/parity_bars.py section 4,
        # profiles/r06_fullsize_errors.json "c4"; every other bin is reported beside them
        err = np.abs(pwr_gpu - pwr) / pwr
        over = err >= 1e-6
        out["gpu_vs_cpu_bins_at_or_over_1e-6"] = int(over.sum())
        out["gpu_vs_cpu_max_rel_err_other_bins"] = float(err[~over].max())
        out["gpu_vs_cpu_note"] = ("the bins over 1e-6 are where the CPU path itself is that far from float64 truth "
                                  "(profiles/r06_fullsize_errors.json, tests/test_gpu_fullsize.py asserts the GPU against the truth there)")
    # real FFTW (the reference's FFT, datastore.cxx:30-33,82), if this box has it: first 400 frames
    try:
        from oracle import fftw_probe
        head = min(R, 400)
        head_bytes = stream[: 2 * N * head]
        o_head = np.zeros(N)
        lib.rpf_oracle_accumulate(N, wp, 32, head_bytes.ctypes.data_as(u8p), head_bytes.size, head,
                                  o_head.ctypes.data_as(dp), ctypes.byref(done))
        out["fftw"] = fftw_probe.report(N, head_bytes, head, {"oracle": o_head}, window)
    except Exception as exc:          # the probe must never take the benchmark down
        out["fftw"] = {"fftw": "probe failed: %r" % (exc,)}
    # a third float32 FFT of independent provenance on every box: Intel MKL's DFTI (torch.fft on CPU tensors;
    # MKL also ships the FFTW3 interface a reference build can link instead of libfftw3f)
    try:
        from oracle import mkl_probe
        head = min(R, 400)
        out["mkl"] = mkl_probe.report(N, stream[: 2 * N * head], head, {"oracle": o_head, "gpu_first_frames": None}, window)
    except Exception as exc:
        out["mkl"] = {"mkl": "probe failed: %r" % (exc,)}
    # ... and a fourth: pocketfft in single precision (scipy.fft on complex64 input)
    try:
        from oracle import pocketfft_probe
        head = min(R, 400)
        out["pocketfft"] = pocketfft_probe.report(N, stream[: 2 * N * head], head, {"oracle": o_head}, window)
    except Exception as exc:
        out["pocketfft"] = {"pocketfft": "probe failed: %r" % (exc,)}
    return out


def end_to_end(rpf, N, R, stream, window, device):
    """(beta) The same stream through the reference's buffer hand-off: pinned host buffers ->
    hipMemcpyAsync -> fused kernel, copies overlapped with compute.  `replay`: rpf_accumulate on a host
    stream the caller has pinned (rpf_stream_register) -- its bytes go to the device from where they lie;
    `replay_through_memcpy`: the same without pinning (the producer memcpys the stream into the pool's
    buffers, what round 4 reported as `replay`); `resident`: the pinned buffers already hold data
    (acquire/submit only: engine + PCIe)."""
    out = []
    for label, buf_length, buffers in (("reference default: 5 x 1638400 B", 1638400, 5),
                                       ("one large -s: 5 x 104857600 B", 104857600, 5)):
        try:
            with rpf.Datastore(rpf.Params(N=N, window=window is not None, repeats=R, buf_length=buf_length,
                                          buffers=buffers), window, device=device) as ds:
                nbytes = 2 * N * R
                ds.accumulate(stream, R)                                  # warm
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    _, done = ds.accumulate(stream, R)
                t_copied = (time.perf_counter() - t0) / reps
                assert done == R
                # the same stream pinned where it lies (rpf_stream_register): no memcpy into the pool
                t0 = time.perf_counter()
                ds.register_stream(stream)
                t_register = time.perf_counter() - t0
                t0 = time.perf_counter()
                ref, done = ds.accumulate(stream, R)                      # first pass over freshly pinned memory (slow: mappings)
                t_first = time.perf_counter() - t0
                t0 = time.perf_counter()
                for _ in range(reps):
                    got, done = ds.accumulate(stream, R)
                t_replay = (time.perf_counter() - t0) / reps
                ds.unregister_stream(stream)
                assert done == R and np.array_equal(got, ref)
                # pinned buffers filled once (untimed pass), then only handed over (timed pass)
                total_frames = 4 * R
                need = 2 * N * total_frames

                # (raw ctypes calls in the timed pass: a numpy view per acquire costs a third of a 1.6 MB buffer's 30 us)
                lib, handle = ds._lib, ds._handle
                ptr, cap = ctypes.c_void_p(), ctypes.c_size_t()

                def hand_over(fill):
                    ds.begin(total_frames)
                    filled = set()
                    sent = 0
                    while sent < need:
                        if fill:
                            buf = ds.acquire()
                            if buf.ctypes.data not in filled:
                                off = ((len(filled) * buf_length) % max(1, nbytes - buf_length)) & ~1
                                n0 = min(buf_length, nbytes - off)
                                buf[:n0] = stream[off:off + n0]
                                filled.add(buf.ctypes.data)
                            ds.submit(buf, min(buf_length, need - sent))
                        else:
                            assert lib.rpf_buffer_acquire(handle, ctypes.byref(ptr), ctypes.byref(cap)) == 0
                            assert lib.rpf_buffer_submit(handle, ptr, min(buf_length, need - sent)) == 0
                        sent += min(buf_length, need - sent)
                    return ds.finish()

                hand_over(True)
                t0 = time.perf_counter()
                done = hand_over(False)
                t_res = time.perf_counter() - t0
                assert done == total_frames
            out.append({"buffers": label,
                        "replay_samples_per_s": N * R / t_replay, "replay_GBps": nbytes / t_replay / 1e9,
                        "replay": "the host stream pinned once (rpf_stream_register: %.1f ms; first pass over it %.1f ms), then "
                                  "replayed without a copy into the pool" % (t_register * 1e3, t_first * 1e3),
                        "replay_through_memcpy_samples_per_s": N * R / t_copied,
                        "resident_samples_per_s": N * total_frames / t_res,
                        "resident_pcie_GBps": need / t_res / 1e9})
        except Exception as exc:
            out.append({"buffers": label, "error": repr(exc)})
    return {"what": "pinned host buffers -> hipMemcpyAsync overlapped with the fused kernel (queue path of "
                    "include/rpf_engine.h); never the headline value. PCIe Gen5 x16 spec 63 GB/s.",
            "cases": out}


def secondary_limits(N, frames_per_launch, kernel_s):
    """FP32-VALU and LDS fractions of the dominant kernel beside the HBM figure (SURVEY.md 8d:
    'secondary limiters side by side').  Nominal flops 5 N log2 N per frame; issue-slot
    figures from the instruction mix of the compiled loop (profiles/isa_mix.json, written by
    tools/isa_mix.py from the shipped library's disassembly) x the guide's cycles per instruction."""
    out = {}
    flops = 5.0 * N * math.log2(N) * frames_per_launch
    out["fp32_nominal_tflops"] = flops / kernel_s / 1e12
    out["fp32_nominal_frac_of_valu_peak"] = out["fp32_nominal_tflops"] / FP32_VALU_PEAK_TFLOPS
    path = os.path.join(ROOT, "profiles", "isa_mix.json")
    try:
        mix = json.load(open(path)).get(str(N))
    except Exception:
        mix = None
    if mix:
        # one loop iteration = mix["frames_per_iteration"] frames on mix["waves"] waves of one workgroup
        iters = frames_per_launch / mix["frames_per_iteration"]
        simd_cycles_avail = kernel_s * CLOCK_GHZ * 1e9 * N_CUS * 4
        lds_cycles_avail = kernel_s * CLOCK_GHZ * 1e9 * N_CUS
        out["valu_issue_frac"] = mix["valu_cycles_per_wave_iteration"] * mix["waves"] * iters / simd_cycles_avail
        out["lds_frac"] = mix["lds_cycles_per_wave_iteration"] * mix["waves"] * iters / lds_cycles_avail
        out["source"] = "profiles/isa_mix.json (%s) x MI355X_MICROARCH.md cycle tables at %.1f GHz" % (
            mix.get("kernel", "?"), CLOCK_GHZ)
        # the same VALU work priced at the packed-f32 issue rate this chip SUSTAINS with every SIMD busy
        # (tools/lds_valu_bench, profiles/valu_rate.json): the clock sags well below 2.4 GHz under that load
        try:
            rate = json.load(open(os.path.join(ROOT, "profiles", "valu_rate.json")))
            pk_ns = rate["pk_fma_ns_per_instruction_per_simd"]
            valu_s = mix["valu_cycles_per_wave_iteration"] / 4.0 * pk_ns * 1e-9 * mix["waves"] * iters / (N_CUS * 4)
            out["valu_frac_of_sustained_pk_rate"] = valu_s / kernel_s
            out["sustained_pk_rate_source"] = "profiles/valu_rate.json: %.3f ns per v_pk_fma_f32 per SIMD (captured %s)" % (
                pk_ns, rate.get("captured"))
        except Exception:
            pass
    return out


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks of the job, one per device; > 1 without WORLD_SIZE in the environment: bench.py "
                         "starts them itself under torch.distributed.run")
    ap.add_argument("--steps", type=int, default=None, help="default: 2000 (C2/C3), 200 (C5), 100 (C4)")
    ap.add_argument("--warmup", type=int, default=None, help="default: steps / 20")
    ap.add_argument("--replay-buffers", type=int, default=0, help="0 = enough to exceed the Infinity Cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the reduce even with one rank")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (the product's exchange); gloo = the reduce staged through the host "
                         "(rehearsal on boxes with fewer GPUs than ranks)")
    ap.add_argument("--share-device", action="store_true",
                    help="(rehearsal, needs --dist-backend gloo) every rank uses device 0")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch: rendezvous port (0 = a free one)")
    ap.add_argument("--workload", choices=["auto", "C2", "C3", "C4", "C5"], default="auto")
    ap.add_argument("--engine-flags", type=int, default=0,
                    help="RPF_FLAG_* bits for the engine (A/B only, e.g. 8 = C4 on the two-kernel four-step path)")
    ap.add_argument("--shard-as", type=int, default=0,
                    help="(diagnostic) C5: process only the shard rank 0 of a job of this many ranks (2, 4, 8) would own "
                         "-- the per-rank step time of that job on one GPU's clock, an upper bound on its speed-up; "
                         "the line's value is then meaningless")
    ap.add_argument("--scans-per-reduce", type=int, default=4,
                    help="C5: scans whose spectra travel in ONE reduce (fewer, larger collectives: the 256 KB of one scan are "
                         "latency-bound; 1 = a reduce per scan)")
    ap.add_argument("--scans-per-launch", type=int, default=0,
                    help="C5: consecutive scans whose hops share ONE persistent kernel launch (a scan's hops already do); "
                         "0 = as many as divide --scans-per-reduce and fit rpf_max_hops_per_launch(); 1 = a launch per scan")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket the fused kernel with HIP events on every k-th timed step")
    return ap


def scans_per_launch(per_block, shards_max, max_hops, asked=0):
    """C5: how many CONSECUTIVE scans share one persistent launch.  The largest divisor of `per_block` (the scans that
    share a reduce block) whose accumulators -- `shards_max` per scan for the rank with the most shards -- fit the
    kernel's hop table (`max_hops` = rpf_max_hops_per_launch()); `asked` > 0 caps it (1 = a launch per scan)."""
    fit = [d for d in range(1, per_block + 1) if per_block % d == 0 and d * shards_max <= max_hops]
    if asked > 0:
        fit = [d for d in fit if d <= asked]
    return max(fit or [1])


def block_row(hop, sub, hops, k_scan):
    """Row of (hop, scan `sub`) in a reduce block of per_block scans x `hops` spectra: launch batch b = sub // k_scan
    first, then hop, then scan of the batch -- the spectra one launch writes (a rank's consecutive hops x the batch's
    scans, or one hop x fewer scans) are then CONSECUTIVE rows, which is what rpf_device_reduce writes."""
    return (sub // k_scan) * hops * k_scan + hop * k_scan + sub % k_scan


class LaunchError(Exception):
    """The requested ranks cannot be started as asked (exit code 2)."""


def check_rank_request(args, world, device_count):
    """Ranks against devices: one rank per device, or the explicit rehearsal."""
    if args.share_device and args.dist_backend != "gloo":
        raise LaunchError("--share-device is the one-GPU rehearsal and needs --dist-backend gloo "
                          "(RCCL cannot put two ranks on one device)")
    if world > 1 and not args.share_device and device_count < world:
        raise LaunchError("%d ranks asked for, %d HIP device(s) visible: refusing to measure fewer GPUs than "
                          "the line would claim (rehearsal on one GPU: --dist-backend gloo --share-device)"
                          % (world, device_count))
    if device_count < 1:
        raise LaunchError("no HIP device visible")


def plan_launch(args, argv, environ, device_count, port=None, python=None, script=None):
    """None: run in this process (one rank, or ranks already made by torchrun: WORLD_SIZE is set).
    Otherwise the command that starts --gpus ranks of this script with the same arguments."""
    if "WORLD_SIZE" in environ or args.gpus <= 1:
        return None
    check_rank_request(args, args.gpus, device_count)
    if port is None:
        port = args.master_port
    if not port:
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def self_launch(cmd, json_out):
    """Run the ranks; rank 0's JSON line is the only thing that reaches stdout."""
    import subprocess
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    lines = 0
    try:
        for line in proc.stdout:
            if line.startswith('{"metric"') and lines == 0:
                json_out.write(line if line.endswith("\n") else line + "\n")
                json_out.flush()
                lines += 1
            else:
                sys.stderr.write(line)
        rc = proc.wait()
    except BaseException:
        proc.kill()
        raise
    if rc == 0 and lines != 1:
        print("bench.py: the ranks exited 0 without printing a line", file=sys.stderr)
        rc = 1
    return rc


def main():
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (RCCL's version
    # banner, for one) goes to stderr instead
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    args = build_parser().parse_args()

    import torch
    import torch.distributed as dist

    try:
        cmd = plan_launch(args, sys.argv[1:], os.environ, torch.cuda.device_count())
    except LaunchError as exc:
        print("bench.py: %s" % exc, file=sys.stderr)
        sys.exit(2)
    if cmd is not None:
        sys.exit(self_launch(cmd, json_out))
    import rtl_power_fftw_amd as rpf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    try:
        check_rank_request(args, world, torch.cuda.device_count())
    except LaunchError as exc:
        if rank == 0:
            print("bench.py: %s" % exc, file=sys.stderr)
        sys.exit(2)
    use_dist = world > 1 or args.force_dist
    gloo = args.dist_backend == "gloo"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if gloo:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node; the container's hostname may not resolve
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d: the line reports the %d rank(s) that ran"
              % (args.gpus, world, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank if use_dist else 0)
    torch.cuda.set_device(dev)
    ctrl_dev = torch.device("cpu") if gloo else dev       # where the timing/flag all-reduces live

    name = args.workload
    selected = "explicit (--workload %s)" % name
    if name == "auto":
        # BASELINE.json: the metric is quoted on config 2 (one GPU); config 5, the sharded scan, is the multi-GPU one
        name = "C2" if world == 1 else "C5"
        selected = "auto: C2 (weak scaling) with one rank, C5 (strong scaling) with several -- this run: %s" % name
    wl = WORKLOADS[name]
    N, R = wl["N"], wl["R"]
    strong = name == "C5"
    hops = wl.get("hops", 1)
    if args.steps is None:                           # C4/C5 steps are 5-20x longer than C2's
        args.steps = {"C4": 100, "C5": 200}.get(name, 2000)
    if args.warmup is None:
        args.warmup = max(1, args.steps // 20)

    # ---- workload ------------------------------------------------------------------------
    # what this rank processes per step: list of (hop, first_frame, frames)
    if strong:
        mine = rpf.sharding.shard_hops(hops, R, args.shard_as or world, rank)
        # hop-major contiguous shards: a rank's hops are consecutive rows of the scan's block
        assert [m[0] for m in mine] == list(range(mine[0][0], mine[0][0] + len(mine)))
    else:
        mine = [(0, 0, R)]
    # the streams, generated on the device (bit-identical to synth.noise_tones_iq)
    base = []
    for hop, first, count in mine:
        seed = wl["seed"] + hop if strong else wl["seed"] + 1000 * rank
        base.append(rpf.synth.noise_tones_iq_torch(seed, N * count, dev, first=N * first))
    step_bytes = sum(int(b.numel()) for b in base)
    nb = args.replay_buffers or max(2, -(-(640 << 20) // max(1, step_bytes)))
    # further replay buffers = the same frames rotated by whole frames (distinct addresses and byte
    # order, same statistics, the same sum over frames up to rounding)
    bufs = [base] + [[torch.roll(b, shifts=2 * N * (37 * i)) for b in base] for i in range(1, nb)]

    window = rpf.synth.hann_window(N) if wl["window"] else None
    ds = rpf.Datastore(rpf.Params(N=N, window=window is not None, repeats=R), window, device=dev.index or 0,
                       flags=args.engine_flags)
    ds_max_hops = ds.max_hops_per_launch()
    # Exchange: one block = the `hops` spectra of a scan (C5) or of `hops`=8 consecutive
    # acquisitions (C2-C4), reduced onto rank 0 with ONE async RCCL reduce -- fewer, larger collectives --
    # on a ring of blocks so that it overlaps the following steps' kernels.
    # C5: `--scans-per-reduce` consecutive scans share a block (the reduce of 256 KB is latency-bound: measured 17 us
    # per scan at one per scan against a 33 us per-rank step at 8 ranks).
    per_block = max(1, args.scans_per_reduce) if strong else 8       # steps whose spectra share a block
    rows = hops * per_block if strong else per_block
    # C5: a rank's hops of k_scan CONSECUTIVE scans share one persistent launch (round 6; each (scan, hop) keeps its own
    # accumulator -- to the kernel they are m x k_scan hops, rpf_device_fused_hops): the ~10 us of per-launch fixed cost
    # are paid once per k_scan scans instead of once per scan, which is what an 8-rank job's 23 us of streaming per scan
    # needs.  k_scan is the same on every rank (the block layout below depends on it): the largest divisor of
    # per_block that fits the kernel's hop table for the rank with the most shards.
    k_scan = 1
    if strong:
        ranks_of_job = args.shard_as or world
        m_max = max(len(rpf.sharding.shard_hops(hops, R, ranks_of_job, r)) for r in range(ranks_of_job))
        k_scan = scans_per_launch(per_block, m_max, ds_max_hops, args.scans_per_launch)

    def row_of(hop, sub):
        return block_row(hop, sub, hops, k_scan)

    nring = 4
    # (strong scaling: rank 0 owns only some rows of a block and must clear the others before reuse;
    # weak scaling: every rank rewrites every row, nothing to clear)
    ring = rpf.sharding.ScanRing(rows, N, dev, nring=nring, dst=0, enabled=use_dist, clear_on_reuse=strong,
                                 host_staged=gloo)
    d_pwr = ring.blocks
    s = torch.cuda.current_stream().cuda_stream

    def launch_scans(i, n, hop_ids, blk, ev=None):
        """Scans i .. i+n-1 of this rank's shards `hop_ids` (indices into `mine`) in ONE persistent launch + ONE reduce
        (rpf_accumulate_device_hops' two halves, so that the events bracket the fused kernel alone)."""
        ptrs, nbytes, counts = [], [], []
        for h in hop_ids:
            for j in range(n):
                ptrs.append(bufs[(i + j) % nb][h].data_ptr())
                nbytes.append(2 * N * mine[h][2])
                counts.append(mine[h][2])
        if ev is not None:
            ev[0].record()
        ds.device_fused_hops(ptrs, nbytes, counts, s)
        if ev is not None:
            ev[1].record()
        ds.device_reduce(d_pwr[blk][row_of(mine[hop_ids[0]][0], i % per_block)].data_ptr(), s)

    def step(i, ev=None, limit=None):
        """Step i of a region of `limit` steps (None: open-ended).  C5: the launches cover k_scan steps at a time, so the
        steps in between have nothing left to do."""
        blk, sub = (i // per_block) % nring, i % per_block
        if strong:
            if sub % k_scan:
                return blk
            n = k_scan if limit is None else min(k_scan, limit - i)
            if sub == 0:
                ring.begin(blk)
            if n == k_scan or len(mine) == 1:
                launch_scans(i, n, list(range(len(mine))), blk, ev)
            else:                       # a region's ragged tail: hop by hop, so that each launch's rows are consecutive
                for h in range(len(mine)):
                    launch_scans(i, n, [h], blk, ev if h == 0 else None)
            if sub + n == per_block:
                ring.submit(blk)
            return blk
        if sub == 0:
            ring.begin(blk)
        streams = bufs[i % nb]
        for k, (hop, first, count) in enumerate(mine):
            out_row = d_pwr[blk][sub]
            if ev is not None and k == 0:
                ev[0].record()
            ds.device_fused(streams[k].data_ptr(), 2 * N * count, count, s)
            if ev is not None and k == 0:
                ev[1].record()
            ds.device_reduce(out_row.data_ptr(), s)
        if sub == per_block - 1:
            ring.submit(blk)
        return blk

    def fence(last_step=None):
        # a block of acquisitions cut short by the step count still owes its (partial) reduce
        if use_dist and last_step is not None and last_step % per_block != per_block - 1:
            blk = (last_step // per_block) % nring
            ring.submit(blk, async_op=False)
        ring.drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # this rank's own spectrum of its first shard in replay buffer 0 (checked against the CPU oracle below)
    d_chk = torch.zeros(N, dtype=torch.float64, device=dev)
    ds.accumulate_device(base[0].data_ptr(), 2 * N * mine[0][2], mine[0][2], d_chk.data_ptr(), s)
    torch.cuda.synchronize()
    pwr_first = d_chk.cpu().numpy().copy()

    # ---- untimed pre-warm, warm-up, timed regions ----------------------------------------------
    t0 = time.perf_counter()
    i = 0
    while True:
        step(i)
        i += 1
        if i % 4 == 0:
            torch.cuda.synchronize()
            flag = torch.tensor([1.0 if time.perf_counter() - t0 < PREWARM_SECONDS else 0.0], device=ctrl_dev)
            if use_dist:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)      # every rank leaves together
            if flag.item() == 0.0:
                break
    prewarm_steps = i
    fence(i - 1)
    for i in range(args.warmup):
        step(i, limit=args.warmup)
    fence(args.warmup - 1 if args.warmup else None)

    events = []
    regions = []
    my_regions = []
    last_blk = 0
    fused_before = ds.fused_status()
    while True:
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev = None
            # (C5: only the steps that launch -- every k_scan-th -- and only whole batches, so that the brackets are
            #  all launches of the same size)
            if args.event_every > 0 and i % args.event_every == 0 and len(events) < 4096 and (
                    not strong or (i % k_scan == 0 and i + k_scan <= args.steps)):
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events.append(ev)
            last_blk = step(i, ev, limit=args.steps)
        fence(args.steps - 1)
        elapsed = time.perf_counter() - t0
        my_regions.append(elapsed)
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctrl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        regions.append(elapsed)
        # repeat short regions (decided on the max-over-ranks time, so every rank agrees)
        target = min(MAX_REGIONS, max(1, int(math.ceil(MIN_REGION_SECONDS / max(regions[0], 1e-9)))))
        if len(regions) >= target:
            break
    elapsed = float(np.median(regions))
    # A fused four-step launch that gives up returns RPF_OK and a NaN spectrum (include/rpf_engine.h): a timed region that
    # contains one has timed work that produced nothing, and its line would be invalid.
    fused_after = ds.fused_status()
    if fused_after["gave_up"] > fused_before["gave_up"]:
        print("bench.py: rank %d: %d fused four-step launch(es) gave up inside the timed regions (NaN spectra): no line"
              % (rank, fused_after["gave_up"] - fused_before["gave_up"]), file=sys.stderr)
        sys.exit(3)

    # ---- C5: the reduced spectra of the last scan against the committed fixtures -----------------
    check = None
    if strong and rank == 0 and not args.shard_as:
        last_sub = (args.steps - 1) % per_block
        got = d_pwr[last_blk][[row_of(hop, last_sub) for hop in range(hops)]].cpu().numpy()
        worst = 0.0
        for hop in range(hops):
            g = np.load(os.path.join(ROOT, "tests", "golden", "c5_hop%d_n4096_r5000.npz" % hop))
            assert int(g["N"]) == N and int(g["repeats"]) == R and int(g["seed"]) == wl["seed"] + hop
            worst = max(worst, float(np.max(np.abs(got[hop] - g["pwr"]) / g["pwr"])))
        check = {"reduced_spectra_vs_float64_fixtures_max_rel": worst, "hops": hops,
                 "fixtures": "tests/golden/c5_hop*_n4096_r5000.npz"}
        assert worst < 1e-6, "reduced C5 spectra differ from the fixtures: %g" % worst

    # the C5 scan on ONE GPU (rank 0 alone, after the measurement, same launch form: as many consecutive scans per launch
    # as the hop table takes): the strong-scaling comparator.  In EVERY C5 line, N = 1 included, and -- as
    # `multi_gpu_reference` -- in the C2 line a `--gpus 1` run prints, so that a 1 / 2 / 4 / 8 sweep of `value` (C2 at
    # one rank, C5 at several) carries its own same-workload denominator.
    def one_gpu_scan_rate():
        c5 = WORKLOADS["C5"]
        n5, r5, h5 = c5["N"], c5["R"], c5["hops"]
        all_hops = [rpf.synth.noise_tones_iq_torch(c5["seed"] + h, n5 * r5, dev) for h in range(h5)]
        per = max(1, ds_max_hops // h5)
        d_one = torch.zeros(per * h5, n5, dtype=torch.float64, device=dev)
        ptrs = [a.data_ptr() for a in all_hops] * per

        def scans():
            ds.accumulate_device_hops(ptrs, [2 * n5 * r5] * (per * h5), [r5] * (per * h5), d_one.data_ptr(), s)
        for _ in range(10):
            scans()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nlaunch = 50
        for _ in range(nlaunch):
            scans()
        torch.cuda.synchronize()
        return h5 * r5 * n5 * per * nlaunch / (time.perf_counter() - t0), per

    one_gpu = None
    if (strong and not args.shard_as) or (name == "C2" and world == 1 and selected.startswith("auto")):
        if rank == 0:
            one_gpu = one_gpu_scan_rate()
        if use_dist:
            dist.barrier()

    # every rank's own view of the timed region (so that a multi-GPU run explains itself): its wall time
    # for the K steps, its fused-kernel bracket, what it processed
    k1_all = [a.elapsed_time(b) for a, b in events]
    mine_report = {"rank": rank, "device": torch.cuda.get_device_name(dev), "device_index": dev.index or 0, "region_s_median": float(np.median(my_regions)),
                   "kernel_ms_median": float(np.median(k1_all)) if k1_all else None,
                   "frames_per_step": sum(m[2] for m in mine) if strong else mine[0][2],
                   "hops_per_step": len(mine) if strong else 1,
                   "reduce_wait_s_per_step": ring.wait_seconds / max(1, ring.waits) if use_dist else None}
    per_rank = [mine_report]
    if use_dist and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine_report)
        per_rank = gathered

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        samples_per_step = hops * R * N if strong else world * N * R
        value = samples_per_step * args.steps / elapsed
        # median of the in-region brackets (a stray preemption in one bracket must not move the figure); mean beside it
        k1_ms = float(np.median(k1_all)) if events else None
        k1_mean_ms = float(np.mean(k1_all)) if events else None
        # one launch of the dominant kernel: one acquisition (C2-C4), the rank's hops of the scan (C5)
        frames_per_launch = k_scan * sum(m[2] for m in mine) if strong else mine[0][2]
        hops_per_launch = k_scan * len(mine) if strong else 1
        alg_bytes = 2 * N * frames_per_launch + hops_per_launch * 8 * N + (4 * N if window is not None else 0)
        info = ds.launch_info()
        roof = None
        if k1_ms:
            achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
            traffic = measured_peak = traffic_note = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    key = {"C2": "fft_accum_c2", "C3": "fft_accum_c3", "C4": "fourstep_c4", "C5": "fft_accum_c5"}[name]
                    traffic = tj.get(key + "_hbm_bytes_per_launch")
                    if traffic and strong:      # captured per launch of N frames (rounds <= 5: one 40000-frame scan): scaled
                        traffic = traffic * frames_per_launch / float(tj.get("fft_accum_c5_frames_per_launch", hops * R))
                    traffic_note = "%s; captured %s" % (tj.get("source"), tj.get("captured", "round 1 (date not recorded)"))
                    measured_peak = tj.get("measured_read_only_GBps")   # tools/hbm_read_bench.hip, same box type
                except Exception:
                    traffic = None
            # (which four-step kernel ran is what the ENGINE says -- rpf_fused_status --, not what the flags asked for:
            #  an engine whose teams did not assemble at creation is on the two-kernel path)
            fused = ds.fused_status()
            kernel = ("fft_accum_scan_kernel<N=4096,P=16> (K1: %d shard(s) x %d consecutive scan(s) = %d accumulators in one "
                      "persistent launch)" % (len(mine), k_scan, hops_per_launch) if strong else
                      "fft_accum_kernel<N=4096,P=16> (K1)" if N == 4096 else
                      "fourstep_fused_kernel<Split<512,512>> (one persistent launch per acquisition, Y handed over in the XCDs' L2)"
                      if fused["active"] else
                      "fourstep transform of one acquisition (two-kernel path: all batches of the column/row kernels)")
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": ("NOT measured in this run: rocprofv3 PMC capture replayed from profiles/traffic.json -- "
                                       + str(traffic_note)) if traffic else None,
                    "kernel": kernel, "kernel_ms": k1_ms, "kernel_ms_mean": k1_mean_ms, "kernel_ms_samples": len(events),
                    "algorithmic_bytes_per_launch": alg_bytes, "frames_per_launch": frames_per_launch,
                    "hops_per_launch": hops_per_launch,
                    "fused_four_step": fused if N != 4096 else None,
                    "samples_per_s_kernel_only": N * frames_per_launch / (k1_ms * 1e-3),
                    # the contracted roofline is HBM read; what actually limits the kernel:
                    "limited_by": "fp32-valu + lds (see secondary): ~55 flop/B against a machine balance of ~20",
                    "secondary": secondary_limits(N, frames_per_launch, k1_ms * 1e-3)}
            if measured_peak:
                roof["measured_read_only_peak"] = measured_peak
                roof["frac_of_measured_read_only"] = achieved / measured_peak
        desc = {
            "C2": "C2: N=4096 bins x 10000 repeats per step per GPU, rectangular window",
            "C3": "C3: N=4096 bins x 10000 repeats per step per GPU, periodic Hann window",
            "C4": "C4: N=262144 bins x 1000 repeats per step per GPU, rectangular window",
            "C5": "C5: 8-hop scan per step, N=4096 bins x 5000 repeats per hop (seed 50+hop), "
                  "hop-major frame-aligned shards over %d GPU(s)" % world,
        }[name]
        out = {
            "metric": "IQ samples/s through FFT+|X|^2-accumulate",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc + ", u8 IQ resident in HBM (%d replay buffers of %d B per GPU)" % (nb, step_bytes),
                       "workload_name": name, "workload_selected": selected,
                       "launch": info,
                       "shards_of_rank0": [list(m) for m in mine] if strong else None,
                       "scans_per_launch": k_scan if strong else None,
                       "reduce": ("one async %s reduce of %d x %d f64 bins per %s" % (
                           "gloo (staged through pinned host memory)" if gloo else "RCCL",
                           rows, N, ("%d scans" % per_block if per_block > 1 else "scan") if strong else "8 steps"))
                       if use_dist else "none"},
            "timing": {"prewarm_steps": prewarm_steps, "regions_s": regions, "reported": "median region",
                       "min_region_s": min(regions), "max_region_s": max(regions)},
            "roofline": roof,
        }
        if use_dist:
            out["per_rank"] = per_rank
            # what the collective library saw (so that a reader of the line can tell an N-rank RCCL run from a rehearsal)
            if gloo:
                out["rccl"] = {"world_size": 0, "version": None, "backend": "gloo (host-staged rehearsal): RCCL not used"}
            else:
                try:
                    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:
                    ver = None
                out["rccl"] = {"world_size": dist.get_world_size(), "version": ver, "backend": dist.get_backend(),
                               "devices": sorted(set(p["device_index"] for p in per_rank))}
        if args.share_device:
            out["rehearsal"] = ("%d ranks share device 0 and time-slice it; the exchange is a gloo reduce staged through "
                                "the host: launcher, shards, ring reuse, per-rank reports and the fixture check are what "
                                "this run exercises -- `value` says nothing about scaling" % world)
        if args.shard_as:
            out["shard_as"] = {"ranks": args.shard_as, "what": "only the shard rank 0 of a job of that many ranks would own; "
                               "ms_per_step = that rank's step on this GPU's clock; `value` is not a throughput"}
        if check:
            out["check"] = check
        if one_gpu:
            ref = {"value": one_gpu[0], "unit": "samples/s", "scans_per_launch": one_gpu[1],
                   "what": "config C5's 8-hop scan on rank 0's GPU alone, no reduce"}
            if strong:
                out["one_gpu_same_workload"] = ref
            else:       # the C2 line of a one-rank run: what a --gpus N > 1 run of this script (C5) is to be divided by
                out["multi_gpu_reference"] = dict(ref, workload_name="C5")
        if world == 1 and not args.force_dist:
            # rank 0's first shard on the host for the CPU legs
            host = base[0].cpu().numpy()
            fr = mine[0][2]
            if not args.no_cpu_baseline:       # the CPU leg is timed at N=1 only
                out["cpu_baseline"] = cpu_baseline(N, fr, host, pwr_first, window)
            if not args.no_end_to_end and N <= 8192:
                out["end_to_end"] = end_to_end(rpf, N, fr, host, window, dev.index or 0)
        print(json.dumps(out), file=json_out, flush=True)

    ds.close()
    ring.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
