"""PCIe-inclusive throughput of the buffer-queue path (pinned host buffers ->
hipMemcpyAsync -> K1), for DESIGN.md.  Never the benchmark's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtl_power_fftw_amd as rpf

N = 4096
for buf_length, buffers, R in ((1638400, 5, 200000), (16 * 1638400, 5, 400000), (64 * 1638400, 4, 800000)):
    ds = rpf.Datastore(rpf.Params(N=N, buf_length=buf_length, buffers=buffers, repeats=R))
    fill = rpf.synth.noise_tones_iq(1, buf_length // 2)
    for rep in range(2):
        ds.begin(R)
        need = 2 * N * R
        sent = 0
        t0 = time.perf_counter()
        while sent < need:
            b = ds.acquire()
            n = min(buf_length, need - sent)
            if rep == 0 and sent < buffers * buf_length:
                b[:n] = fill[:n]            # touch every pinned buffer once; later passes replay them
            ds.submit(b, n)
            sent += n
        done = ds.finish()
        dt = time.perf_counter() - t0
    print("buf_length=%9d buffers=%d repeats=%d done=%d  %.3f s  %.1f Gsample/s  %.1f GB/s over PCIe  hist=%s"
          % (buf_length, buffers, R, done, dt, N * done / dt / 1e9, 2 * N * done / dt / 1e9, ds.queue_histogram), flush=True)
    ds.close()
