"""Host-side mirror of ``class Datastore`` (/root/reference/src/datastore.h:35-68)
over the C-ABI of include/rpf_engine.h.  Used by tests/ and bench.py; the C++
host wrapper with the same shape lives in rtl-power-fftw_amd/host/."""
import ctypes
import sys

import numpy as np

from . import _lib
from ._lib import RPFError, ReturnValue

BASE_BUF = 16384                # params.h:26
DEFAULT_BUF_MULTIPLIER = 100    # params.h:27


class Params:
    """The Params fields the hot path reads, with the reference's defaults
    (/root/reference/src/params.h:33-66)."""

    def __init__(self, N=512, buffers=5, buf_length=BASE_BUF * DEFAULT_BUF_MULTIPLIER,
                 repeats=None, window=False, sample_rate=2000000, cfreq=1420405752,
                 linear=False, baseline=False):
        self.N = N
        self.buffers = buffers
        self.buf_length = buf_length
        # params.h:56: repeats = buf_length/(2*N) unless -n/-t say otherwise
        self.repeats = buf_length // (2 * N) if repeats is None else repeats
        self.window = window
        self.sample_rate = sample_rate
        self.cfreq = cfreq
        self.linear = linear
        self.baseline = baseline


class Datastore:
    """``Datastore(params, window_values)``: buffer pool + FFT/accumulate worker."""

    def __init__(self, params, window_values=None, device=0, flags=0):
        self.params = params
        self._lib = _lib.load()
        self._handle = ctypes.c_void_p()
        self._window = None
        cfg = _lib.rpf_config()
        cfg.struct_size = ctypes.sizeof(_lib.rpf_config)
        cfg.N = params.N
        if params.window:
            if window_values is None or len(window_values) != params.N:
                raise RPFError("Error reading window function. Expected %d values, found %d."
                               % (params.N, 0 if window_values is None else len(window_values)),
                               ReturnValue.InvalidInput)
            self._window = np.ascontiguousarray(window_values, dtype=np.float32)
            cfg.window = self._window.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        cfg.n_buffers = params.buffers
        cfg.buffer_capacity = params.buf_length
        cfg.device = device
        cfg.flags = flags
        rc = self._lib.rpf_engine_create(ctypes.byref(cfg), ctypes.byref(self._handle))
        if rc != 0:
            self._handle = ctypes.c_void_p()
            raise RPFError(self._lib.rpf_last_global_error().decode(), rc)
        self.pwr = np.zeros(params.N, dtype=np.float64)
        self.repeats_done = 0

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None):
            self._lib.rpf_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RPFError(self._lib.rpf_last_error(self._handle).decode(), rc)

    # -- the hand-off protocol of Acquisition::run ------------------------
    def begin(self, repeats=None):
        """acquisition.cxx:252-256"""
        self._check(self._lib.rpf_begin(self._handle,
                                        self.params.repeats if repeats is None else repeats))

    def acquire(self):
        """acquisition.cxx:278-285 -> writable uint8 view of a pinned buffer"""
        ptr = ctypes.c_void_p()
        cap = ctypes.c_size_t()
        self._check(self._lib.rpf_buffer_acquire(self._handle, ctypes.byref(ptr), ctypes.byref(cap)))
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)),
                                    shape=(cap.value,))
        return arr

    def submit(self, buf, nbytes):
        """acquisition.cxx:302,320-323"""
        self._check(self._lib.rpf_buffer_submit(self._handle, ctypes.c_void_p(buf.ctypes.data), nbytes))

    def unget(self, buf):
        """acquisition.cxx:310-314"""
        self._check(self._lib.rpf_buffer_unget(self._handle, ctypes.c_void_p(buf.ctypes.data)))

    def finish(self):
        """acquisition.cxx:343-347; afterwards pwr / repeats_done are valid"""
        done = ctypes.c_int64()
        self._check(self._lib.rpf_finish(self._handle, ctypes.byref(done)))
        self.repeats_done = done.value
        self._check(self._lib.rpf_get_power(self._handle,
                                            self.pwr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return self.repeats_done

    @property
    def queue_histogram(self):
        out = (ctypes.c_int * (self.params.buffers + 1))()
        self._check(self._lib.rpf_get_histogram(self._handle, out))
        return list(out)

    def printQueueHistogram(self, file=sys.stderr):
        """datastore.cxx:98-103"""
        file.write("Buffer queue histogram: " + "".join("%d " % v for v in self.queue_histogram) + "\n")

    # -- whole-stream conveniences -----------------------------------------
    def accumulate(self, stream, repeats=None):
        """Run one acquisition over a contiguous host byte stream through the
        buffer queues.  Returns (pwr copy, repeats_done)."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        done = ctypes.c_int64()
        self._check(self._lib.rpf_accumulate(
            self._handle, ctypes.c_void_p(stream.ctypes.data), stream.size,
            self.params.repeats if repeats is None else repeats,
            self.pwr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(done)))
        self.repeats_done = done.value
        return self.pwr.copy(), self.repeats_done

    def register_stream(self, stream):
        """rpf_stream_register: pin a host array that will be replayed more than once; accumulate() on it (or on a slice
        of it) then skips the copy into the pool.  Keep the array alive until unregister_stream / close."""
        assert stream.dtype == np.uint8 and stream.flags["C_CONTIGUOUS"]
        self._check(self._lib.rpf_stream_register(self._handle, ctypes.c_void_p(stream.ctypes.data), stream.size))

    def unregister_stream(self, stream):
        self._check(self._lib.rpf_stream_unregister(self._handle, ctypes.c_void_p(stream.ctypes.data)))

    def accumulate_device(self, d_stream_ptr, nbytes, repeats, d_pwr_ptr, hip_stream=0):
        """Enqueue the fused kernel over a stream resident in HBM (raw device
        pointers; asynchronous).  Returns the number of frames that will be summed."""
        done = ctypes.c_int64()
        self._check(self._lib.rpf_accumulate_device(
            self._handle, ctypes.c_void_p(d_stream_ptr), nbytes, repeats,
            ctypes.c_void_p(d_pwr_ptr), ctypes.c_void_p(hip_stream), ctypes.byref(done)))
        return done.value

    def device_fused(self, d_stream_ptr, nbytes, repeats, hip_stream=0):
        """K1 only (measurement hook, rpf_device_fused)."""
        done = ctypes.c_int64()
        self._check(self._lib.rpf_device_fused(self._handle, ctypes.c_void_p(d_stream_ptr), nbytes,
                                               repeats, ctypes.c_void_p(hip_stream), ctypes.byref(done)))
        return done.value

    def device_reduce(self, d_pwr_ptr, hip_stream=0):
        """K3 only (measurement hook, rpf_device_reduce)."""
        self._check(self._lib.rpf_device_reduce(self._handle, ctypes.c_void_p(d_pwr_ptr),
                                                ctypes.c_void_p(hip_stream)))

    @staticmethod
    def _hop_arrays(d_stream_ptrs, nbytes, repeats):
        H = len(d_stream_ptrs)
        assert len(nbytes) == H and len(repeats) == H
        return (H, (ctypes.c_void_p * H)(*[int(p) for p in d_stream_ptrs]),
                (ctypes.c_size_t * H)(*[int(b) for b in nbytes]),
                (ctypes.c_int64 * H)(*[int(r) for r in repeats]), (ctypes.c_int64 * H)())

    def accumulate_device_hops(self, d_stream_ptrs, nbytes, repeats, d_pwr_ptr, hip_stream=0):
        """A whole scan of device-resident hops in one call (rpf_accumulate_device_hops): hop h's
        spectrum lands in d_pwr[h*N : (h+1)*N].  Returns the frames summed per hop."""
        H, ptrs, nb, rep, done = self._hop_arrays(d_stream_ptrs, nbytes, repeats)
        self._check(self._lib.rpf_accumulate_device_hops(self._handle, ptrs, nb, rep, H, ctypes.c_void_p(d_pwr_ptr),
                                                         ctypes.c_void_p(hip_stream), done))
        return list(done)

    def device_fused_hops(self, d_stream_ptrs, nbytes, repeats, hip_stream=0):
        """K1 over up to max_hops_per_launch() hops in ONE launch (measurement hook); device_reduce()
        then writes all their spectra."""
        H, ptrs, nb, rep, done = self._hop_arrays(d_stream_ptrs, nbytes, repeats)
        self._check(self._lib.rpf_device_fused_hops(self._handle, ptrs, nb, rep, H, ctypes.c_void_p(hip_stream), done))
        return list(done)

    def max_hops_per_launch(self):
        return self._lib.rpf_max_hops_per_launch()

    def fused_status(self):
        """rpf_fused_status: is the fused four-step kernel what the next launch runs, how many of its launches gave
        up, how many of those the queue worker ran again on the two-kernel path."""
        active = ctypes.c_int()
        gave_up, recovered = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.rpf_fused_status(self._handle, ctypes.byref(active), ctypes.byref(gave_up),
                                               ctypes.byref(recovered)))
        return {"active": bool(active.value), "gave_up": gave_up.value, "recovered": recovered.value}

    def debug_fused_fault(self, mode, skip=0, count=-1):
        """rpf_debug_fused_fault (test hook)."""
        self._check(self._lib.rpf_debug_fused_fault(self._handle, mode, skip, count))

    def launch_info(self):
        vals = [ctypes.c_int() for _ in range(4)]
        self._check(self._lib.rpf_last_launch_info(self._handle, *[ctypes.byref(v) for v in vals]))
        return dict(zip(("grid", "block", "frames_per_wg", "lds_bytes"), (v.value for v in vals)))
