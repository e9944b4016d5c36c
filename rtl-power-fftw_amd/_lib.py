"""ctypes binding of include/rpf_engine.h (the C-ABI drop-in boundary)."""
import ctypes
import enum
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ReturnValue(enum.IntEnum):
    """Process exit codes of the reference (/root/reference/src/exceptions.h:25-34)."""
    Success = 0
    NoDeviceFound = 1
    InvalidDeviceIndex = 2
    InvalidArgument = 3
    TCLAPerror = 4
    InvalidInput = 5
    AcquisitionError = 6
    HardwareError = 7


class RPFError(RuntimeError):
    """Mirror of ``RPFexception(what, ReturnValue)`` (exceptions.h:39-47)."""

    def __init__(self, what, retval):
        super().__init__(what)
        self.retval = ReturnValue(retval)

    def returnValue(self):
        return self.retval


class rpf_config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("N", ctypes.c_int32),
        ("window", ctypes.POINTER(ctypes.c_float)),
        ("n_buffers", ctypes.c_int32),
        ("buffer_capacity", ctypes.c_int64),
        ("device", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
    ]


FLAG_NO_LDS_DMA = 1
FLAG_FOURSTEP_FUSED = 2
FLAG_NO_MIXED_RADIX = 4
FLAG_NO_FOURSTEP_FUSED = 8

# every symbol include/rpf_engine.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
_SYMBOLS = [
    ("rpf_abi_version", ctypes.c_int, []),
    ("rpf_supported_n", ctypes.c_int, [ctypes.c_int]),
    ("rpf_last_global_error", ctypes.c_char_p, []),
    ("rpf_engine_create", ctypes.c_int, [ctypes.POINTER(rpf_config), ctypes.POINTER(_P)]),
    ("rpf_engine_destroy", None, [_P]),
    ("rpf_last_error", ctypes.c_char_p, [_P]),
    ("rpf_begin", ctypes.c_int, [_P, ctypes.c_int64]),
    ("rpf_buffer_acquire", ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t)]),
    ("rpf_buffer_submit", ctypes.c_int, [_P, _P, ctypes.c_size_t]),
    ("rpf_buffer_unget", ctypes.c_int, [_P, _P]),
    ("rpf_finish", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_get_power", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double)]),
    ("rpf_get_repeats_done", ctypes.c_int64, [_P]),
    ("rpf_get_histogram", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    ("rpf_accumulate", ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int64,
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_accumulate_device", ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int64, _P, _P,
                                             ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_device_fused", ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int64, _P,
                                        ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_device_reduce", ctypes.c_int, [_P, _P, _P]),
    ("rpf_accumulate_device_hops", ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t),
                                                  ctypes.POINTER(ctypes.c_int64), ctypes.c_int, _P, _P,
                                                  ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_device_fused_hops", ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t),
                                             ctypes.POINTER(ctypes.c_int64), ctypes.c_int, _P,
                                             ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_max_hops_per_launch", ctypes.c_int, []),
    ("rpf_copy_power_device", ctypes.c_int, [_P, _P, _P, ctypes.c_int]),
    ("rpf_scan_reducer_create", ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(_P)]),
    ("rpf_scan_reducer_destroy", None, [_P]),
    ("rpf_scan_reducer_last_error", ctypes.c_char_p, [_P]),
    ("rpf_scan_reducer_begin", ctypes.c_int, [_P]),
    ("rpf_scan_reducer_deposit", ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P]),
    ("rpf_scan_reducer_reduce", ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    ("rpf_stream_register", ctypes.c_int, [_P, _P, ctypes.c_size_t]),
    ("rpf_stream_unregister", ctypes.c_int, [_P, _P]),
    ("rpf_fused_status", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int64)]),
    ("rpf_last_launch_info", ctypes.c_int, [_P] + [ctypes.POINTER(ctypes.c_int)] * 4),
]


# test hooks (csrc/rpf_engine_testing.h): exported by the library, NOT part of include/rpf_engine.h
_TEST_HOOKS = [
    ("rpf_debug_fused_fault", ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
]


def symbol_names():
    return [s[0] for s in _SYMBOLS]


def test_hook_names():
    return [s[0] for s in _TEST_HOOKS]


def lib_path():
    # RPF_ENGINE_LIB: alternative build of the same library (A/B measurements)
    return os.environ.get("RPF_ENGINE_LIB") or os.path.join(_HERE, "librpf_engine.so")


def build(force=False):
    """Compile the gfx950 engine library in-tree with hipcc (no GPU needed)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src_dir]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    return lib_path()


def load():
    """Load librpf_engine.so.  There is no fallback: a missing library is an error.

    torch (if installed) is imported first so that this library binds to the
    HIP runtime already in the process (both export SONAME libamdhip64.so.7);
    loading in the other order would put two HIP runtimes in one process."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RPFError("%s not found: build it with __graft_entry__.build() "
                       "(there is no CPU fallback)" % path, ReturnValue.HardwareError)
    if not os.environ.get("RPF_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(path)
    for name, restype, argtypes in _SYMBOLS + _TEST_HOOKS:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rpf_abi_version() != 2:
        raise RPFError("librpf_engine.so ABI mismatch", ReturnValue.HardwareError)
    _LIB = lib
    return lib
