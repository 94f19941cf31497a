"""
ctypes binding of libprosstt_amd.so (include/prosstt_amd.h).

There is NO CPU fallback: if the library is missing, or no gfx950 device is
visible, every numeric entry point of the package raises.  torch is used only
as plumbing (device memory, the current stream, torch.distributed).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PROSSTT_AMD_LIB") or os.path.join(_HERE, "lib", "libprosstt_amd.so")

OK, EINVAL, EDOMAIN, EHIP, ENOMEM, ENODEV, ERCCL = 0, -1, -2, -3, -4, -5, -6
HOST_INPUTS, HOST_OUTPUT, CHECK_DOMAIN, TIME_KERNEL, CHECK_DEFERRED, MEANS_CACHED, PARAMS_NONNEG = 1, 2, 4, 8, 16, 32, 64

# every symbol include/prosstt_amd.h declares
SYMBOLS = [
    "prosstt_amd_version", "prosstt_amd_last_error", "prosstt_amd_device_count",
    "prosstt_amd_ctx_create", "prosstt_amd_ctx_destroy", "prosstt_amd_ctx_synchronize",
    "prosstt_amd_last_kernel_ms", "prosstt_amd_sample_counts", "prosstt_amd_plan_order", "prosstt_amd_last_list", "prosstt_amd_nb_params",
    "prosstt_amd_hw_math", "prosstt_amd_hw_math_at", "prosstt_amd_comm_unique_id", "prosstt_amd_comm_init", "prosstt_amd_comm_destroy",
    "prosstt_amd_gather_counts", "prosstt_amd_comm_selftest", "prosstt_amd_domain_status", "prosstt_amd_numpy_programs",
    "prosstt_amd_lineage_attempt", "prosstt_amd_lineage_attempt_batch", "prosstt_amd_lineage_walk",
    "prosstt_amd_lineage_walk_batch",
    "prosstt_amd_lineage_commit",
    "prosstt_amd_gene_max",
    "prosstt_amd_means_from_rel",
]


HOST_LIB_PATH = os.environ.get("PROSSTT_AMD_HOST_LIB") or os.path.join(_HERE, "lib", "libprosstt_amd_host.so")
# every symbol include/prosstt_amd_host.h declares
HOST_SYMBOLS = ["prosstt_amd_host_widen_i32_i64", "prosstt_amd_host_widen_u16_i64", "prosstt_amd_host_widen_u16_i32",
                "prosstt_amd_host_widen_u8_i64", "prosstt_amd_host_widen_u8_i32", "prosstt_amd_host_scatter_i32",
                "prosstt_amd_host_has_avx2"]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("prosstt_amd error %d: %s" % (code, message))
        self.code = code


_lib = None
_lock = threading.Lock()


_host_lib = None


def load_host():
    """libprosstt_amd_host.so (include/prosstt_amd_host.h: host-side helpers, no HIP), once.  Raises if it has not been built."""
    global _host_lib
    with _lock:
        if _host_lib is not None:
            return _host_lib
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("%s not found: build it with `make -C prosstt_amd/csrc/host` (or "
                               "`python -c 'import __graft_entry__ as g; g.build()'`)" % HOST_LIB_PATH)
        L = ctypes.CDLL(HOST_LIB_PATH)
        L.prosstt_amd_host_scatter_i32.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32]
        L.prosstt_amd_host_scatter_i32.restype = ctypes.c_int
        for name in ("prosstt_amd_host_widen_i32_i64", "prosstt_amd_host_widen_u16_i64", "prosstt_amd_host_widen_u16_i32",
                     "prosstt_amd_host_widen_u8_i64", "prosstt_amd_host_widen_u8_i32"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32]
            getattr(L, name).restype = ctypes.c_int
        L.prosstt_amd_host_has_avx2.restype = ctypes.c_int
        _host_lib = L
        return L


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s not found: build it with `make -C prosstt_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). prosstt_amd has no "
                "CPU fallback." % LIB_PATH)
        # torch first: it bundles its own libamdhip64.so.7, and the dynamic loader shares one
        # copy per SONAME.  Loading ours first would bind torch to /opt/rocm's runtime instead;
        # either way both must sit on ONE HIP runtime for streams and device pointers to be
        # exchangeable, and torch's own is the combination the wheel was built against.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, u32, u64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                  ctypes.c_uint32, ctypes.c_uint64)
        L.prosstt_amd_version.restype = ctypes.c_int
        L.prosstt_amd_last_error.restype = ctypes.c_char_p
        L.prosstt_amd_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
        L.prosstt_amd_ctx_create.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
        L.prosstt_amd_ctx_destroy.argtypes = [vp]
        L.prosstt_amd_ctx_synchronize.argtypes = [vp]
        L.prosstt_amd_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.prosstt_amd_sample_counts.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, i64, u64, u64,
                                                vp, vp, i64, u32]
        L.prosstt_amd_plan_order.argtypes = [vp, i64, i64, vp]
        L.prosstt_amd_last_list.argtypes = [vp, vp, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i32)]
        L.prosstt_amd_nb_params.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, u32]
        L.prosstt_amd_hw_math.argtypes = [vp, i32, u32, u64, vp, u32]
        L.prosstt_amd_hw_math_at.argtypes = [vp, i32, vp, u64, vp, u32]
        L.prosstt_amd_comm_unique_id.argtypes = [vp]
        L.prosstt_amd_comm_init.argtypes = [vp, vp, i32, i32, ctypes.POINTER(vp)]
        L.prosstt_amd_comm_destroy.argtypes = [vp]
        L.prosstt_amd_gather_counts.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.prosstt_amd_comm_selftest.argtypes = [vp, vp, u64]
        L.prosstt_amd_domain_status.argtypes = [vp, ctypes.POINTER(i32)]
        L.prosstt_amd_numpy_programs.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.prosstt_amd_lineage_attempt.argtypes = [vp, vp, i32, i32, vp, i64, i32, vp, vp, vp, vp]
        L.prosstt_amd_lineage_attempt_batch.argtypes = [vp, vp, i32, i32, i32, vp, i64, i32, vp, vp, vp, vp]
        L.prosstt_amd_lineage_walk.argtypes = [vp, u64, u64, i32, i32, vp]
        L.prosstt_amd_lineage_walk_batch.argtypes = [vp, u64, u64, i32, i32, i32, vp]
        L.prosstt_amd_lineage_commit.argtypes = [vp, vp, i32, i32, vp, i64, vp, vp]
        L.prosstt_amd_gene_max.argtypes = [vp, vp, i64, i64, vp]
        L.prosstt_amd_means_from_rel.argtypes = [vp, vp, vp, i64, i64, vp]
        for name in SYMBOLS:
            if name not in ("prosstt_amd_last_error", "prosstt_amd_version"):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
        return L


def check(code):
    if code != OK:
        msg = load().prosstt_amd_last_error().decode("utf-8", "replace")
        if code == EDOMAIN:
            raise ValueError(msg)          # what scipy raises in the reference (simulation.py:647)
        raise NativeError(code, msg)


def device_count():
    n = ctypes.c_int(0)
    rc = load().prosstt_amd_device_count(ctypes.byref(n))
    return n.value if rc == OK else 0
