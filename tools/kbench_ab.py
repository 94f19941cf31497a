"""A/B timing of the count sampler's dominant kernel over several builds of the library IN ONE PROCESS,
rounds interleaved (variant 1, variant 2, ..., variant 1, ...): same device, same clock state, same inputs.
    python tools/kbench_ab.py [C3] [rounds] lib_a.so lib_b.so ...      ("shipped" = the in-tree library)
Prints per variant the median / min of the stream kernel's HIP-event time and of the whole call, and the
sum of the counts (timing-only variants of tools/ablate.py give other sums: their outputs are wrong by
construction)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prosstt_amd import _native, device, workloads

args = sys.argv[1:]
cfg = args.pop(0) if args and args[0] in workloads.CONFIGS else "C3"
rounds = int(args.pop(0)) if args and args[0].isdigit() else 12
libs = args or ["shipped"]
BURST = int(os.environ.get("KBENCH_BURST", "1"))
ctx0 = device.get_context()
w = workloads.build(cfg)
pt, br, sc, rows = w.plan(int(os.environ["KBENCH_CELLS"]) if "KBENCH_CELLS" in os.environ else (125000 if cfg == "C5" else None))
if os.environ.get("KBENCH_SORT") == "1":      # diagnosis: cells in the order of their rows of the mean tensor
    o = np.argsort(rows, kind="stable"); rows, sc = np.asarray(rows)[o], np.asarray(sc)[o]
G = w.tree.G
dm = w.tree.device_means(); dr = ctx0.tensor(rows, torch.int32); ds = ctx0.tensor(sc, torch.float64)
da = ctx0.tensor(w.alpha, torch.float64); db = ctx0.tensor(w.beta, torch.float64)
out = torch.empty((len(rows), G), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
vp, i32, i64, u32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64


class Variant:
    def __init__(self, path):
        self.name = os.path.basename(path)
        self.lib = ctypes.CDLL(_native.LIB_PATH if path == "shipped" else os.path.abspath(path))
        self.lib.prosstt_amd_ctx_create.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
        self.lib.prosstt_amd_sample_counts.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, i64, u64, u64, vp, vp, i64, u32]
        self.lib.prosstt_amd_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        self.lib.prosstt_amd_last_error.restype = ctypes.c_char_p
        self.h = vp()
        assert self.lib.prosstt_amd_ctx_create(torch.cuda.current_device(), vp(stream), ctypes.byref(self.h)) == 0
        self.kernel, self.call, self.sum = [], [], None

    def run(self, seed):
        # KBENCH_BURST=n: n calls back to back per round (the device keeps its clock), the call time is their mean
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(BURST):
            rc = self.lib.prosstt_amd_sample_counts(self.h, vp(dm.data_ptr()), dm.shape[0], G, vp(dr.data_ptr()), vp(ds.data_ptr()),
                                                    vp(da.data_ptr()), vp(db.data_ptr()), len(rows), seed, 0, None,
                                                    vp(out.data_ptr()), out.stride(0), _native.TIME_KERNEL)
            assert rc == 0, self.lib.prosstt_amd_last_error()
        e1.record(); torch.cuda.synchronize()
        ms = ctypes.c_float(0)
        self.lib.prosstt_amd_last_kernel_ms(self.h, ctypes.byref(ms))
        self.kernel.append(ms.value); self.call.append(e0.elapsed_time(e1) / BURST)


vs = [Variant(p) for p in libs]
for v in vs:                       # warm-up: workspace growth, code upload
    v.run(0); v.run(1)
    v.sum = int(out.sum()); v.kernel.clear(); v.call.clear()
for r in range(rounds):
    for v in (vs if r % 2 == 0 else vs[::-1]):
        v.run(2)
n = len(rows) * G
base = float(np.median(vs[0].kernel))
for v in vs:
    k = float(np.median(v.kernel))
    print("%-44s %s kernel median %.3f min %.3f ms (%+5.1f %% vs first) | call median %.3f (%+5.2f %%) | %.1f %% of 8 TB/s | sum %d" % (
        v.name[-44:], cfg, k, min(v.kernel), (k / base - 1) * 100, float(np.median(v.call)),
        (float(np.median(v.call)) / float(np.median(vs[0].call)) - 1) * 100, n * 4.0325 / (k * 1e-3) / 8e12 * 100, v.sum))
