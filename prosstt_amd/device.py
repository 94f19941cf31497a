"""
Thin object layer over the C ABI: one ``Context`` per (device, stream).

torch supplies device memory and the stream handle; every numeric step is a
call into libprosstt_amd.so.  Nothing here computes on the CPU.
"""
import ctypes
import os
import sys
import threading

import numpy as np

from . import _native


def _torch():
    import torch
    return torch


def require_gpu():
    """Raise unless the HIP library loads and a gfx950 device is visible."""
    _native.load()
    torch = _torch()
    if not torch.cuda.is_available() or _native.device_count() == 0:
        raise RuntimeError("prosstt_amd needs an AMD MI355X (gfx950) device: there is no CPU fallback")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Context:
    """Owns a prosstt_amd_ctx bound to ``device`` and to torch's current stream there."""

    def __init__(self, device=None):
        require_gpu()
        torch = _torch()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.torch_device = torch.device("cuda", self.device)
        self.stream = torch.cuda.current_stream(self.device)
        self._lib = _native.load()
        handle = ctypes.c_void_p()
        _native.check(self._lib.prosstt_amd_ctx_create(
            self.device, ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(handle)))
        self._h = handle
        self._checked_means = None      # (pointer, rows, G, token) of the mean tensor whose row flags the ctx holds

    def close(self):
        if getattr(self, "_h", None):
            self._lib.prosstt_amd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ------------------------------------------------------------
    def tensor(self, array, dtype):
        """Host array (or tensor) -> contiguous device tensor of ``dtype``."""
        torch = _torch()
        if isinstance(array, torch.Tensor):
            return array.to(device=self.torch_device, dtype=dtype).contiguous()
        host = np.ascontiguousarray(array)
        if not host.flags.writeable:               # e.g. a broadcast view: torch wants an owned buffer
            host = host.copy()
        return torch.as_tensor(host, dtype=dtype).to(self.torch_device)

    def synchronize(self):
        _native.check(self._lib.prosstt_amd_ctx_synchronize(self._h))

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        _native.check(self._lib.prosstt_amd_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    # ---- K3: fused count sampler -------------------------------------------
    def sample_counts(self, means, row_of_cell, scaling, alpha, beta, seed, cell_offset=0,
                      out=None, check_domain=True, time_kernel=False, cell_index=None, means_token=None):
        """int32 device tensor (N, G) of counts; see prosstt_amd_sample_counts.

        check_domain  True: the reference's argument check, raised here (ValueError; synchronises);
                      "deferred": the same check, enqueued with the call and not waited for -- the verdict is
                      raised by the next ``domain_status()`` (the host-returning entry points call it behind the
                      copy that synchronises anyway); False: no check.
        means_token   any hashable that changes whenever the content of ``means`` does (the host layer passes the
                      tree's fingerprint): with the same tensor and token as the previous checked call the per-row
                      flags of the mean tensor are reused instead of rescanned."""
        torch = _torch()
        means = self.tensor(means, torch.float32)
        rows, G = means.shape
        row_of_cell = self.tensor(row_of_cell, torch.int32)
        N = row_of_cell.numel()
        scaling = self.tensor(scaling, torch.float64)
        nonneg = bool(check_domain) and self._params_nonneg(alpha, beta)
        alpha = self.tensor(alpha, torch.float64)
        beta = self.tensor(beta, torch.float64)
        if scaling.numel() != N or alpha.numel() != G or beta.numel() != G:
            raise ValueError("scaling must have one entry per cell, alpha/beta one per gene")
        if out is None:
            out = torch.empty((N, G), dtype=torch.int32, device=self.torch_device)
        elif out.dtype != torch.int32 or out.shape != (N, G) or out.stride(1) != 1:
            raise ValueError("out must be an int32 (N, G) tensor with unit column stride")
        if cell_index is not None:
            cell_index = self.tensor(cell_index, torch.int64)
            if cell_index.numel() != N:
                raise ValueError("cell_index must have one entry per cell")
        # (row_of_cell indexes the mean tensor unchecked in the kernels -- device pointers are the caller's
        # contract in the C ABI; with check_domain the library's domain pass also reports an index outside
        # the tensor, as PROSSTT_AMD_EINVAL)
        if check_domain not in (True, False, "deferred"):
            raise ValueError("check_domain must be True, False or 'deferred'")
        flags = _native.TIME_KERNEL if time_kernel else 0
        if check_domain:
            flags |= _native.CHECK_DEFERRED if check_domain == "deferred" else _native.CHECK_DOMAIN
            if nonneg:
                flags |= _native.PARAMS_NONNEG
            key = (means.data_ptr(), rows, G, means_token)
            if means_token is not None and key == self._checked_means:
                flags |= _native.MEANS_CACHED
            self._checked_means = None              # the ctx holds this tensor's row flags only once the call has gone through
        _native.check(self._lib.prosstt_amd_sample_counts(
            self._h, _ptr(means), rows, G, _ptr(row_of_cell), _ptr(scaling), _ptr(alpha), _ptr(beta),
            N, ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(cell_offset), _ptr(cell_index),
            _ptr(out), out.stride(0) if N else G, flags))
        if check_domain and N and G and means_token is not None:
            self._checked_means = key
        return out

    def _params_nonneg(self, alpha, beta):
        """alpha >= 0 and beta >= 1 for every gene, as far as the HOST can tell for free (then alpha*m + beta < 1 cannot
        happen and the checked call tells the library so: PROSSTT_AMD_PARAMS_NONNEG).  Host arrays are looked at here
        (O(G)).  Device tensors are not looked at at all: the preparation kernel tests them anyway and the per-sample
        pass at the end of the call's second kernel runs only if it found a gene with alpha < 0 or beta < 1 -- no torch
        kernel (whose first use in a process costs tens of milliseconds of code loading: tools/cold_probe.py), no cache
        of verdicts that could outlive its tensor."""
        torch = _torch()
        for arr, floor in ((alpha, 0.0), (beta, 1.0)):
            if isinstance(arr, torch.Tensor) or not bool(np.all(np.asarray(arr) >= floor)):
                return False
        return True

    def domain_status(self):
        """Raise what the ``check_domain="deferred"`` calls since the last time found (ValueError where the
        reference's scipy call raises, NativeError for a row index outside the tensor); synchronises."""
        status = ctypes.c_int32(0)
        _native.check(self._lib.prosstt_amd_domain_status(self._h, ctypes.byref(status)))
        _native.check(status.value)

    def last_list(self, cap=1 << 22):
        """(cells, genes, total, overflowed): the samples the streaming kernel of the last
        sample_counts call left to its second kernel (see prosstt_amd_last_list)."""
        cells = np.empty(cap, np.int64)
        genes = np.empty(cap, np.int32)
        total = ctypes.c_int64(0)
        over = ctypes.c_int32(0)
        _native.check(self._lib.prosstt_amd_last_list(
            self._h, cells.ctypes.data_as(ctypes.c_void_p), genes.ctypes.data_as(ctypes.c_void_p), cap,
            ctypes.byref(total), ctypes.byref(over)))
        n = min(cap, total.value)
        return cells[:n], genes[:n], total.value, bool(over.value)

    HW_OPS = dict(rcp=0, log2=1, exp2neg=2)

    def hw_math(self, op, first_bits, count):
        """float32 host array y[i] = f(as_float(first_bits + i)), i < count, of one of the three hardware
        functions of the sampler's definition ('rcp', 'log2', 'exp2neg'), computed by the device
        (prosstt_amd_hw_math): the tables the checking model is given."""
        out = np.empty(int(count), np.float32)
        _native.check(self._lib.prosstt_amd_hw_math(
            self._h, self.HW_OPS[op], ctypes.c_uint32(first_bits), ctypes.c_uint64(int(count)),
            out.ctypes.data_as(ctypes.c_void_p), _native.HOST_OUTPUT))
        return out

    HW_OPS_AT = dict(rcp=0, log2=1, exp2neg=2, sqrt=3, rsq=4, cos=5)

    def hw_math_at(self, op, x):
        """float32 host array y[i] = f(x[i]) of one of the hardware functions of the sampler's definition (a name of
        HW_OPS_AT or its code; 'cos' takes revolutions), computed by the device (prosstt_amd_hw_math_at): what the
        checking model asks while it evaluates the gamma-Poisson class."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.size, np.float32)
        code = self.HW_OPS_AT[op] if isinstance(op, str) else int(op)
        _native.check(self._lib.prosstt_amd_hw_math_at(
            self._h, code, x.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(x.size),
            out.ctypes.data_as(ctypes.c_void_p), _native.HOST_OUTPUT | _native.HOST_INPUTS))
        return out

    def nb_params(self, means, row_of_cell, scaling, alpha, beta):
        """(mu, p, r, path) device tensors (N, G) -- the sampler's deterministic intermediates."""
        torch = _torch()
        means = self.tensor(means, torch.float32)
        rows, G = means.shape
        row_of_cell = self.tensor(row_of_cell, torch.int32)
        N = row_of_cell.numel()
        scaling = self.tensor(scaling, torch.float64)
        alpha = self.tensor(alpha, torch.float64)
        beta = self.tensor(beta, torch.float64)
        mu = torch.empty((N, G), dtype=torch.float32, device=self.torch_device)
        p = torch.empty_like(mu)
        r = torch.empty_like(mu)
        path = torch.empty((N, G), dtype=torch.int32, device=self.torch_device)
        _native.check(self._lib.prosstt_amd_nb_params(
            self._h, _ptr(means), rows, G, _ptr(row_of_cell), _ptr(scaling), _ptr(alpha), _ptr(beta),
            N, _ptr(mu), _ptr(p), _ptr(r), _ptr(path), 0))
        return mu, p, r, path

    # ---- K2: lineage ---------------------------------------------------------
    def lineage_attempt_batch(self, programs, H, sib_programs=()):
        """B attempts of one branch in one launch: ``programs`` is (B, T, K); returns
        (max(programs[b]@H) as (B,) float64, #genes with r<0 per sibling as (B, n_sib) int64)
        -- simulation.py:269-272 per attempt."""
        programs = np.ascontiguousarray(programs, np.float64)
        B, T, K = programs.shape
        sibs = [np.ascontiguousarray(s, np.float64) for s in sib_programs]
        n = len(sibs)
        for s in sibs:
            if s.shape[1] != K:
                raise ValueError("sibling programs must have %d columns" % K)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[s.ctypes.data for s in sibs])
        lens = (ctypes.c_int32 * max(n, 1))(*[s.shape[0] for s in sibs])
        top = np.empty(B, np.float64)
        counts = np.zeros((B, max(n, 1)), np.int64)
        _native.check(self._lib.prosstt_amd_lineage_attempt_batch(
            self._h, programs.ctypes.data_as(ctypes.c_void_p), B, T, K, _ptr(H), H.shape[1], n,
            ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(lens, ctypes.c_void_p),
            top.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p)))
        return top, counts[:, :n]

    def lineage_attempt(self, programs, H, sib_programs=()):
        """(max(programs@H), [#genes with r<0 per sibling]) -- simulation.py:269-272."""
        top, counts = self.lineage_attempt_batch(np.asarray(programs, np.float64)[None], H, sib_programs)
        return float(top[0]), [int(c) for c in counts[0]]

    def lineage_walk(self, seed, stream_id, T, K):
        """(T, K) float64 expression programs drawn on the device (K1, PRLW-1)."""
        out = np.empty((T, K), np.float64)
        _native.check(self._lib.prosstt_amd_lineage_walk(
            self._h, ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(stream_id), T, K,
            out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def lineage_walks(self, seed, stream_ids, T, K):
        """(B, T, K) programs of the walk streams ``stream_ids`` (consecutive ids: one launch, one copy back)."""
        ids = [int(i) for i in stream_ids]
        if not ids:
            return np.empty((0, T, K), np.float64)
        if ids != list(range(ids[0], ids[0] + len(ids))):
            return np.stack([self.lineage_walk(seed, i, T, K) for i in ids])
        out = np.empty((len(ids), T, K), np.float64)
        _native.check(self._lib.prosstt_amd_lineage_walk_batch(
            self._h, ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(ids[0]), len(ids), T, K,
            out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def lineage_commit(self, programs, H, rel_out=None, gene_max=None):
        programs = np.ascontiguousarray(programs, np.float64)
        T, K = programs.shape
        _native.check(self._lib.prosstt_amd_lineage_commit(
            self._h, programs.ctypes.data_as(ctypes.c_void_p), T, K, _ptr(H), H.shape[1],
            _ptr(rel_out), _ptr(gene_max)))

    def gene_max(self, rel, gene_max):
        """gene_max[g] = max(gene_max[g], max over rows of rel[:, g]) -- sim_utils.py:423-425 in log space."""
        rel = rel.contiguous()
        _native.check(self._lib.prosstt_amd_gene_max(self._h, _ptr(rel), rel.shape[0], rel.shape[1], _ptr(gene_max)))
        return gene_max

    def means_from_rel(self, rel, base, out=None):
        torch = _torch()
        rows, G = rel.shape
        if out is None:
            out = torch.empty((rows, G), dtype=torch.float32, device=self.torch_device)
        _native.check(self._lib.prosstt_amd_means_from_rel(self._h, _ptr(rel), _ptr(base), rows, G, _ptr(out)))
        return out


def plan_order(row_of_cell, rows=None):
    """int32 array ``order``: the cells grouped by their row of the mean tensor (stable): the order of presentation that
    keeps a gene tile's rows of the mean tensor in cache (prosstt_amd_plan_order; host arrays, no device work).  Present
    cell ``order[i]`` at position i with ``cell_index[i]`` = its global id; row i of the result is that cell's."""
    roc = np.ascontiguousarray(row_of_cell, dtype=np.int32)
    order = np.empty(roc.size, dtype=np.int32)
    if roc.size:
        n_rows = int(roc.max()) + 1 if rows is None else int(rows)
        _native.check(_native.load().prosstt_amd_plan_order(
            roc.ctypes.data_as(ctypes.c_void_p), roc.size, n_rows, order.ctypes.data_as(ctypes.c_void_p)))
    return order


def host_fingerprint(arrays):
    """Identity and content fingerprint of host arrays that a device tensor mirrors: the caches of
    ``Tree.device_means`` and of ``simulate_lineage`` compare it before they trust their device copy,
    so that the caller's arrays stay writable (as the reference's are) and an in-place edit is seen.
    One pass at memory speed (64 MB in ~6 ms)."""
    try:
        from xxhash import xxh3_64_intdigest as digest
    except ImportError:                              # slower (64 MB in ~60 ms), as position-sensitive
        import zlib

        def digest(buf):
            view = memoryview(buf).cast("B")
            return (zlib.crc32(view) << 32) | zlib.adler32(view)
    out = []
    for a in arrays:
        if isinstance(a, np.ndarray):
            c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
            out.append((id(a), a.shape, a.dtype.str, digest(c.data if c.size else b"")))
        else:
            out.append((id(a), None, None, None))
    return tuple(out)


PINNED_RETURN_MAX = int(os.environ.get("PROSSTT_AMD_PINNED_MAX_BYTES", str(32 << 30)))


HOST_DTYPES = {"numpy": np.int64, "numpy32": np.int32, "numpy16": np.uint16}


def to_host(counts, dtype=np.int64, chunk_bytes=256 << 20, row_order=None):
    """int32 device counts -> host ndarray of ``dtype``: int64 (the reference's return type,
    simulation.py:651), int32 (what the device holds) or uint16 (raises OverflowError if a count does not fit: C3's
    largest is 84 036).

    int64 and int32 of 2^24 counts or more (``_to_host_widened``): the matrix travels a chunk of rows at a time by
    asynchronous copies on a second stream, in a WIRE format -- the low 8 (or 16) bits of every count, the few entries with
    higher bits set beside them as (position, value) pairs, or the int32 as it lies (``WIRE``) -- into two page-locked
    bounce buffers, and the host library's threads widen chunk i - 1 into the result while chunk i is on the bus; the
    result lies in pageable memory on huge pages that is recycled once the caller has dropped it (``RESULT_MEMORY``,
    ``_result_array``).  Everything else -- uint16, smaller matrices, ``WIDEN_ON`` = "device" -- as until round 5: the
    chunks land by DMA straight in a page-locked result from torch's caching host allocator (at most
    PROSSTT_AMD_PINNED_MAX_BYTES, default 32 GiB; beyond that, or when page-locking fails, in ordinary memory); int64 is
    formed and uint16 narrowed on the device, chunk by chunk (two staging buffers) under the transfer of the previous chunk,
    int32 is copied as it lies.

    row_order: the device matrix holds its cells in an order of PRESENTATION (``plan_order``): row i is cell
    ``row_order[i]``.  The host array comes back in plan order -- row ``row_order[i]`` = device row i -- the rows of every
    chunk gathered on the device into the staging buffer that the conversion uses anyway (a gather of whole rows under
    the transfer of the previous chunk: nothing is added to the PCIe-bound copy)."""
    torch = _torch()
    dtype = np.dtype(dtype)
    n, g = counts.shape
    if dtype not in (np.dtype(np.int64), np.dtype(np.int32), np.dtype(np.uint16)):
        raise ValueError("host counts are int64, int32 or uint16")
    if n == 0 or g == 0:
        return np.zeros((n, g), dtype=dtype)
    if dtype.itemsize >= 4 and WIDEN_ON == "host" and n * g >= (1 << 24):
        for wire in WIRES[WIRES.index(WIRE):] if WIRE in WIRES else WIRES[2:]:
            if wire == "i32" and dtype.itemsize == 4:
                break                                   # (int32 as it lies needs no widening: the copy below)
            try:
                return _to_host_widened(counts, chunk_bytes, row_order, dtype, wire)
            except _WireTooNarrow:
                continue
            except _NoBounceBuffers:
                break                                   # (the device-widened copy below needs no page-locked memory)
    # (torch has no arithmetic on uint16: the device narrows to int16 bit patterns, viewed as uint16 on the host)
    t_dtype = {8: torch.int64, 4: torch.int32, 2: torch.int16}[dtype.itemsize]
    host = None
    if n * g * dtype.itemsize <= PINNED_RETURN_MAX:
        try:
            host = torch.empty((n, g), dtype=t_dtype, pin_memory=True)
        except RuntimeError:
            host = None
    if host is None:
        host = torch.empty((n, g), dtype=t_dtype)
    rows = max(1, min(n, int(chunk_bytes) // (g * dtype.itemsize)))
    dev = counts.device
    compute = torch.cuda.current_stream(dev)
    copier = torch.cuda.Stream(dev)
    inv = None
    if row_order is not None:
        order = np.asarray(row_order, dtype=np.int64)
        if order.shape != (n,):
            raise ValueError("row_order must have one entry per row")
        inv_host = np.empty(n, dtype=np.int64)
        inv_host[order] = np.arange(n, dtype=np.int64)        # device row of host row j
        inv = torch.as_tensor(inv_host).to(dev)
    convert = t_dtype != torch.int32 or inv is not None
    staging = [torch.empty((rows, g), dtype=t_dtype, device=dev) for _ in range(2 if rows < n else 1)] if convert else []
    # (gathered rows of a chunk that is also widened or narrowed: ONE int32 scratch for the whole copy, not a fresh
    # temporary per chunk beside a count matrix that may fill most of the device)
    gathered = torch.empty((rows, g), dtype=torch.int32, device=dev) if (inv is not None and t_dtype != torch.int32) else None
    copied = [None, None]
    too_big = torch.zeros((), dtype=torch.int32, device=dev) if dtype.itemsize == 2 else None
    for i, lo in enumerate(range(0, n, rows)):
        hi = min(lo + rows, n)
        if convert:
            slot = i & 1
            if copied[slot] is not None:
                compute.wait_event(copied[slot])           # the buffer's previous chunk has left
            stage = staging[slot][:hi - lo]
            if too_big is not None:
                too_big = torch.maximum(too_big, counts[lo:hi].max())
            if inv is None:
                stage.copy_(counts[lo:hi])                  # int32 -> int64, or the low 16 bits
            elif t_dtype == torch.int32:
                torch.index_select(counts, 0, inv[lo:hi], out=stage)
            else:
                torch.index_select(counts, 0, inv[lo:hi], out=gathered[:hi - lo])
                stage.copy_(gathered[:hi - lo])
        else:
            stage = counts[lo:hi]
        ready = torch.cuda.Event()
        ready.record(compute)
        copier.wait_event(ready)
        with torch.cuda.stream(copier):
            host[lo:hi].copy_(stage, non_blocking=True)
            if convert:
                copied[slot] = torch.cuda.Event()
                copied[slot].record(copier)
    copier.synchronize()
    if too_big is not None and int(too_big) > 65535:
        raise OverflowError("a count of %d does not fit uint16: ask for 'numpy32'" % int(too_big))
    out = host.numpy()
    return out.view(np.uint16) if dtype.itemsize == 2 else out


# How an int64 / int32 host return of 2^24 counts or more travels.  WIDEN_ON = "host" (default): the matrix crosses PCIe in
# a WIRE format narrower than its type and the host library's threads (libprosstt_amd_host.so) widen chunk i - 1 into the
# result while chunk i is on the bus; "device": as rounds 2 to 5 did -- int64 formed on the device, 8 bytes per count over
# PCIe, int32 copied as it lies (the choice for a host with very few cores).  WIRE = "u8" (default): the low 8 bits of
# every count, 1 byte over PCIe, and beside them the entries that have higher bits set (two thirds of a count matrix are
# zeros; one count in a thousand of C3 is above 255, the largest is 84 036) as (position, value) pairs that the pool writes
# over the widened matrix at the end; a matrix in which more than one entry in 256 of a chunk is such an exception is sent
# again with the next wider wire: "u16" (the low 16 bits), then "i32" (the int32 as it lies).  WIRE names the narrowest
# wire that is tried.
# (tools/host_widen_probe.py, tools/e2e_threads.py on an MI355X box's host: the pool writes 250 - 340 GB/s from eight threads
# on; PCIe carries 52 GB/s.  C3 end to end: int64 147 -> 39 - 51 ms, int32 76 -> 32 - 39 ms, out="csr" 58 -> 38 ms.)
WIDEN_ON = os.environ.get("PROSSTT_AMD_WIDEN", "host")
WIRE = os.environ.get("PROSSTT_AMD_WIRE", "u8")
WIRES = ("u8", "u16", "i32")
RESULT_MEMORY = os.environ.get("PROSSTT_AMD_RESULT_MEMORY", "pageable")
HOST_THREADS = int(os.environ.get("PROSSTT_AMD_HOST_THREADS", str(max(1, min(16, os.cpu_count() or 1)))))


_result_blocks = []            # pageable result memory handed out before: [uint8 ndarray]; see _result_array
_result_lock = threading.Lock()
RESULT_CACHE_BYTES = int(os.environ.get("PROSSTT_AMD_RESULT_CACHE_BYTES", str(40 << 30)))


def _result_array(shape, dtype):
    """A pageable (n, g) result of the host-widened copy.  The memory is a numpy allocation (numpy asks for transparent
    huge pages: the pool's first touch of 8 GB costs 45 ms instead of the 500 ms of 4 KB pages) that this module keeps a
    reference to: once the caller has dropped the result and every view of it -- the block's reference count says so --
    the next result of that size or less is laid over the same, already touched pages, as torch's caching host allocator
    does for page-locked memory, without its 0.6 - 0.9 s of page-locking in front of the first 8 GB result."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    with _result_lock:
        block = cached = None
        for i in range(len(_result_blocks)):
            cached = _result_blocks[i]
            # (references: the list's, this variable's, getrefcount's own argument)
            if cached.size >= nbytes and cached.size <= 2 * nbytes + (1 << 20) and sys.getrefcount(cached) == 3:
                block = _result_blocks.pop(i)
                break
        del cached
        if block is None:
            block = np.empty(nbytes, dtype=np.uint8)
        _result_blocks.append(block)
        held = sum(b.size for b in _result_blocks)
        while len(_result_blocks) > 1 and held > RESULT_CACHE_BYTES:
            held -= _result_blocks.pop(0).size         # (the oldest; its memory goes when its last user does)
    return block[:nbytes].view(dtype).reshape(shape)


def release_result_memory():
    """Forget the recycled result blocks (PROSSTT_AMD_RESULT_CACHE_BYTES bounds what is kept: 40 GiB): the memory of the
    ones nobody holds goes back to the system at once, the others' when their last user drops them."""
    with _result_lock:
        del _result_blocks[:]


class _WireTooNarrow(Exception):
    """Too many entries of a chunk do not fit the wire: the copy starts again with the next wider one."""


class _NoBounceBuffers(Exception):
    """The page-locked bounce buffers of the host-widened copy were refused."""


def _to_host_widened(counts, chunk_bytes, row_order, dtype, wire):
    """``to_host`` for int64 / int32 with the widening on the host: the chunks arrive in their wire format (int32 as the
    device holds it, or narrowed to uint16 on the device) in two page-locked bounce buffers by asynchronous copies on a
    second stream; chunk i - 1 is widened into the result by HOST_THREADS threads of the host library's pool
    (libprosstt_amd_host.so, include/prosstt_amd_host.h: non-temporal AVX2 stores) while chunk i is on the bus.  The copy
    is bound by the wire's 4 or 2 bytes per count over PCIe, not by the 8 or 4 of the result."""
    torch = _torch()
    n, g = (int(v) for v in counts.shape)
    t_out = torch.int64 if dtype.itemsize == 8 else torch.int32
    t_wire = {"u8": torch.uint8, "u16": torch.int16, "i32": torch.int32}[wire]     # (uint16 bit patterns: torch has no arithmetic on uint16)
    high_bits = {"u8": -256, "u16": -65536, "i32": 0}[wire]
    narrow = wire != "i32"
    lib = _native.load_host()
    widen = {("i32", 8): lib.prosstt_amd_host_widen_i32_i64, ("u16", 8): lib.prosstt_amd_host_widen_u16_i64,
             ("u16", 4): lib.prosstt_amd_host_widen_u16_i32, ("u8", 8): lib.prosstt_amd_host_widen_u8_i64,
             ("u8", 4): lib.prosstt_amd_host_widen_u8_i32}[(wire, dtype.itemsize)]
    # The result is written by the host's threads, not by DMA: it needs no page-locking.  RESULT_MEMORY = "pageable"
    # (default): numpy memory on transparent huge pages, recycled once the caller has dropped the result (_result_array) --
    # C3 end to end: first call of a process 160 - 420 ms, then 40 - 50 ms; "pinned": from torch's caching host allocator --
    # 0.5 to 0.9 s to page-lock C3's 8 GB the first time a size is asked for (first call 560 - 960 ms, then 54 - 80 ms;
    # tools/hugepage_probe.py, tools/first_call.py).
    host = None
    if RESULT_MEMORY == "pinned" and n * g * dtype.itemsize <= PINNED_RETURN_MAX:
        try:
            host = torch.empty((n, g), dtype=t_out, pin_memory=True).numpy()
        except RuntimeError:
            host = None
    if host is None:
        host = _result_array((n, g), dtype)
    rows = max(1, min(n, int(chunk_bytes) // (g * 4)))
    dev = counts.device
    compute = torch.cuda.current_stream(dev)
    copier = torch.cuda.Stream(dev)
    inv = None
    if row_order is not None:
        order = np.asarray(row_order, dtype=np.int64)
        if order.shape != (n,):
            raise ValueError("row_order must have one entry per row")
        inv_host = np.empty(n, dtype=np.int64)
        inv_host[order] = np.arange(n, dtype=np.int64)        # device row of host row j
        inv = torch.as_tensor(inv_host).to(dev)
    slots = 2 if rows < n else 1
    try:
        bounce = [torch.empty((rows, g), dtype=t_wire, pin_memory=True) for _ in range(slots)]
    except RuntimeError as exc:
        raise _NoBounceBuffers() from exc
    # device staging: the chunk's rows gathered (int32), and -- for the uint16 wire -- narrowed
    gathered = torch.empty((rows, g), dtype=torch.int32, device=dev) if (inv is not None and narrow) else None
    staged = [torch.empty((rows, g), dtype=t_wire, device=dev) for _ in range(slots)] if (inv is not None or narrow) else None
    arrived = [None, None]
    bounds = list(range(0, n, rows)) + [n]
    host_at = host.ctypes.data
    # A narrow wire: what did not fit travels beside the chunk as (position in the host matrix, value) pairs -- up to one
    # entry in 256 of the chunk, in fixed-size buffers, so that nothing on the way waits for the device (the number of
    # exceptions of a chunk is read when the chunk has arrived; more than fit: the copy starts again with a wider wire).
    exceptions = []
    room = max(1, rows * g // 256)
    if narrow:
        exc_where = [torch.empty(room, dtype=torch.int64, device=dev) for _ in range(slots)]
        exc_value = [torch.empty(room, dtype=torch.int32, device=dev) for _ in range(slots)]
        exc_count = [torch.empty((), dtype=torch.int64, device=dev) for _ in range(slots)]
        try:
            h_where = [torch.empty(room, dtype=torch.int64, pin_memory=True) for _ in range(slots)]
            h_value = [torch.empty(room, dtype=torch.int32, pin_memory=True) for _ in range(slots)]
            h_count = [torch.empty((), dtype=torch.int64, pin_memory=True) for _ in range(slots)]
        except RuntimeError as exc:
            raise _NoBounceBuffers() from exc

    def note_exceptions(block, lo, slot):
        """The entries of the chunk (int32, rows in host order from row ``lo``) with higher bits set, into the slot's buffers."""
        flat = block.reshape(-1)
        high = torch.bitwise_and(flat, high_bits)
        exc_count[slot].copy_(torch.count_nonzero(high))
        where = torch.nonzero_static(high, size=room, fill_value=0).squeeze(1)     # (padded with position 0: written with its own value)
        torch.index_select(flat, 0, where, out=exc_value[slot])
        torch.add(where, lo * g, out=exc_where[slot])

    def widen_chunk(i):
        lo, hi = bounds[i], bounds[i + 1]
        slot = i % slots
        arrived[slot].synchronize()
        if narrow:
            k = int(h_count[slot])
            if k > room or k * 256 > (hi - lo) * g:
                copier.synchronize()                    # (nothing of this attempt is in flight when its buffers go back)
                torch.cuda.current_stream(dev).synchronize()
                raise _WireTooNarrow()
            if k:
                exceptions.append((h_where[slot][:k].clone(), h_value[slot][:k].clone()))
        if widen(ctypes.c_void_p(bounce[slot].data_ptr()), ctypes.c_void_p(host_at + lo * g * dtype.itemsize),
                 ctypes.c_uint64((hi - lo) * g), HOST_THREADS) != 0:
            raise RuntimeError("the host library refused its arguments")

    copier.wait_stream(compute)                        # the matrix itself
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        slot = i % slots
        if staged is None:
            src = counts[lo:hi]
        else:
            if arrived[slot] is not None:
                compute.wait_event(arrived[slot])       # the staging buffer's previous chunk has left
            src = staged[slot][:hi - lo]
            if inv is None:
                src.copy_(counts[lo:hi])                # the low bits
                note_exceptions(counts[lo:hi], lo, slot)
            elif not narrow:
                torch.index_select(counts, 0, inv[lo:hi], out=src)
            else:
                torch.index_select(counts, 0, inv[lo:hi], out=gathered[:hi - lo])
                src.copy_(gathered[:hi - lo])
                note_exceptions(gathered[:hi - lo], lo, slot)
            ready = torch.cuda.Event()
            ready.record(compute)
            copier.wait_event(ready)
        # (the bounce buffer's previous chunk, i - 2, was widened in the last turn of this loop)
        with torch.cuda.stream(copier):
            bounce[slot][:hi - lo].copy_(src, non_blocking=True)
            if narrow:
                h_count[slot].copy_(exc_count[slot], non_blocking=True)
                h_where[slot].copy_(exc_where[slot], non_blocking=True)
                h_value[slot].copy_(exc_value[slot], non_blocking=True)
            arrived[slot] = torch.cuda.Event()
            arrived[slot].record(copier)
        if i >= 1:
            widen_chunk(i - 1)
    widen_chunk(len(bounds) - 2)
    if exceptions:
        where = np.ascontiguousarray(torch.cat([e[0] for e in exceptions]).numpy())
        values = np.ascontiguousarray(torch.cat([e[1] for e in exceptions]).numpy())
        if lib.prosstt_amd_host_scatter_i32(ctypes.c_void_p(host_at), dtype.itemsize, ctypes.c_void_p(where.ctypes.data),
                                            ctypes.c_void_p(values.ctypes.data), ctypes.c_uint64(where.size), HOST_THREADS) != 0:
            raise RuntimeError("the host library refused its arguments")
    return host


def to_host_csr(counts, chunk_bytes=256 << 20, row_order=None, _narrow=(True, True)):
    """int32 device counts -> ``scipy.sparse.csr_matrix`` (int32 data, int32 column indices sorted within a row, int64
    row pointers) in plan order: what the single-cell toolchains downstream of the reference's count files hold a count
    matrix in.  Two thirds of such a matrix are zeros, and only the NON-ZEROS cross PCIe -- 3 bytes each (the value's low 8
    bits, a 16-bit column index; 8 bytes for a matrix of large counts or of more than 65 536 columns), widened by the host
    library's threads -- and the dense matrix never exists on the host.

    Two passes over the device matrix, a chunk of rows at a time: the non-zeros per row first (the row pointers, and with
    them the exact size of the page-locked result), then every chunk's non-zeros are compacted on the device (row-major:
    rows ascending, columns ascending within a row) into one of two staging pairs and copied on a second stream straight
    into their final place while the next chunk is compacted.  ``row_order``: as in ``to_host`` (device row i is cell
    ``row_order[i]``; the gather of whole rows rides in the first step of each chunk)."""
    torch = _torch()
    import scipy.sparse as sparse
    n, g = (int(v) for v in counts.shape)
    if n == 0 or g == 0:
        return sparse.csr_matrix((n, g), dtype=np.int32)
    if g >= 2 ** 31:
        raise ValueError("column indices are int32")
    dev = counts.device
    rows = max(1, min(n, int(chunk_bytes) // (g * 4)))
    inv = None
    if row_order is not None:
        order = np.asarray(row_order, dtype=np.int64)
        if order.shape != (n,):
            raise ValueError("row_order must have one entry per row")
        inv_host = np.empty(n, dtype=np.int64)
        inv_host[order] = np.arange(n, dtype=np.int64)        # device row of host row j
        inv = torch.as_tensor(inv_host).to(dev)
    # pass 1: non-zeros of every row, in the order of the rows on the host
    per_row = torch.empty(n, dtype=torch.int64, device=dev)
    for lo in range(0, n, rows):
        hi = min(lo + rows, n)
        per_row[lo:hi] = torch.count_nonzero(counts[lo:hi], dim=1)
    if inv is not None:
        per_row = per_row.index_select(0, inv)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(per_row.cpu().numpy(), out=indptr[1:])
    total = int(indptr[-1])
    bounds = list(range(0, n, rows)) + [n]
    cap = max(int(indptr[b] - indptr[a]) for a, b in zip(bounds[:-1], bounds[1:]))

    def host_array(written_by_the_host):
        """A result array as a torch tensor: page-locked when DMA lands in it (or RESULT_MEMORY says so), else over a
        recycled pageable block (_result_array)."""
        if written_by_the_host and RESULT_MEMORY != "pinned":
            return torch.from_numpy(_result_array((total,), np.int32))
        if 0 < total * 4 <= PINNED_RETURN_MAX:
            try:
                return torch.empty(total, dtype=torch.int32, pin_memory=True)
            except RuntimeError:
                pass
        return torch.empty(total, dtype=torch.int32)

    compute = torch.cuda.current_stream(dev)
    copier = torch.cuda.Stream(dev)
    slots = 2 if len(bounds) > 2 else 1
    gathered = torch.empty((rows, g), dtype=torch.int32, device=dev) if inv is not None else None
    # The wire (as in ``to_host``): values as their low 8 bits -- the few above 255 beside them as (position, value) pairs --
    # and column indices as 16 bits when the matrix has at most 65 536 columns: 3 bytes per non-zero over PCIe instead of
    # 8, widened into the two int32 arrays by the host library's threads under the transfer of the next chunk.
    # (_narrow: which of the two this attempt may narrow -- the call starts again without the first for a matrix of large
    # counts, without both when the page-locked bounce buffers are refused)
    narrow_vals = _narrow[0] and WIDEN_ON == "host" and WIRE == "u8" and total >= (1 << 22)
    narrow_cols = _narrow[1] and WIDEN_ON == "host" and WIRE in ("u8", "u16") and total >= (1 << 22) and g <= 65536
    lib = _native.load_host() if (narrow_vals or narrow_cols) else None
    data, indices = host_array(narrow_vals), host_array(narrow_cols)
    t_vals = torch.uint8 if narrow_vals else torch.int32
    t_cols = torch.int16 if narrow_cols else torch.int32
    vals = [torch.empty(max(cap, 1), dtype=t_vals, device=dev) for _ in range(slots)]
    cols = [torch.empty(max(cap, 1), dtype=t_cols, device=dev) for _ in range(slots)]
    vals32 = torch.empty(max(cap, 1), dtype=torch.int32, device=dev) if narrow_vals else None
    try:
        bounce_v = [torch.empty(max(cap, 1), dtype=t_vals, pin_memory=True) for _ in range(slots)] if narrow_vals else None
        bounce_c = [torch.empty(max(cap, 1), dtype=t_cols, pin_memory=True) for _ in range(slots)] if narrow_cols else None
    except RuntimeError:
        return to_host_csr(counts, chunk_bytes, row_order, _narrow=(False, False))
    copied = [None, None]
    spans = [None, None]
    exceptions = []
    # (the values above 255 of a chunk: in fixed-size buffers beside it, their number read when the chunk has arrived --
    # as in ``_to_host_widened``; more than one in sixteen: 4-byte values after all)
    room = max(1, cap // 16)
    if narrow_vals:
        exc_where = [torch.empty(room, dtype=torch.int64, device=dev) for _ in range(slots)]
        exc_value = [torch.empty(room, dtype=torch.int32, device=dev) for _ in range(slots)]
        exc_count = [torch.empty((), dtype=torch.int64, device=dev) for _ in range(slots)]
        try:
            h_where = [torch.empty(room, dtype=torch.int64, pin_memory=True) for _ in range(slots)]
            h_value = [torch.empty(room, dtype=torch.int32, pin_memory=True) for _ in range(slots)]
            h_count = [torch.empty((), dtype=torch.int64, pin_memory=True) for _ in range(slots)]
        except RuntimeError:
            return to_host_csr(counts, chunk_bytes, row_order, _narrow=(False, False))

    class _LargeCounts(Exception):
        pass

    def widen_chunk(slot):
        """The narrow halves of the chunk in ``slot``: wait for them, widen them into their place."""
        if spans[slot] is None:
            return
        first, last = spans[slot]
        spans[slot] = None
        copied[slot].synchronize()
        k = last - first
        if narrow_vals:
            n_big = int(h_count[slot])
            if n_big > room or n_big * 16 > k:
                raise _LargeCounts()
            if n_big:
                exceptions.append((h_where[slot][:n_big].clone(), h_value[slot][:n_big].clone()))
        if narrow_vals and lib.prosstt_amd_host_widen_u8_i32(ctypes.c_void_p(bounce_v[slot].data_ptr()), ctypes.c_void_p(data.data_ptr() + 4 * first),
                                                             ctypes.c_uint64(k), HOST_THREADS) != 0:
            raise RuntimeError("the host library refused its arguments")
        if narrow_cols and lib.prosstt_amd_host_widen_u16_i32(ctypes.c_void_p(bounce_c[slot].data_ptr()), ctypes.c_void_p(indices.data_ptr() + 4 * first),
                                                              ctypes.c_uint64(k), HOST_THREADS) != 0:
            raise RuntimeError("the host library refused its arguments")

    # pass 2 (the number of non-zeros of every chunk is known from pass 1: torch.nonzero_static, no waiting for the device)
    def pass_two():
        for i, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            first, last = int(indptr[lo]), int(indptr[hi])
            k = last - first
            if k == 0:
                continue
            slot = i % slots
            if copied[slot] is not None:
                compute.wait_event(copied[slot])               # the staging pair's previous chunk has left
            if inv is None:
                block = counts[lo:hi]
            else:
                block = torch.index_select(counts, 0, inv[lo:hi], out=gathered[:hi - lo])
            flat = block.reshape(-1)
            where = torch.nonzero_static(flat, size=k, fill_value=0).squeeze(1)     # ascending flat positions = CSR order
            if narrow_vals:
                torch.index_select(flat, 0, where, out=vals32[:k])
                vals[slot][:k].copy_(vals32[:k])                # the low 8 bits
                high = torch.bitwise_and(vals32[:k], -256)
                exc_count[slot].copy_(torch.count_nonzero(high))
                big = torch.nonzero_static(high, size=room, fill_value=0).squeeze(1)
                torch.index_select(vals32[:k], 0, big, out=exc_value[slot])
                torch.add(big, first, out=exc_where[slot])
            else:
                torch.index_select(flat, 0, where, out=vals[slot][:k])
            cols[slot][:k].copy_(torch.remainder(where, g))
            del where
            widen_chunk(slot)                                  # (the bounce buffers' previous chunk, i - 2)
            ready = torch.cuda.Event()
            ready.record(compute)
            copier.wait_event(ready)
            with torch.cuda.stream(copier):
                (bounce_v[slot][:k] if narrow_vals else data[first:last]).copy_(vals[slot][:k], non_blocking=True)
                (bounce_c[slot][:k] if narrow_cols else indices[first:last]).copy_(cols[slot][:k], non_blocking=True)
                if narrow_vals:
                    h_count[slot].copy_(exc_count[slot], non_blocking=True)
                    h_where[slot].copy_(exc_where[slot], non_blocking=True)
                    h_value[slot].copy_(exc_value[slot], non_blocking=True)
                copied[slot] = torch.cuda.Event()
                copied[slot].record(copier)
            spans[slot] = (first, last)
            if slots == 2:
                widen_chunk(slot ^ 1)                          # chunk i - 1, while chunk i is on the bus
        copier.synchronize()
        for slot in range(slots):
            widen_chunk(slot)
    try:
        pass_two()
    except _LargeCounts:                                   # a matrix of large counts: 4-byte values after all
        copier.synchronize()
        compute.synchronize()
        return to_host_csr(counts, chunk_bytes, row_order, _narrow=(False, _narrow[1]))
    if exceptions:
        where = np.ascontiguousarray(torch.cat([e[0] for e in exceptions]).numpy())
        values = np.ascontiguousarray(torch.cat([e[1] for e in exceptions]).numpy())
        if lib.prosstt_amd_host_scatter_i32(ctypes.c_void_p(data.data_ptr()), 4, ctypes.c_void_p(where.ctypes.data),
                                            ctypes.c_void_p(values.ctypes.data), ctypes.c_uint64(where.size), HOST_THREADS) != 0:
            raise RuntimeError("the host library refused its arguments")
    out = sparse.csr_matrix((data.numpy(), indices.numpy(), indptr), shape=(n, g), copy=False)
    out.has_sorted_indices = True
    return out


HOST_OUTS = tuple(HOST_DTYPES) + ("csr",)
OUT_CHOICES = "out must be 'numpy', 'numpy32', 'numpy16', 'csr' or 'torch'"


def host_return(counts, out, row_order=None):
    """The host form ``out`` names (``HOST_OUTS``) of a device count matrix, rows in plan order."""
    if out == "csr":
        return to_host_csr(counts, row_order=row_order)
    return to_host(counts, HOST_DTYPES[out], row_order=row_order)


class Comm:
    """The one exchange of the path on RCCL directly, through the C ABI (prosstt_amd_comm_* / prosstt_amd_gather_counts:
    what a host binding without torch.distributed uses; prosstt_amd.parallel is the torch.distributed form).

        uid = Comm.unique_id()                 # on rank 0; carry the 128 bytes to the other ranks yourself
        comm = Comm(ctx, uid, rank, world)     # collective
        full = comm.gather_counts(counts, rows_of_rank, root=0)      # (sum(rows_of_rank), G) on the root, None elsewhere:
                                                                     # rank 0's rows, then rank 1's, ... (shard order)
    """

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _native.check(_native.load().prosstt_amd_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("the unique id is 128 bytes")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        _native.check(self._lib.prosstt_amd_comm_init(ctx._h, ctypes.c_char_p(unique_id), self.rank, self.world,
                                                      ctypes.byref(self._h)))

    def gather_counts(self, counts, rows_of_rank, root=0):
        torch = _torch()
        rows_of_rank = np.ascontiguousarray(rows_of_rank, dtype=np.int64)
        if rows_of_rank.shape != (self.world,) or int(rows_of_rank[self.rank]) != counts.shape[0]:
            raise ValueError("rows_of_rank must hold every rank's row count, this rank's equal to its rows")
        counts = counts.contiguous()
        G = int(counts.shape[1])
        dst = torch.empty((int(rows_of_rank.sum()), G), dtype=torch.int32, device=counts.device) if self.rank == root else None
        _native.check(self._lib.prosstt_amd_gather_counts(
            self.ctx._h, self._h, ctypes.c_void_p(counts.data_ptr()), rows_of_rank.ctypes.data_as(ctypes.c_void_p), G, int(root),
            ctypes.c_void_p(dst.data_ptr()) if dst is not None else None))
        return dst

    def selftest(self, nbytes=1 << 20):
        _native.check(self._lib.prosstt_amd_comm_selftest(self.ctx._h, self._h, ctypes.c_uint64(int(nbytes))))

    def close(self):
        if self._h:
            self._lib.prosstt_amd_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PresentedCounts:
    """The count matrix as it lies on the device after a call that PRESENTED its cells grouped by their row of the mean
    tensor (what keeps a gene tile's rows of the tensor in cache: 2 to 5 % of the kernel, and 1.75 x less HBM traffic on a
    32-branch tree): ``counts`` is the (N, G) int32 device tensor, row i of it belongs to cell ``cell_of_row[i]`` of the
    plan (the order of the pseudotime / branch / scaling arrays the call returns).  Every count is keyed by the cell's
    position in the plan, so ``counts`` holds exactly the rows of the plan-ordered matrix, permuted.

        counts, cell_of_row = presented                  # unpacks like a pair
        presented.in_plan_order()                        # a new device tensor, rows in plan order (one gather pass)
        presented.to_host("numpy32")                     # host ndarray in plan order (the gather rides in the copy)
        presented.row_of_cell                            # the inverse permutation
    """

    def __init__(self, counts, cell_of_row):
        self.counts = counts
        self.cell_of_row = np.asarray(cell_of_row, dtype=np.int64)
        self._row_of_cell = None

    def __iter__(self):
        return iter((self.counts, self.cell_of_row))

    def __len__(self):
        return 2

    @property
    def shape(self):
        return tuple(self.counts.shape)

    @property
    def row_of_cell(self):
        if self._row_of_cell is None:
            inv = np.empty_like(self.cell_of_row)
            inv[self.cell_of_row] = np.arange(self.cell_of_row.size, dtype=np.int64)
            self._row_of_cell = inv
        return self._row_of_cell

    def in_plan_order(self):
        torch = _torch()
        return self.counts.index_select(0, torch.as_tensor(self.row_of_cell).to(self.counts.device))

    def to_host(self, out="numpy"):
        return host_return(self.counts, out, row_order=self.cell_of_row)


def to_host_int64(counts, chunk_bytes=256 << 20):
    """int32 device counts -> the reference's int64 ndarray (simulation.py:651); see ``to_host``."""
    return to_host(counts, np.int64, chunk_bytes)


_contexts = {}


def get_context(device=None):
    """Process-wide Context of a device, bound to torch's current stream at first use."""
    torch = _torch()
    require_gpu()
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _contexts:
        _contexts[key] = Context(dev)
    return _contexts[key]
