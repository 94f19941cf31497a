#!/usr/bin/env python3
"""
bench.py -- simulated cells x genes / second of the PROSSTT sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(run without a launcher, ``--gpus N`` with N > 1 starts that launcher itself as a child process)

Workload (BASELINE.json metric): config C3 -- 8-branch tree (T = 50 per branch, K = 25
programs), 20 000 genes, 50 000 cells PER GPU; the tree goes through the product's own
lineage stage on the device (timed separately, not part of the metric), the cells are drawn
from the tree's density.  A *step* = one pass of the fused count sampler
(prosstt_amd_sample_counts: parameter prep + cell records + K3 stream kernel + K3h) over the
rank's cells, all inputs resident in HBM, output left in HBM.  With N GPUs the plan is sharded by
branch (prosstt_amd.parallel), no collective on the data path: 50 000 x N cells under
``--scaling weak`` (default), the configuration's own cell count under ``--scaling strong``
(e.g. ``--config C4 --scaling strong``: 200 000 cells over the N GPUs).  With N > 1 the line also
carries ``gather_ms`` (the one exchange of the path: count rows to rank 0, point-to-point) and
``strong_scaling``: C4 (200 000 cells) and C5 (1 000 000 cells) split over the N GPUs.

One JSON line on rank 0.
 * ``ms_per_step`` / ``value`` (= ``ms_per_step_strict``): the sampler as the product API calls it by default
   (``strict=True``: the reference's argument check rides in the call's own kernels, its verdict is read
   behind the timed steps); ``ms_per_step_unchecked``: the same without the check; ``ms_per_step_cold``: the
   first five steps of the process, before any clock ramp (what a user's first calls see).
 * ``roofline``: the dominant kernel (k3::sample_counts_stream_kernel): algorithmic bytes per
   launch / its mean duration from HIP events on the launch stream;
   ``frac_whole_step`` prices the same bytes against ms_per_step (K3h, prep kernels, launch gaps
   included).  ``traffic`` comes from the committed rocprofv3 PMC passes and is reported only
   while the kernel sources are the ones that were profiled.
 * ``cpu_baseline`` (N = 1 only): the oracle's restatement of the reference's draw_counts WITH the
   reference's loop structure (oracle/ref_numpy.draw_counts_as_reference; within 1 % of the imported
   reference's time, tools/cpu_port_vs_reference.py) on a bounded sample of the same workload,
   1 core, as the reference is single-threaded; ``cpu_baseline_all_cores``: the same on one process
   per physical core over disjoint cell ranges.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(n_cells, G, rows):
    """4 B/count written once + the mean tensor read once + per-gene and per-cell vectors
    (SURVEY.md section 8 d)."""
    return 4 * n_cells * G + 4 * rows * G + 8 * G + 8 * n_cells


def kernel_source_sha():
    """Fingerprint of the kernel sources (what a profile is a profile OF): the code of the device translation
    unit and its Makefile with comments and blank space taken out, so that an edited comment does not orphan a profile."""
    import hashlib
    import re
    h = hashlib.sha256()
    for fn in ("k3_stream.h", "k3_heavy.h", "prnb_device.h", "numpy_stream.h", "prosstt_amd.hip"):
        text = open(os.path.join(ROOT, "prosstt_amd", "csrc", fn), "r").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
        text = re.sub(r"//[^\n]*", " ", text)                       # line comments (no string of these sources holds //)
        h.update(" ".join(text.split()).encode())
    # and the flags they are compiled with (the Makefile without its comments)
    mk = open(os.path.join(ROOT, "prosstt_amd", "csrc", "Makefile"), "r").read()
    h.update(" ".join(re.sub(r"#[^\n]*", " ", mk).split()).encode())
    return h.hexdigest()[:16]


def profiled_traffic(suffix=""):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS
    command (profiles/rNN_summary.txt, written by tools/profile_bench.sh: separate --pmc runs for
    FETCH_SIZE and WRITE_SIZE).  Units and corrections as MI355X_MICROARCH.md prescribes: both
    counters are KiB; WRITE_SIZE is exact for 16-B/lane stores, FETCH_SIZE reads half of a wide
    coalesced stream on gfx950 and is doubled.  bench.py cannot run the profiler on itself, so
    this is the last profiled value, not a live one: (None, why) when no profile is committed or
    the committed one was taken on other kernel sources."""
    import glob
    import re
    # profiles/rNN_summary.txt: the default run (the headline workload); profiles/rNN_t32_summary.txt (suffix "_t32"): T32's
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_summary.txt"))
                   if re.fullmatch(r"r\d+%s_summary\.txt" % re.escape(suffix), os.path.basename(f)))
    if not files:
        return None, "no profile committed"
    text = open(files[-1]).read()
    name = os.path.relpath(files[-1], ROOT)
    m = re.search(r"kernel_source_sha:\s*([0-9a-f]+)", text)
    if not m or m.group(1) != kernel_source_sha():
        return None, "%s was taken on other kernel sources" % name
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        m = re.search(r"sample_counts_stream_kernel<true(?:, true)?>\s+%s\s+([0-9.e+]+)" % counter, text)
        if not m:
            return None, "%s lacks %s" % (name, counter)
        vals[counter] = float(m.group(1)) * 1024.0
    return vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"], name


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); half the
    logical CPUs when that cannot be read."""
    try:
        pairs, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((pkg, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _ref_tree(tree):
    from oracle import ref_numpy
    ref = ref_numpy.RefTree(tree.topology, {b: int(tree.time[b]) for b in tree.branches},
                            modules=tree.modules, G=tree.G)
    ref.means = tree.means                      # host float64 view of the device tensor
    return ref


def cpu_baseline(work, pt, br, sc, cells):
    """Reference-equivalent CPU path: oracle/ref_numpy.draw_counts_as_reference -- the reference's
    simulation.draw_counts with its own loop structure and library calls (bit-identical output
    at equal seed and within 1 % of the imported reference's wall time in the build container:
    tools/cpu_port_vs_reference.py) -- on the first `cells` cells, 1 core."""
    from oracle import ref_numpy
    ref = _ref_tree(work.tree)
    np.random.seed(12345)
    # chunks of 500 cells keep the reference's ~77 B per cell x gene of temporaries bounded
    done, t0 = 0, time.perf_counter()
    for lo in range(0, cells, 500):
        hi = min(lo + 500, cells)
        x = ref_numpy.draw_counts_as_reference(ref, pt[lo:hi], list(br[lo:hi]), sc[lo:hi], work.alpha, work.beta)
        done += x.size
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="cells*genes/s", cores=1, kind="port",
                sample="oracle/ref_numpy.draw_counts_as_reference (simulation.py:602-651 with the reference's "
                       "per-cell branch_times()/get_pr_umi loops and scipy.stats.nbinom(...).rvs()) on the first "
                       "%d cells x %d genes of the same plan, %.1f s" % (cells, work.tree.G, dt),
                host_cpus=os.cpu_count())


_WORKER = """
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import ref_numpy
d = np.load(sys.argv[2], allow_pickle=True)
lo, hi = int(sys.argv[3]), int(sys.argv[4])
branches = [str(b) for b in d["branches"]]
ref = ref_numpy.RefTree([[str(a), str(b)] for a, b in d["topology"]], {b: int(t) for b, t in zip(branches, d["times"])},
                        modules=int(d["modules"]), G=int(d["G"]))
ref.means = {b: d["means_" + b] for b in branches}
np.random.seed(1000 + lo)
for a in range(lo, hi, 250):
    b = min(a + 250, hi)
    ref_numpy.draw_counts_as_reference(ref, d["pt"][a:b], list(d["br"][a:b]), d["sc"][a:b], d["alpha"], d["beta"])
"""


def cpu_baseline_all_cores(work, pt, br, sc, cells, procs):
    """The same port on `procs` host processes over disjoint cell ranges (variant 2):
    child processes that never touch the GPU; the wall time of the slowest one counts."""
    import subprocess
    import tempfile
    tree = work.tree
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "plan.npz")
        means = tree.means
        np.savez(path, topology=np.array([[str(a), str(b)] for a, b in tree.topology]), branches=np.array([str(b) for b in tree.branches]),
                 times=np.array([int(tree.time[b]) for b in tree.branches]), modules=tree.modules, G=tree.G,
                 pt=pt[:cells], br=np.array([str(b) for b in br[:cells]]), sc=sc[:cells], alpha=work.alpha,
                 beta=work.beta, **{"means_" + str(b): means[b] for b in tree.branches})
        edges = np.linspace(0, cells, procs + 1).astype(int)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        t0 = time.perf_counter()
        kids = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, path, str(lo), str(hi)], env=env)
                for lo, hi in zip(edges[:-1], edges[1:]) if hi > lo]
        rcs = [k.wait() for k in kids]
        dt = time.perf_counter() - t0
    if any(rcs):
        raise RuntimeError("a CPU baseline worker failed: %r" % (rcs,))
    return dict(value=cells * tree.G / dt, unit="cells*genes/s", cores=len(kids), kind="port",
                sample="oracle/ref_numpy.draw_counts_as_reference on %d processes (one per physical core), %d cells x "
                       "%d genes in all, %.1f s wall (process start-up and plan loading included)"
                       % (len(kids), cells, tree.G, dt),
                host_cpus=os.cpu_count())


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(gpus, argv, script=None):
    """``python bench.py --gpus N`` outside a launcher: start the N ranks as a CHILD process
    (``python -m torch.distributed.run``; nothing in this parent has touched the GPU, and it never
    replaces itself), let rank 0's JSON line through on the inherited stdout, return the child's
    exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    rccl_debug_to_file(env)
    return subprocess.run(cmd, env=env).returncode


LINE_OUT = sys.stdout


def keep_stdout_for_the_line():
    """Only the contract's JSON line may reach stdout, and libraries print there too (RCCL's version banner under
    NCCL_DEBUG, the ROCm runtime's notices): the line goes out through a private copy of the descriptor, and
    descriptor 1 itself is pointed at stderr for everything else in this process."""
    global LINE_OUT
    sys.stdout.flush()
    LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def rccl_debug_to_file(env):
    """PROSSTT_BENCH_RCCL_DEBUG=1: RCCL says why it failed (NCCL_DEBUG=WARN, into a FILE).  Off by default: with
    NCCL_DEBUG set, RCCL also prints a version banner when the process group goes down -- behind the contract's JSON
    line, on whatever the caller reads."""
    if env.get("PROSSTT_BENCH_RCCL_DEBUG") != "1":
        return
    env.setdefault("NCCL_DEBUG", "WARN")
    env.setdefault("NCCL_DEBUG_FILE", os.path.join(os.environ.get("TMPDIR", "/tmp"), "prosstt_bench_rccl.%h.%p.log"))


def rccl_debug_tail(limit=1500):
    import glob
    pattern = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", "*").replace("%p", "*")
    text = ""
    for f in sorted(glob.glob(pattern))[-8:] if pattern else []:
        try:
            text += "[%s] %s\n" % (os.path.basename(f), open(f).read()[-400:].strip())
        except OSError:
            pass
    return text[-limit:]


class Job:
    """What every case of one bench.py process shares: ranks, device context, fences."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # functional test of the N > 1 path on a 1-GPU box: PROSSTT_BENCH_BACKEND=gloo PROSSTT_BENCH_ONE_GPU=1
        self.backend = os.environ.get("PROSSTT_BENCH_BACKEND", "nccl")
        if os.environ.get("PROSSTT_BENCH_ONE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        self.use_dist = self.world > 1 or os.environ.get("PROSSTT_BENCH_FORCE_DIST") == "1"   # the latter: RCCL path on 1 GPU
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            rccl_debug_to_file(os.environ)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            try:
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                else:
                    dist.init_process_group(self.backend)
            except Exception:
                sys.stderr.write("process group initialisation failed; RCCL's own log (PROSSTT_BENCH_RCCL_DEBUG=1 turns it on):\n%s\n"
                                 % rccl_debug_tail())
                raise
        from prosstt_amd import device
        self.ctx = device.get_context(local)
        self.red_dev = self.ctx.torch_device if self.backend == "nccl" else torch.device("cpu")

    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.red_dev)
        if self.use_dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.destroy_process_group()


def run_case(job, config, scaling, cells_per_gpu, steps, warmup, strict_steps, gather, ramp_ms=0.0):
    """One workload on the job's ranks: W untimed + K timed passes of the count sampler over each rank's
    shard (barrier + synchronize on both sides, max over ranks), optionally the row gather to rank 0."""
    from prosstt_amd import parallel, workloads
    torch, ctx, world, rank = job.torch, job.ctx, job.world, job.rank
    work = workloads.build(config)          # the tree goes through the product's lineage stage (not part of the metric)
    tree, G = work.tree, work.tree.G
    if scaling == "strong":
        n_total = cells_per_gpu * world if cells_per_gpu else work.cfg["N"]
        per_gpu = n_total // world
    else:
        per_gpu = cells_per_gpu or work.cfg["N"]
        n_total = per_gpu * world
    pt, br, sc, rows = work.plan(n_total)
    if tree._branch_owner is not None:      # the lineage was built sharded: a cell is sampled where its branch's rows are
        mine = parallel.cells_of_rank(br, tree._branch_owner, rank)
    else:
        mine, _ = parallel.shard_cells(br, rank, world)

    means = tree.device_means()
    # The step is the device part of the call simulation.draw_counts makes: the rank's cells PRESENTED grouped by their row
    # of the mean tensor (a host-side counting sort of the plan, prosstt_amd_plan_order: part of planning, like the row index
    # itself), every count keyed by the cell's position in the global plan (cell_index) -- so the matrix in HBM holds the counts
    # of an unordered call, bit for bit, its rows in the order of presentation; draw_counts puts them back in plan order inside
    # its copy to the host (end_to_end_ms includes that).  --plain-order presents the cells as planned (ms_per_step_plan_order).
    from prosstt_amd import device as _dev
    mine = np.asarray(mine, dtype=np.int64)
    plan_order_ms = None
    if not job.args.plain_order:
        t_po = time.perf_counter()
        perm = _dev.plan_order(rows[mine], means.shape[0])
        plan_order_ms = (time.perf_counter() - t_po) * 1e3      # host-side counting sort of the plan: once per plan, outside the step
        mine = mine[perm]
    d_rows = ctx.tensor(rows[mine], torch.int32)
    d_sc = ctx.tensor(sc[mine], torch.float64)
    d_al = ctx.tensor(work.alpha, torch.float64)
    d_be = ctx.tensor(work.beta, torch.float64)
    d_idx = ctx.tensor(mine, torch.int64)
    out = torch.empty((len(mine), G), dtype=torch.int32, device=ctx.torch_device)

    token = tree.means_token()

    def step(seed, timed=False, strict=True):
        # strict: the product API's default -- the reference's argument check (scipy behind simulation.py:647-648), riding in
        # the call's own kernels, its verdict read once behind the timed steps (simulation.draw_counts reads it behind the
        # copy to the host); the per-row flags of the mean tensor are kept while the tensor is unchanged
        ctx.sample_counts(means, d_rows, d_sc, d_al, d_be, seed=seed, out=out, cell_index=d_idx,
                          check_domain="deferred" if strict else False, time_kernel=timed, means_token=token)

    # The first steps of a process, before anything has loaded the device: idle clocks, cold code (reported as
    # ms_per_step_cold; a user's first calls see this, not the steady state below).
    job.fence()
    t0 = time.perf_counter()
    for i in range(5):
        step(3000 + i)
    ctx.domain_status()
    job.fence()
    ms_cold = job.max_over_ranks(time.perf_counter() - t0) / 5 * 1e3
    # The device comes out of the (mostly host-side) setup at idle clocks and takes a few hundred ms of load to
    # reach the clock it then holds (measured: the same kernel 1.88 ms in the first 3 steps, 1.63 ms after 60);
    # the W warmup steps of the contract are too short for that, so untimed passes of the same step run first.
    ramp_calls = 0
    if ramp_ms > 0:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:
            for i in range(8):
                step(2000 + ramp_calls + i)
            torch.cuda.synchronize()
            ramp_calls += 8
    for i in range(warmup):
        step(1000 + i)
    ctx.domain_status()
    job.fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, timed=True)                    # HIP events bracket K3 on the launch stream
    ctx.domain_status()                        # the verdict of the K checked steps (synchronises; inside the timed region)
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    kms = job.max_over_ranks(ctx.last_kernel_ms())      # mean over the K launches of the timed region
    res = dict(work=work, plan=(pt, br, sc), G=G, n_total=n_total, per_gpu=per_gpu, cells_on_rank=int(len(mine)),
               ms_per_step=elapsed / steps * 1e3, value=n_total * G / (elapsed / steps), kernel_ms=kms,
               rows_total=work.info["resident_rows"], ms_unchecked=None, ms_cold=ms_cold, gather_ms=None,
               ramp_calls=ramp_calls, plan_order_ms=plan_order_ms)

    # the same steps without the domain check (strict=False)
    if strict_steps > 0:
        step(99, strict=False)
        job.fence()
        t0 = time.perf_counter()
        for i in range(strict_steps):
            step(i, strict=False)
        job.fence()
        res["ms_unchecked"] = job.max_over_ranks(time.perf_counter() - t0) / strict_steps * 1e3

    # the same steps with the cells presented in the order of the plan (what a caller who does not group them gets)
    res["ms_plan_order"] = None
    if strict_steps > 0 and not job.args.plain_order:
        plain = np.sort(mine)
        p_rows, p_sc, p_idx = ctx.tensor(rows[plain], torch.int32), ctx.tensor(sc[plain], torch.float64), ctx.tensor(plain, torch.int64)

        def plain_step(seed):
            ctx.sample_counts(means, p_rows, p_sc, d_al, d_be, seed=seed, out=out, cell_index=p_idx, check_domain="deferred",
                              means_token=token)
        plain_step(98)
        job.fence()
        t0 = time.perf_counter()
        for i in range(strict_steps):
            plain_step(i)
        ctx.domain_status()
        job.fence()
        res["ms_plan_order"] = job.max_over_ranks(time.perf_counter() - t0) / strict_steps * 1e3
        step(97)                      # (leaves `out` as the timed steps wrote it: the sanity check below goes by d_rows)

    # sanity inside the bench: first moment of this rank's shard (catches a silently dead kernel)
    mu_sum = float((means.double().sum(dim=1)[d_rows.long()] * d_sc).sum())
    x_sum = sum(int(out[lo:lo + 16384].sum(dtype=torch.int64)) for lo in range(0, len(mine), 16384))   # bounded temporaries
    res["ratio"] = x_sum / mu_sum if mu_sum > 0 else float("nan")
    res["tree"] = tree
    if gather and world > 1:
        time_gather(job, res, out, mine)
    else:
        res["shard"] = (out, mine)          # for a later time_gather (the main case: after the line is assembled)
    return res


def hbm_achievable(job, n_cells, G, reps=10):
    """What this device's HBM delivers to the simplest kernels, measured in this process beside the vendor peak that
    ``roofline.frac`` is priced against (SURVEY section 8 d: "use the vendor peak as denominator; also report the
    achievable"): a fill of an (N, G) int32 matrix -- write-only, like the sampler's traffic -- and a device-to-device
    copy of one (bytes read + bytes written).  torch's own kernels: a yardstick, not part of the path."""
    torch = job.torch
    dev = job.ctx.torch_device
    a = torch.empty((n_cells, G), dtype=torch.int32, device=dev)
    b = torch.empty((n_cells, G), dtype=torch.int32, device=dev)
    nbytes = 4.0 * n_cells * G

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    t_fill = timed(lambda: a.fill_(3))
    t_copy = timed(lambda: b.copy_(a))
    del a, b
    return {"fill_GBps": nbytes / t_fill / 1e9, "copy_GBps": 2.0 * nbytes / t_copy / 1e9, "matrix_bytes": nbytes,
            "note": "torch's fill_ (write-only) and copy_ (read + write) of an int32 matrix of the workload's size, %d repeats each; "
                    "the sampler writes every count once, so fill_GBps is the rate a kernel with no arithmetic at all reaches on "
                    "this traffic" % reps}


def time_gather(job, res, out, mine):
    """The one exchange of the path: count rows to rank 0 (point-to-point over xGMI), timed on its own.
    Bounded so that root's copy of the whole matrix plus its own shard stays far inside 288 GB."""
    from prosstt_amd import parallel
    n_total, G = res["n_total"], res["G"]
    to_host = job.args.gather_to == "host"
    if job.world == 1 or (4 * n_total * G > 64e9 and not to_host):
        return
    small = min(64, len(mine))
    parallel.gather_rows(out[:small], mine[:small], n_total)        # connections come up outside the timed region
    job.fence()
    if job.backend == "nccl":
        t0 = time.perf_counter()
        full = parallel.gather_rows(out, mine, n_total, to_host=to_host)
        job.fence()
        res["gather_ms"] = job.max_over_ranks(time.perf_counter() - t0) * 1e3
        res["gather_to"] = job.args.gather_to
        del full
    else:
        # functional run on another backend (gloo stages device tensors through the host at ~25 MB/s):
        # the exchange is exercised on 2048 rows per rank and not timed
        part = min(2048, len(mine))
        full = parallel.gather_rows(out[:part], mine[:part], n_total)
        job.fence()
        res["gather_rows_functional"] = int(job.max_over_ranks(part))
        del full


def time_pipeline(job, res):
    """Sampling AND the exchange as one pipeline (parallel.sample_and_gather: chunk c travels to rank 0 while chunk c + 1 is
    sampled, received straight into final rows): whole-job cells x genes per second WITH the gather -- what a user who wants
    the matrix on one GPU gets (SURVEY section 8 e).  Bounded like time_gather; timed on RCCL only."""
    from prosstt_amd import parallel
    n_total, G, tree, work = res["n_total"], res["G"], res["tree"], res["work"]
    if job.world == 1 or 4 * n_total * G > 64e9:
        return
    seed_of = work.cfg["seed"] + 1
    np.random.seed(seed_of)
    if job.backend != "nccl":
        parallel.sample_and_gather(tree, 1024 * job.world, alpha=work.alpha, beta=work.beta, seed=5, chunk_cells=256)
        job.fence()
        res["pipeline_functional"] = 1024 * job.world
        return
    parallel.sample_and_gather(tree, 4096, alpha=work.alpha, beta=work.beta, seed=5)       # connections, code paths
    job.fence()
    np.random.seed(seed_of)
    t0 = time.perf_counter()
    full, _, _, _, _ = parallel.sample_and_gather(tree, n_total, alpha=work.alpha, beta=work.beta, seed=5)
    job.fence()
    dt = job.max_over_ranks(time.perf_counter() - t0)
    res["pipeline_ms"] = dt * 1e3
    res["value_with_gather"] = n_total * G / dt
    del full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "T32", "C4", "C5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --cells-per-gpu (default: the config's cell count) on every GPU; "
                         "strong: the config's cell count in all, split over the GPUs")
    ap.add_argument("--cells-per-gpu", type=int, default=None)
    ap.add_argument("--cpu-cells", type=int, default=10000, help="cells timed by the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="host processes of the all-cores CPU baseline (-1: one per physical core, 0: skip)")
    ap.add_argument("--strict-steps", type=int, default=5, help="steps also timed WITHOUT the domain check (0 = skip)")
    ap.add_argument("--strong-configs", default="C4,C5",
                    help="with N > 1: configurations also run at their own cell count split over the GPUs "
                         "(reported under 'strong_scaling'; '' = none)")
    ap.add_argument("--no-gather", action="store_true", help="do not time the row gather to rank 0 (N > 1)")
    ap.add_argument("--gather-to", default="device", choices=["device", "host"],
                    help="where rank 0 assembles the gathered matrix: its device (default; skipped beyond 64 GB) or "
                         "page-locked host memory (any size: C5's 120 GB of counts)")
    ap.add_argument("--ramp-ms", type=float, default=400.0,
                    help="untimed passes of the step for this long before the W warmup steps (device clock ramp; 0 = none)")
    ap.add_argument("--fail-on-extras-error", action="store_true",
                    help="N > 1: exit with code 3 when the gather or a strong-scaling case failed or stalled "
                         "(the line still goes out with 'extras_error'; default: exit code 0)")
    ap.add_argument("--no-target-shape", action="store_true",
                    help="skip the extra case on north_star's own target shape (T32: 32-branch tree, 50k x 20k; N = 1, default C3 run)")
    ap.add_argument("--plain-order", action="store_true",
                    help="present the cells in the order of the plan instead of grouped by mean-tensor row (what a caller who "
                         "does not order its cells gets; simulation.draw_counts groups them)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end-to-end time of the drop-in sample_density call (N = 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # before anything here has touched torch.cuda
    keep_stdout_for_the_line()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        sys.exit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world_env))

    job = Job(args)
    world, rank = job.world, job.rank
    main_case = run_case(job, args.config, args.scaling, args.cells_per_gpu, args.steps, args.warmup,
                         args.strict_steps, gather=False, ramp_ms=args.ramp_ms)
    work, G, n_total = main_case["work"], main_case["G"], main_case["n_total"]
    pt, br, sc = main_case["plan"]

    # BASELINE.json's north_star quotes its target ("at >= 40 % HBM roofline on 1 MI355X") on a 32-branch tree at the
    # headline's size; the metric's own configuration (configs[2], the headline above) is the 8-branch tree.  The default
    # 1-GPU run therefore carries that shape too, under "north_star_shape" (the same step, the same timing rules).
    target_case = None
    default_shape = args.config == "C3" and args.cells_per_gpu is None and args.scaling == "weak" and world == 1
    if default_shape and not args.no_target_shape:
        target_case = run_case(job, "T32", "weak", None, args.steps, args.warmup, 0, gather=False, ramp_ms=min(args.ramp_ms, 150.0))
        target_case.pop("shard", None)
    main_case["target_case"] = target_case

    end_to_end = None
    main_case["achievable"] = None
    if world == 1:
        main_case.pop("shard", None)
        if 8.0 * main_case["cells_on_rank"] * G < 0.4 * job.torch.cuda.get_device_properties(job.ctx.device).total_memory:
            main_case["achievable"] = hbm_achievable(job, main_case["cells_on_rank"], G)
    if world == 1 and not args.no_end_to_end and rank == 0:
        end_to_end = tuple(end_to_end_ms(main_case["tree"], work, n_total, out) for out in ("numpy", "numpy32", "csr"))

    # With N > 1 the measurements beyond the contract's line (the row gather, the strong-scaling configurations)
    # run under a watchdog: if one of them raises or stalls, rank 0 still prints the line -- with what was
    # measured and the reason under "extras_error" -- and every rank leaves with exit code 0.
    strong, extras_error = [], None

    def emit(why=None):
        """Print the line (rank 0).  ``why``: the watchdog's reason when it is the caller."""
        if rank != 0:
            return
        line = assemble_line(args, job, main_case, strong, end_to_end, why or extras_error)
        LINE_OUT.write(json.dumps(line) + "\n")
        LINE_OUT.flush()

    guard = ExtrasGuard(rank, float(os.environ.get("PROSSTT_BENCH_EXTRAS_TIMEOUT_S", "600")),
                        exit_code=3 if args.fail_on_extras_error else 0)
    if world > 1:
        guard.arm(lambda why: emit(why))
        try:
            if not args.no_gather:
                time_gather(job, main_case, *main_case["shard"])
            main_case.pop("shard", None)
            if not args.no_gather:
                time_pipeline(job, main_case)
            from prosstt_amd import workloads
            sharing = world if os.environ.get("PROSSTT_BENCH_ONE_GPU") == "1" else 1    # ranks on one device (functional test)
            for cfg in [c for c in args.strong_configs.split(",") if c]:
                spec = workloads.CONFIGS[cfg]
                per_device = 4.0 * spec["N"] * spec["G"] / world * sharing
                if per_device > 0.35 * job.torch.cuda.get_device_properties(job.ctx.device).total_memory:
                    strong.append({"config": cfg, "skipped": "%.0f GB of counts per device" % (per_device / 1e9)})
                    continue
                case = run_case(job, cfg, "strong", None, max(3, args.steps // 2), 2, 0, gather=not args.no_gather,
                                ramp_ms=args.ramp_ms / 2)
                strong.append({"config": cfg, "cells_total": case["n_total"], "genes": case["G"],
                               "cells_on_rank_0": case["cells_on_rank"], "value": case["value"], "unit": "cells*genes/s",
                               "ms_per_step": case["ms_per_step"], "kernel_ms_max_over_ranks": case["kernel_ms"],
                               "gather_ms": case["gather_ms"], "lineage_s": round(case["work"].info["lineage_s"], 3),
                               "lineage_attempts": case["work"].info["attempts"],
                               "sum_counts_over_sum_means": round(case["ratio"], 5)})
                del case
        except Exception as exc:          # noqa: BLE001 -- whatever it is, the line must still go out
            extras_error = "%s: %s" % (type(exc).__name__, str(exc)[:300])
            tail = rccl_debug_tail(400)
            if tail:
                extras_error += " | RCCL: " + tail.replace("\n", " ")
    main_case.pop("shard", None)

    if not guard.disarm():
        import threading
        threading.Event().wait()    # the watchdog is printing and will end the process
    emit()
    if extras_error is not None:
        # the other ranks may be waiting in a collective this rank left: no orderly shutdown is possible.  The line has
        # gone out with the reason; --fail-on-extras-error turns the reason into exit code 3 for callers that go by the
        # exit status (the default keeps 0: the contract's line itself was measured)
        sys.stdout.flush()
        os._exit(3 if args.fail_on_extras_error else 0)
    job.close()


class ExtrasGuard:
    """Deadline for the measurements that are not part of the contract's line (N > 1)."""

    def __init__(self, rank, seconds, exit_code=0):
        import threading
        self.rank, self.seconds, self.lock, self.state, self.timer = rank, seconds, threading.Lock(), "idle", None
        self.exit_code = exit_code

    def arm(self, emit):
        import threading

        def fire():
            with self.lock:
                if self.state != "armed":
                    return
                self.state = "fired"
            try:
                emit("extras timed out after %.0f s (PROSSTT_BENCH_EXTRAS_TIMEOUT_S); what was measured until then is reported" % self.seconds)
            finally:
                sys.stdout.flush()
                os._exit(self.exit_code)

        self.state = "armed"
        self.timer = threading.Timer(self.seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        """False if the deadline has already fired (the caller must not print)."""
        with self.lock:
            if self.state == "fired":
                return False
            self.state = "done"
        if self.timer is not None:
            self.timer.cancel()
        return True


def assemble_line(args, job, main_case, strong, end_to_end, extras_error):
    """The one JSON line of the contract, from what has been measured (rank 0)."""
    world = job.world
    work, G, n_total = main_case["work"], main_case["G"], main_case["n_total"]
    pt, br, sc = main_case["plan"]
    kms = main_case["kernel_ms"]
    abytes = algorithmic_bytes(main_case["cells_on_rank"], G, main_case["rows_total"])
    achieved = abytes / (kms * 1e-3)
    ms_per_step = main_case["ms_per_step"]
    default_shape = args.config == "C3" and args.cells_per_gpu is None and args.scaling == "weak" and world == 1
    t32_shape = args.config == "T32" and args.cells_per_gpu is None and args.scaling == "weak" and world == 1
    traffic, traffic_src = profiled_traffic() if default_shape else (
        profiled_traffic("_t32") if t32_shape else (None, "profiles are of the default 1-GPU C3 run and of --config T32"))
    line = {
        "metric": "simulated cells*genes per second (count sampling: sample_density -> draw_counts)",
        "value": main_case["value"], "unit": "cells*genes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_step_strict": ms_per_step,
        "ms_per_step_unchecked": main_case["ms_unchecked"], "ms_per_step_plan_order": main_case.get("ms_plan_order"),
        "ms_per_step_cold": main_case["ms_cold"],
        "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d-branch tree (T=50, K=25), %d genes, %d cells per GPU "
                               "(%d total), density sampling; lineage via the product pipeline"
                               % (args.config, work.info["branches"], G, main_case["per_gpu"], n_total),
                   "cells_on_rank_0": main_case["cells_on_rank"],
                   "parallelism": "cells sharded by branch, %d rank(s), no data-path collective" % world,
                   "lineage_attempts": work.info["attempts"], "lineage_s": round(work.info["lineage_s"], 3),
                   "lineage_sharded_by_genes": bool(work.info["sharded"]),
                   "clock_ramp": "%d untimed passes (%.0f ms) before the %d warmup steps" % (main_case["ramp_calls"], args.ramp_ms, args.warmup),
                   "cold_steps": 5,
                   "presentation": "plan order" if args.plain_order else
                                   "cells grouped by mean-tensor row (simulation.draw_counts' call: prosstt_amd_plan_order); rows of "
                                   "the matrix in that order, put back in plan order inside the copy to the host",
                   "api_call": ("the step is the device part of prosstt_amd.simulation.draw_counts(tree, pseudotime, branches, scalings, "
                                "alpha, beta, out='torch'%s) -- the same prosstt_amd_sample_counts call on the same arrays; its return is "
                                "the matrix as timed (device.PresentedCounts: counts + cell_of_row); the host-side grouping of the plan "
                                "(plan_order_host_ms, once per plan) and the row-index computation are outside the step"
                                % (", order='plan'" if args.plain_order else "")),
                   "plan_order_host_ms": main_case.get("plan_order_ms"),
                   "library_version": _library_version(), "sampler_definition": "PRNB-7",
                   "gcn_arch": job.torch.cuda.get_device_properties(job.ctx.device).gcnArchName,
                   "sum_counts_over_sum_means": round(main_case["ratio"], 5)},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k3::sample_counts_stream_kernel<%s, %s>" % (
                         "true" if G % 4 == 0 else "false",
                         "true" if 4.0 * main_case["cells_on_rank"] * G >= 1073741824.0 and main_case["cells_on_rank"] * ((G + 255) // 256) >= 64 * 4 * 5 * 1024 else "false"),
                     "kernel_ms": kms,
                     "frac_whole_step": abytes / (ms_per_step * 1e-3) / HBM_PEAK,
                     "algorithmic_bytes_per_launch": abytes, "kernel_source_sha": kernel_source_sha(),
                     "achievable": main_case.get("achievable"),
                     "note": "frac = algorithmic bytes / the dominant kernel's mean duration (HIP events, max over "
                             "ranks); frac_whole_step prices them against ms_per_step (K3h, prep kernels, gaps). "
                             "VALU-issue-bound sampler: DESIGN.md section 6 and profiles/"},
    }
    tc = main_case.get("target_case")
    if tc is not None:
        tb = algorithmic_bytes(tc["cells_on_rank"], tc["G"], tc["rows_total"])
        line["north_star_shape"] = {
            "workload": "T32: %d-branch tree (T=50, K=25), %d genes, %d cells, 1 GPU -- north_star's target shape (C4's tree, "
                        "the headline's size)" % (tc["work"].info["branches"], tc["G"], tc["n_total"]),
            "value": tc["value"], "unit": "cells*genes/s", "ms_per_step": tc["ms_per_step"], "kernel_ms": tc["kernel_ms"],
            "frac": tb / (tc["kernel_ms"] * 1e-3) / HBM_PEAK, "frac_whole_step": tb / (tc["ms_per_step"] * 1e-3) / HBM_PEAK,
            "algorithmic_bytes_per_launch": tb, "steps": args.steps,
            "traffic": profiled_traffic("_t32")[0], "traffic_source": profiled_traffic("_t32")[1],
            "sum_counts_over_sum_means": round(tc["ratio"], 5), "lineage_attempts": tc["work"].info["attempts"]}
    if world > 1:
        line["scaling_note"] = ("value is the WEAK figure (the C3 cell count on every GPU): value(N) / value(1) is the "
                                "weak-scaling factor; BASELINE.json's '>= 6x at 8 GPUs' for its 32-branch configuration "
                                "is strong_scaling[config C4].value / the 1-GPU C4 rate (python bench.py --config C4), "
                                "sampling only -- gather_ms is the one exchange on top")
    if main_case["gather_ms"] is not None:
        line["gather_ms"] = main_case["gather_ms"]
        line["gather_to"] = main_case.get("gather_to", "device")
    if "gather_rows_functional" in main_case:
        line["gather_ms"] = None
        line["gather_note"] = "backend %s: gather exercised on %d rows per rank, not timed" % (job.backend, main_case["gather_rows_functional"])
    if main_case.get("pipeline_ms") is not None:
        line["value_with_gather"] = main_case["value_with_gather"]
        line["pipeline_ms"] = main_case["pipeline_ms"]
        line["pipeline_note"] = ("parallel.sample_and_gather: host plan + sampling + the exchange to rank 0 as one pipeline "
                                 "(chunks of 256 MB travel while the next is sampled; rows land in their final place, the "
                                 "permutation is returned); wall time incl. the host-side plan, max over ranks")
    if "pipeline_functional" in main_case:
        line["pipeline_note"] = "backend %s: sample_and_gather exercised on %d cells, not timed" % (job.backend, main_case["pipeline_functional"])
    if strong:
        line["strong_scaling"] = strong
    if end_to_end is not None:
        line["end_to_end_ms"], line["end_to_end_ms_int32"], line["end_to_end_ms_csr"] = end_to_end
    if world == 1 and args.cpu_cells > 0:
        line["cpu_baseline"] = cpu_baseline(work, pt, br, sc, min(args.cpu_cells, n_total))
        line["speedup_vs_cpu_1core"] = main_case["value"] / line["cpu_baseline"]["value"]
    procs = physical_cores() if args.cpu_procs < 0 else args.cpu_procs
    if world == 1 and args.cpu_cells > 0 and procs > 1:
        line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(
            work, pt, br, sc, min(max(args.cpu_cells, 250 * procs), n_total), procs)
    if extras_error:
        line["extras_error"] = extras_error
    return line


def _library_version():
    from prosstt_amd import _native
    return int(_native.load().prosstt_amd_version())


def end_to_end_ms(tree, work, n_cells, out="numpy"):
    """Wall time of the drop-in call a reference user makes -- ``simulation.sample_density`` returning the
    (N, G) matrix on the host: host plan, kernels, domain check, and the device-to-host copy (PCIe-inclusive;
    never ``value``).  out="numpy": the reference's int64 ndarray; "numpy32": int32 (both: 1 byte per count over PCIe -- the low 8 bits, the
    larger counts beside them -- widened by host threads under the transfer: device.to_host; until round 5 int64 was formed on the
    device and 8 bytes per count crossed PCIe, int32 4); "csr": scipy.sparse.csr_matrix compacted on the device (8 bytes per
    non-zero over PCIe).  Third of three calls (the first sizes the pinned buffers)."""
    from prosstt_amd import simulation as sim
    best = None
    for _ in range(3):
        np.random.seed(work.cfg["seed"] + 1)
        t0 = time.perf_counter()
        x = sim.sample_density(tree, n_cells, alpha=work.alpha, beta=work.beta, out=out)[0]
        dt = (time.perf_counter() - t0) * 1e3
        assert x.shape == (n_cells, tree.G) and x.dtype == (np.int64 if out == "numpy" else np.int32)
        assert out != "csr" or x.format == "csr"
        del x
        best = dt
    return best


if __name__ == "__main__":
    main()
