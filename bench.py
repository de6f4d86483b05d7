#!/usr/bin/env python3
"""Benchmark of the hot path: component table resident in HBM -> solution x.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one pass of the hot path over one batch of synthetic circuits per GPU: for every
circuit symbolic + numeric assembly + solve, the equivalent of the reference's
`Circuit(netlist, sparse)` + `.solve()` (reference nodal/nodal.py:306-336).  Inputs are
uploaded before the timed region; nothing is cached between circuits or steps.

Workloads (BASELINE.json `configs`):
    cfg3  1000x1000 resistor grid (1e6 nodes), sparse CSR path        the headline at EVERY N
          (the configuration north_star's ">= 50x scipy.sparse" target is quoted on)
    cfg4  batch of 100x100 grids with per-member values, 128 members per GPU, solved as
          one block-diagonal system per rank and gathered over RCCL   (`also`, at every N)
    cfg2  100x100 resistor grid, dense G, fp64 block elimination / LU
    cfg5  1000x1000 grid + 1% E + CCCS/VCVS, sparse path

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank processes itself
(fresh children, before this process touches the GPU); under torchrun (WORLD_SIZE set) it
is a rank.  ONE workload at every N (round 5): each rank solves its own independent cfg3
circuits (north_star: "partition independent netlists ... across the 8 GPUs"), the solution of
every finished circuit is all-gathered over RCCL inside the timed region (8 MB per circuit and
rank, overlapped with the next solve: nodal_amd.batch.ShardedCircuits), `value` = circuits all
ranks solved / the slowest rank's time -- weak scaling in the number of circuits.  At N = 1
there is no process group: the line is BENCH's.  cfg4, the value sweep that shards as one
block system per rank, is measured in `also` at every N.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the workload,
timed with HIP events on the library's stream, plus the end-to-end figure of SURVEY.md
section 8d; `cpu_baseline` is the oracle (the reference's algorithm restated, same numpy /
scipy calls) timed on this box's host cores; `also` carries the other configurations, each
with its own cpu_baseline.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_MFMA_PEAK_TF = 78.6    # vendor dense fp64 matrix peak
# measured on this hardware with tools/mfma_f64_peak.hip / tools/fp64_mix.hip
# (profiles/r01_mfma_f64_peak.txt, profiles/r01_fp64_mix.txt):
FP64_MFMA_444_TF = 73.0     # v_mfma_f64_4x4x4_4b_f64 -- the instruction gemm_f64.hip issues
FP64_MFMA_INSTR_TF = 36.2   # v_mfma_f64_16x16x4_f64, 138 cycles/instruction/wave
FP64_VALU_FMA_TF = 59.3     # v_fma_f64

CIRCUITS_PER_STEP = {"cfg3": 32, "cfg5": 16, "cfg2": 8, "cfg4": 128}
PROFILE_ROUND = "r05"


def pmc_traffic(workload):
    """HBM bytes per launch of the workload's dominant kernel from the committed rocprofv3
    PMC summary (FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/<round>_pmc_traffic.json).
    None if that workload was not profiled."""
    for rnd in (PROFILE_ROUND, "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")) as f:
                return json.load(f)["dominant"][workload]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def pmc_by_class(workload):
    """HBM bytes per dispatch of every class of kernel, from counters: the committed rocprofv3 --pmc passes of
    `bench.py --workload W` summed per class by tools/pmc_by_class.py (profiles/<round>_pmc_by_class.json; the counters
    cannot be collected inside the timed run).  None when that workload was not profiled."""
    for rnd in (PROFILE_ROUND,):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_by_class.json")) as f:
                got = json.load(f)
            if got.get("workload") == workload:
                got["file"] = f"profiles/{rnd}_pmc_by_class.json"
                return got
        except Exception:
            continue
    return None


LIVE_CLASSES = {}  # workload -> classes measured by THIS run's kernel-trace child (live_classes)


def live_classes(workload, n, nnz):
    """Where the GPU time of one solve goes, by class of kernel, MEASURED IN THIS RUN: a child process --
    `rocprofv3 --kernel-trace -- python3 bench.py --workload W --steps 2 --warmup 1 ...`, started before this
    process has made any GPU call -- runs the same workload under the kernel tracer and tools/prof_classes.py
    sorts its dispatches (by kernel name and grid size) into level-0 passes with their algorithmic bytes,
    coarse levels, hierarchy setup, stamping.  None when rocprofv3 is not on PATH or the child fails."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    try:
        from tools import prof_classes
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--kernel-trace", "-d", d, "-o", "classes", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu", "--no-also", "--concurrent", "0",
                   "--no-classes"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return None
            res = prof_classes.classify(d, workload, n, nnz)[workload]
        res["source"] = ("live: rocprofv3 --kernel-trace child of this bench.py run (2 steps; tracing slows the "
                         "launch-bound classes by ~15 %, the shares are those of the traced run)")
        return res
    except Exception as e:  # noqa: BLE001 -- the headline must not depend on the profiler
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def profile_classes(workload):
    """This run's classes (live_classes) or, when the profiler child could not run, the committed ones of an
    earlier run (profiles/<round>_classes.json), marked as such."""
    live = LIVE_CLASSES.get(workload)
    if live and "by_class" in live:
        return live
    for rnd in (PROFILE_ROUND, "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_classes.json")) as f:
                got = json.load(f).get(workload)
            if got:
                got["source"] = f"COMMITTED profiles/{rnd}_classes.json (this run's profiler child did not run)"
                return got
        except Exception:
            continue
    return None


def alg_bytes(table, n, nnz):
    """B_asm = sum of record bytes + 12 nnz + 4(n+1) + 8n;
    B_solve,min = 12 nnz + 4(n+1) + 16n."""
    import numpy as np
    dependent = int(np.count_nonzero(table.type >= 3))
    records = 17 * (table.ncomp - dependent) + 29 * dependent
    b_asm = records + 12 * nnz + 4 * (n + 1) + 8 * n
    b_solve = 12 * nnz + 4 * (n + 1) + 16 * n
    return b_asm, b_solve


# ---------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------

class SingleCircuits:
    """cfg2 / cfg3 / cfg5: `per_step` independent circuits per step, one after the other on
    one handle; every circuit runs symbolic + numeric + solve."""

    def __init__(self, name, rank, per_step, device, dist=None, force_collective=False):
        import numpy as np
        from nodal_amd import _ffi
        from nodal_amd import generators as gen
        self.name, self.per_step = name, per_step
        self.dense = name == "cfg2"
        if name == "cfg2":
            self.table = gen.grid_table(100)
            self.desc = "grid(100) 1e4 nodes, dense G, fp64 block elimination"
        elif name == "cfg3":
            self.table = gen.grid_table(1000)
            self.desc = "grid(1000) 1e6 nodes, sparse CSR path"
        else:
            self.table = gen.cfg5_table(1000)
            self.desc = "grid(1000) + 1% E + CCCS/VCVS (non-symmetric MNA), sparse path"
        # N > 1 (or --force-collective): the rank's circuits through nodal_amd.batch.ShardedCircuits, which shares
        # every finished circuit's x with all ranks (all_gather over RCCL, overlapped with the next solve); at
        # N = 1 there is no group and ShardedCircuits.solve_next is the bare nodal_run below
        from nodal_amd.batch import ShardedCircuits
        self.sc = ShardedCircuits(self.table, dist, device, force_collective=force_collective, dense=self.dense)
        self.h = self.sc.h
        if self.sc.collective:
            self.desc += (f"; every circuit's x all_gathered over {'RCCL' if self.sc.backend == 'nccl' else self.sc.backend}"
                          f" ({self.table.n * 8 / 1e6:.1f} MB per rank and circuit, overlapped with the next solve)")
        self.h2d_first_ms = self.sc.upload_first_ms  # (device buffers allocated and zero-filled there)
        t0 = time.perf_counter()
        self.h.upload(self.table)
        self.h.synchronize()
        self.h2d_ms = (time.perf_counter() - t0) * 1e3
        self.reuse = False  # True: the symbolic phases (stamping lists, multigrid patterns) of the last circuit are kept
        self.phase = np.zeros(3)
        self.kern_ms = self.kern_n = 0
        self.kern_alg = 0.0
        self.circuits_done = 0

    def step(self):
        h = self.h
        for _ in range(self.per_step):
            info = self.sc.solve_next(reuse_symbolic=self.reuse)
            if info != 0:
                raise RuntimeError(f"solver reported info={info}")
            ms, launches, alg = h.kernel_stats()
            self.kern_ms += ms
            self.kern_n += launches
            self.kern_alg = alg
            self.phase += h.timings()
            self.circuits_done += 1
        self.sc.drain()  # (the last gathers of the step: inside the timed region)

    @property
    def gather_ms(self):
        return self.sc.gather_ms

    def reset_stats(self):
        self.phase[:] = 0
        self.kern_ms = self.kern_n = 0
        self.circuits_done = 0
        self.sc.gather_ms = 0.0

    def finish(self):
        import numpy as np
        h = self.h
        gathered_ok = None
        if self.sc.collective:  # every rank must hold every rank's last solution (identical circuits: identical x)
            g = self.sc.latest()
            gathered_ok = bool(g.shape[0] == self.sc.world and np.isfinite(g).all() and
                               all(np.array_equal(g[r], g[self.sc.rank]) for r in range(g.shape[0])))
        x = h.download_x()  # (the first one allocates its page-locked block: _ffi.host_empty recycles it from then on)
        del x
        t0 = time.perf_counter()
        x = h.download_x()
        d2h_ms = (time.perf_counter() - t0) * 1e3
        if gathered_ok:
            gathered_ok = bool(np.array_equal(self.sc.latest()[self.sc.rank], x))
        iterations, levels, _ = h.solve_info()
        out = dict(resid=h.residual(), x0=float(x[0]), n=h.n, nnz=h.nnz, iterations=iterations,
                   amg_levels=levels, d2h_ms=d2h_ms, h2d_ms=self.h2d_ms, h2d_first_ms=self.h2d_first_ms,
                   h2d_bytes=int(sum(np.asarray(getattr(self.table, f)).nbytes for f in
                                     (("type", "value", "a", "b") if self.table.B == 0 else
                                      ("type", "value", "a", "b", "c", "d", "drv", "k")))),
                   d2h_bytes=int(x.nbytes), gathered_ok=gathered_ok)
        self.sc.close()
        return out


class BatchShard:
    """cfg4: this rank's 128 members of the value sweep as one block-diagonal system
    (nodal_run_batch), results gathered from device memory over RCCL when world > 1 (or with
    --force-collective: the gather of a single rank, the RCCL path on a one-GPU box).  The
    step itself is nodal_amd.batch.ShardedBatch -- the entry solve_batch_distributed and the
    tests use; this class only adds the timers."""

    def __init__(self, rank, world, per_gpu, device, dist, force_collective=False):
        import numpy as np
        from nodal_amd import generators as gen
        from nodal_amd import _ffi
        from nodal_amd.batch import ShardedBatch
        self.name, self.per_step, self.dist, self.world = "cfg4", per_gpu, dist, world
        self.dense = False
        self.table = gen.grid_table(100)
        self.shard = ShardedBatch(self.table, per_gpu * world, dist, device, force_collective=force_collective)
        collective = self.shard.gathered is not None
        self.desc = (f"batch of {per_gpu * world} grid(100) value sweeps, {per_gpu} per GPU as one "
                     "block-diagonal system, sparse path" +
                     ((", all_gather over RCCL" if self.shard.backend == "nccl" else f", all_gather over {self.shard.backend}")
                      if collective else ""))
        vals = np.ones((per_gpu, self.table.ncomp))
        for i in range(per_gpu):
            vals[i, :-1] = gen.cfg4_values(rank * per_gpu + i, 100)
        h = self.shard.session.h
        h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)  # one handle per rank, one call at a time
        self.h2d_first_ms = None
        h.assemble_symbolic()  # per-member n, nnz for the byte counts (untimed)
        self.n, self.nnz = h.n, h.nnz
        t0 = time.perf_counter()
        self.shard.upload(vals)
        self.h2d_ms = (time.perf_counter() - t0) * 1e3
        self.h2d_bytes = int(vals.nbytes)
        self.phase = np.zeros(3)
        self.kern_ms = self.kern_n = 0
        self.kern_alg = 0.0
        self.circuits_done = 0
        self.reuse = False  # True: the block system's symbolic phases (one member's lists, replicated; the
        # hierarchy's patterns) are kept from step to step: a long value sweep on one topology

    @property
    def gather_ms(self):
        return self.shard.gather_ms

    def step(self):
        self.shard.step(reuse_symbolic=self.reuse)
        h = self.shard.session.h
        ms, launches, alg = h.kernel_stats()
        self.kern_ms += ms
        self.kern_n += launches
        self.kern_alg = alg
        self.phase += h.timings()
        self.circuits_done += self.per_step

    def reset_stats(self):
        self.phase[:] = 0
        self.kern_ms = self.kern_n = 0
        self.shard.gather_ms = 0.0
        self.circuits_done = 0

    def finish(self):
        import numpy as np
        h = self.shard.session.h
        t0 = time.perf_counter()
        x = self.shard.own_block()
        d2h_ms = (time.perf_counter() - t0) * 1e3
        iterations, levels, _ = h.solve_info()
        # every rank must hold every member after the gather
        gathered_ok = None
        if self.shard.gathered is not None:
            g = self.shard.result()
            lo, hi = self.shard.lo, self.shard.hi
            gathered_ok = bool(np.array_equal(g[lo:hi], x) and np.isfinite(g).all())
        out = dict(resid=h.residual(), x0=float(x[0, 0]), n=self.n, nnz=self.nnz,
                   iterations=iterations, amg_levels=levels, d2h_ms=d2h_ms, h2d_ms=self.h2d_ms,
                   h2d_bytes=self.h2d_bytes, d2h_bytes=int(x.nbytes), gathered_ok=gathered_ok)
        self.shard.close()
        return out


def concurrent_throughput(name, device, streams, per_stream, reuse=False):
    """Throughput with `streams` independent solves in flight (one handle, HIP stream and host thread
    each): a solve alternates bandwidth-bound passes over the fine level with latency-bound
    launches on the coarse levels, so independent circuits overlap well.  Reported beside the
    single-stream headline, whose per-kernel timings it would blur."""
    import threading
    from nodal_amd import _ffi
    from nodal_amd import generators as gen
    table = {"cfg3": lambda: gen.grid_table(1000), "cfg5": lambda: gen.cfg5_table(1000),
             "cfg2": lambda: gen.grid_table(100)}[name]()
    dense = name == "cfg2"
    handles = []
    # four and more solves in flight: main streams of default priority (a context's main stream is of the
    # highest priority, which serves one or two solves best -- 9.1 instead of 9.7 ms for one -- but the
    # device has few high-priority queues: 102 instead of 149 circuits/s with four)
    prio = os.environ.get("NODAL_STREAM_PRIORITY")
    if streams >= 4 and prio is None:
        os.environ["NODAL_STREAM_PRIORITY"] = "normal"
    for _ in range(streams):
        h = _ffi.Handle(device)
        h.upload(table)
        if h.run(dense) != 0:  # warm-up: buffers grow to their final size
            raise RuntimeError("solver reported a singular system")
        handles.append(h)

    def work(h):
        for _ in range(per_stream):
            if h.run(dense, 0, reuse) != 0:
                raise RuntimeError("solver reported a singular system")

    if streams >= 4 and prio is None:
        del os.environ["NODAL_STREAM_PRIORITY"]
    threads = [threading.Thread(target=work, args=(h,)) for h in handles]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    elapsed = time.perf_counter() - t0
    for h in handles:
        h.close()
    n = streams * per_stream
    return {"streams": streams, "circuits": n, "symbolic_phases_kept": bool(reuse), "circuits_per_sec": n / elapsed,
            "ms_per_circuit": elapsed / n * 1e3, "ms_latency_per_solve": elapsed / per_stream * 1e3}


def print_solution_seconds(device):
    """str(Solution) of the 1e6-node grid (sorted() over 1e6 string names, shortest-repr
    formatting of 1e6 doubles) -- the reference's output contract at scale, on the netlist as the command line gets it:
    read from its file (the potentials' lines then come from libnodal_csv.so on the host threads).  Returns (seconds,
    seconds of Netlist(path))."""
    import tempfile
    import numpy as np
    from nodal_amd import generators as gen
    from nodal_amd.circuit import Solution
    from nodal_amd.netlist import Netlist
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        path = os.path.join(d, "grid1000.csv")
        gen.write_csv(gen.grid_rows(1000), path)
        Netlist(path)  # (pages the file and the tokenizer in)
        t0 = time.perf_counter()
        nl = Netlist(path)
        read_s = time.perf_counter() - t0
    sol = Solution(np.linspace(0.0, 1.0, nl.nums["kcl"]), nl, [])
    t0 = time.perf_counter()
    text = str(sol)
    dt = time.perf_counter() - t0
    assert text.count("\n") == nl.nums["kcl"]
    return dt, read_s


def direct_route_times(device):
    """The route behind the iterations (csrc/sparse_direct.hip: multifrontal LU + fp64 refinement; what a
    system the presolve declines or an iteration gives up on is solved by) on config 5's matrix, forced:
    first solve (analysis on the host + numeric factorisation + refinement) and a repeated one (analysis kept)."""
    from nodal_amd import _ffi
    from nodal_amd import generators as gen
    table = gen.cfg5_table(1000)
    h = _ffi.Handle(device)
    h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)
    h.upload(table)
    h.assemble_symbolic()
    h.assemble_numeric()
    ms = []
    for _ in range(2):
        t0 = time.perf_counter()
        _x, info, iters, _rr = h.solve_sparse(method=_ffi.SPARSE_DIRECT, download=False)
        h.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    res = h.residual()
    h.close()
    return {"workload": "cfg5: grid(1000) + 1% E + CCCS/VCVS, n = 1 034 717, nodal_solve_sparse(NODAL_SPARSE_DIRECT)",
            "first_ms": ms[0], "repeated_ms_analysis_kept": ms[1], "info": info, "refinement_iterations": iters,
            "scaled_residual": res,
            "note": "not on the default route of this configuration (presolve + FGMRES: see also.cfg5); the "
                    "reference's SuperLU takes 33-55 s on this matrix"}


def resistance_sweep_times(device, npairs=192):
    """SURVEY 8f N1 on the sparse path: equivalent resistance of `npairs` random node pairs of config 3's network on
    one set of stamps and one hierarchy (nodal_solve_pairs; the reference re-reads the netlist, re-stamps and calls
    spsolve per pair, nodal/equiv.py:31-61).  The first pair is a solve of its own (it sets the hierarchy up); the
    others go sixteen at a time through the block iteration (csrc/sagg_multi.h), stopped on the functional."""
    import numpy as np
    from nodal_amd import _ffi
    from nodal_amd import generators as gen
    table = gen.grid_table(1000)
    rng = np.random.RandomState(3)
    ia = rng.randint(0, table.K, size=npairs).astype(np.int32)
    ib = rng.randint(-1, table.K, size=npairs).astype(np.int32)
    ib[ib == ia] = -1
    h = _ffi.Handle(device)
    h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)
    h.upload(table)
    h.assemble_symbolic()
    h.assemble_numeric()
    secs = []
    for _ in range(2):
        t0 = time.perf_counter()
        res, info = h.solve_pairs(ia, ib, False)
        secs.append(time.perf_counter() - t0)
    # the factor-once route (csrc/sparse.hip sparse_solve_pairs_direct: one multifrontal LU, sixteen pairs per
    # substitution + one refinement step), forced: automatic from 256 pairs on when the analysis is at hand
    os.environ["NODAL_PAIRS_DIRECT"] = "1"
    try:
        dsecs = []
        for _ in range(2):
            t0 = time.perf_counter()
            dres, dinfo = h.solve_pairs(ia, ib, False)
            dsecs.append(time.perf_counter() - t0)
        big = 1024
        ja = rng.randint(0, table.K, size=big).astype(np.int32)
        jb = rng.randint(-1, table.K, size=big).astype(np.int32)
        jb[jb == ja] = -1
        t0 = time.perf_counter()
        _r, _i = h.solve_pairs(ja, jb, False)
        dbig = time.perf_counter() - t0
    finally:
        del os.environ["NODAL_PAIRS_DIRECT"]
    os.environ["NODAL_PAIRS_DIRECT"] = "0"
    try:
        t0 = time.perf_counter()
        _r, _i = h.solve_pairs(ja, jb, False)
        bbig = time.perf_counter() - t0
    finally:
        del os.environ["NODAL_PAIRS_DIRECT"]
    h.close()
    agree = float(np.abs(dres - res).max() / np.abs(res).max())
    return {"workload": f"cfg3's network (grid(1000), 1e6 nodes), {npairs} random pairs, nodal_solve_pairs(dense = 0)",
            "first_s": secs[0], "repeated_s": secs[1], "ms_per_pair": secs[1] / npairs * 1e3, "info": int(info),
            "factor_once": {"what": "the same pairs through one sparse LU + sixteen right-hand sides per substitution (forced)",
                            "first_s_with_analysis": dsecs[0], "repeated_s_analysis_kept": dsecs[1],
                            "ms_per_pair": dsecs[1] / npairs * 1e3, "info": int(dinfo),
                            "max_relative_difference_to_block_iteration": agree,
                            "pairs_1024_s": dbig, "pairs_1024_block_iteration_s": bbig},
            "note": "first_s includes the hierarchy setup and the growth of the block iteration's buffers; parity of "
                    "the block iteration with one solve per pair and with a sparse LU: tests/test_gpu_parity.py::"
                    "test_block_iteration_matches_one_solve_per_pair; of the factor-once route: tests/test_gpu_direct.py::"
                    "test_factor_once_route_of_a_pair_sweep_matches_the_oracle"}


def make_workload(name, rank, world, device, dist, per_step, force_collective=False):
    if name == "cfg4":
        return BatchShard(rank, world, per_step, device, dist, force_collective)
    return SingleCircuits(name, rank, per_step, device, dist, force_collective)


def time_workload(name, rank, world, device, dist, steps, warmup, per_step, force_collective=False, reuse=False):
    import torch
    wl = make_workload(name, rank, world, device, dist, per_step, force_collective)
    if reuse:
        wl.reuse = True
    for _ in range(max(warmup, 1)):  # at least one untimed pass: buffers grow to their final size
        wl.step()
    wl.reset_stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = dict(name=name, desc=wl.desc, dense=wl.dense, elapsed=elapsed, steps=steps,
              circuits_per_step=wl.per_step, circuits=wl.circuits_done,
              phase_ms=(wl.phase / max(wl.circuits_done if name != "cfg4" else steps, 1)).tolist(),
              kern_ms=wl.kern_ms, kern_n=wl.kern_n, kern_alg=wl.kern_alg, table=wl.table,
              gather_ms=getattr(wl, "gather_ms", 0.0) / max(steps, 1))
    st.update(wl.finish())
    return st


# ---------------------------------------------------------------------------------
# reporting
# ---------------------------------------------------------------------------------

def roofline_of(st, circuits_per_sec_per_gpu):
    b_asm, b_solve = alg_bytes(st["table"], st["n"], st["nnz"])
    out = None
    if st["kern_n"]:
        avg_s = st["kern_ms"] / st["kern_n"] * 1e-3
        if st["dense"]:
            achieved = st["kern_alg"] / avg_s / 1e12
            out = {"bound": "mfma", "kernel": "gemm_sub_kernel<SUB, TA, UPPER>: bulk update A22 -= V^T W of the symmetric "
                                              "block elimination, K = 256 / 512 (gemm_f64.hip)",
                   "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                   "frac": achieved / FP64_MFMA_PEAK_TF, "traffic": pmc_traffic(st["name"]),
                   "avg_launch_us": avg_s * 1e6, "launches_timed": st["kern_n"],
                   "alg_flops_per_launch": st["kern_alg"],
                   "measured_instruction_ceiling": {"v_mfma_f64_4x4x4_4b": FP64_MFMA_444_TF,
                                                    "v_mfma_f64_16x16x4": FP64_MFMA_INSTR_TF,
                                                    "v_fma_f64": FP64_VALU_FMA_TF, "unit": "TFLOP/s"}}
        else:
            achieved = st["kern_alg"] / avg_s / 1e9
            out = {"bound": "hbm", "kernel": DOMINANT_SPARSE_KERNEL.get(st["name"]),
                   "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(st["name"]),
                   "avg_launch_us": avg_s * 1e6, "launches_timed": st["kern_n"],
                   "alg_bytes_per_launch": st["kern_alg"]}
    if out is None:
        out = {"bound": "mfma" if st["dense"] else "hbm", "kernel": None, "achieved": None,
               "peak": FP64_MFMA_PEAK_TF if st["dense"] else HBM_PEAK_GBS,
               "unit": "TFLOP/s" if st["dense"] else "GB/s", "frac": None, "traffic": None}
    if not st["dense"]:
        classes = profile_classes(st["name"])
        if classes and classes.get("by_class"):
            # `roofline` describes the class of kernels that DOMINATES BY TIME in this run's kernel trace; the
            # event-timed Krylov SpMV -- the fastest kernel of the solve, 4 % of its time -- moves to `best_kernel`
            by = classes["by_class"]
            dom = max(by, key=lambda k: by[k].get("share_of_gpu_time", 0.0))
            best = {k: out.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                            "launches_timed", "alg_bytes_per_launch")}
            d = by[dom]
            launches = max(d.get("launches", 0), 1)
            gbs = d.get("GB_per_s")
            what = {"coarse_levels": "coarse-level cycle kernels (levels >= 1 and the tail: latency-bound, 4-7 us "
                                     "per dependent launch whatever the rows)",
                    "level0_passes": "level-0 passes (smoother, transfer, Krylov kernels over the 1e6-row level)",
                    "hierarchy_setup": "multigrid setup kernels", "stamping": "stamping kernels"}.get(dom, dom)
            # bytes from counters where a committed PMC pass has them (per dispatch of the class), else none; the
            # algorithmic bytes of the coarse levels are an ESTIMATE (64 B x grid threads: tools/prof_classes.py) and say so
            pmc = pmc_by_class(st["name"])
            pmc_cls = (pmc or {}).get("by_class", {}).get(dom)
            avg_us = d.get("us", 0.0) / launches
            traffic = pmc_cls.get("hbm_bytes_per_dispatch") if pmc_cls else None
            out = {"bound": "hbm", "kernel": f"class `{dom}`: {what}", "achieved": gbs, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": (gbs / HBM_PEAK_GBS) if gbs else None,
                   "achieved_is": ("algorithmic bytes ESTIMATED as 64 B x grid threads / measured time" if dom == "coarse_levels"
                                   else "algorithmic bytes (rows x bytes per row) / measured time"),
                   "traffic": traffic,
                   "traffic_source": (f"{pmc['file']}: FETCH_SIZE x 2 + WRITE_SIZE of every dispatch of the class, separate "
                                      "--pmc passes (tools/pmc_by_class.py)") if traffic else None,
                   "traffic_GB_per_s": (traffic / (avg_us * 1e-6) / 1e9) if traffic and avg_us else None,
                   "traffic_frac_of_peak": (traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic and avg_us else None,
                   "share_of_gpu_time": d.get("share_of_gpu_time"),
                   "avg_launch_us": avg_us, "launches": d.get("launches"),
                   "alg_bytes_per_launch": (d.get("alg_bytes", 0.0) / launches) if d.get("alg_bytes") else None,
                   "how": "kernel durations of this run's rocprofv3 --kernel-trace child, summed per class; algorithmic "
                          "bytes per launch = rows x bytes per row of what the kernel reads and writes "
                          "(tools/prof_classes.py)",
                   "best_kernel": best}
            if pmc:
                out["hbm_bytes_per_circuit_by_class"] = {k: v.get("hbm_bytes_per_circuit") for k, v in pmc["by_class"].items()}
                out["hbm_bytes_per_circuit"] = pmc.get("hbm_bytes_per_circuit")
            out["top_kernel_by_time"] = classes.get("top_kernel_by_time")
            out["by_class"] = by
            out["by_class_source"] = classes.get("source")
    if st["dense"]:
        flops = 2.0 / 3.0 * st["n"] ** 3 + 2.0 * st["n"] ** 2
        out["end_to_end"] = {"what": "circuits/s x (2/3 n^3 + 2 n^2) / fp64 matrix peak",
                             "flops_per_circuit": flops,
                             "frac": circuits_per_sec_per_gpu * flops / (FP64_MFMA_PEAK_TF * 1e12)}
        # the symmetric block elimination executes about half of the LU count it is priced against (only the upper
        # block triangle is updated): n^3 / 3 + the chain's inversions and small products (~ 3 n 256^2 x 2)
        executed = st["n"] ** 3 / 3.0 + 2.0 * st["n"] ** 2 + 6.0 * st["n"] * 256.0 * 256.0
        out["end_to_end_executed"] = {"what": "circuits/s x flops actually executed (symmetric form: n^3 / 3 + chain) "
                                              "/ fp64 matrix peak -- the MFMA utilisation of the whole solve",
                                      "flops_per_circuit": executed,
                                      "frac": circuits_per_sec_per_gpu * executed / (FP64_MFMA_PEAK_TF * 1e12)}
    else:
        out["end_to_end"] = {"what": "circuits/s x B_alg / 8e12 (SURVEY.md 8d primary figure)",
                             "B_asm": b_asm, "B_solve_min": b_solve, "B_alg": b_asm + b_solve,
                             "frac": circuits_per_sec_per_gpu * (b_asm + b_solve) / (HBM_PEAK_GBS * 1e9)}
    return out


DOMINANT_SPARSE_KERNEL = {
    # The solve is ~900 launches and no kernel holds more than 9 % of it (profiles/r02_*_kernel_stats.txt);
    # the largest class by bytes is the pass over the level-0 matrix: the Krylov SpMV timed here and
    # the two smoother passes (k_smooth_residual<5>, k_post<5, true>) that read the same ELL arrays.
    "cfg3": "f_dir_spmv<5>: direction update + fp64 ELL SpMV of the flexible CG on the 1e6-node level (sagg_cycle.h)",
    "cfg4": "f_dir_spmv<5>: direction update + fp64 ELL SpMV of the flexible CG on the block-diagonal fine level (sagg_cycle.h)",
    "cfg5": "k_ell_spmv<W>: fp64 ELL SpMV of FGMRES on the presolved system (sagg_cycle.h)",
}


def cpu_baseline(name, table):
    """The oracle (reference algorithm restated; same numpy / scipy calls the reference
    makes) on this box's host cores.  One circuit for cfg2 / cfg3 / cfg5, eight members
    for cfg4; `value` = circuits actually solved / time."""
    import numpy as np
    from oracle import nodal_oracle as oracle
    from nodal_amd import generators as gen
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    dense = name == "cfg2"
    members = 8 if name == "cfg4" else 1
    t_asm = t_solve = 0.0
    x0 = None
    for m in range(members):
        t = table
        if name == "cfg4":
            t = table.truncated(table.ncomp)
            t.value[:-1] = gen.cfg4_values(m, 100)
        t0 = time.perf_counter()
        G, A = oracle.assemble_fast(t)
        G = G.toarray() if dense else G.tocsr()
        t_asm += time.perf_counter() - t0
        t0 = time.perf_counter()
        x, _ = oracle.solve(G, A, not dense)
        t_solve += time.perf_counter() - t0
        x0 = float(x[0]) if x0 is None else x0
    if dense:
        cores, what = blas_threads, "numpy.linalg.solve (LAPACK dgesv)"
    else:
        cores, what = 1, "scipy.sparse.linalg.spsolve (SuperLU)"
    return {"value": members / (t_asm + t_solve), "unit": "circuits/s", "cores": cores, "kind": "port",
            "sample": f"{members} circuit(s) of {name}: vectorised numpy stamping {t_asm:.2f} s + {what} "
                      f"{t_solve:.2f} s (host has {os.cpu_count()} cores); the reference's own per-component "
                      "Python stamping is slower (BASELINE.md section 2)",
            "solve_only_circuits_per_s": members / t_solve, "seconds": t_asm + t_solve, "x0": x0}


def summary(st, world, with_cpu):
    circuits = st["circuits"] * world
    value = circuits / st["elapsed"]
    out = {"workload": f"{st['name']}: {st['desc']}", "circuits_per_sec": value,
           "ms_per_solve": st["elapsed"] / st["circuits"] * 1e3,
           "phase_ms": {"symbolic": st["phase_ms"][0], "numeric": st["phase_ms"][1], "solve": st["phase_ms"][2]},
           "h2d_ms": st["h2d_ms"], "d2h_ms": st["d2h_ms"], "h2d_bytes": st["h2d_bytes"],
           "d2h_bytes": st["d2h_bytes"], "scaled_residual": st["resid"],
           "solver": {"iterations": st["iterations"], "amg_levels": st["amg_levels"]},
           "roofline": roofline_of(st, value / world)}
    if not st["dense"] and st["name"] != "cfg4":
        b_asm, _ = alg_bytes(st["table"], st["n"], st["nnz"])
        asm_ms = st["phase_ms"][0] + st["phase_ms"][1]
        if asm_ms > 0:
            out["assembly"] = {"what": "B_asm / (symbolic + numeric), SURVEY.md 8d", "B_asm": b_asm, "ms": asm_ms,
                               "GB_per_s": b_asm / asm_ms / 1e6, "frac_of_hbm_peak": b_asm / asm_ms / 1e6 / HBM_PEAK_GBS}
    if st["name"] == "cfg5":
        out["h2d_note"] = ("h2d_ms is the whole nodal_upload_components call: DMA of the eight columns plus, because the "
                           "table has branches, the host copy the presolve plans on and the listing of its branch rows "
                           "(outside `value`, inside pcie_inclusive)")
    if st["name"] != "cfg4" and st.get("h2d_ms") is not None:
        # table up (pinned host memory -> HBM by DMA) and x down for EVERY circuit: what `value` leaves out
        per = st["elapsed"] / st["circuits"] * 1e3 + st["h2d_ms"] + st["d2h_ms"]
        out["pcie_inclusive"] = {"circuits_per_sec": world * 1e3 / per, "ms_per_circuit": per,
                                 "h2d_GB_per_s": st["h2d_bytes"] / st["h2d_ms"] / 1e6 if st["h2d_ms"] > 0 else None,
                                 "h2d_first_ms": st.get("h2d_first_ms")}
    if st.get("gathered_ok") is not None:
        out["gather_ms_per_step"] = st["gather_ms"]  # cfg4: the gather; cfg3: what the overlap left the host waiting for
        out["gathered_ok"] = st["gathered_ok"]
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(st["name"], st["table"])
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if abs(out["cpu_baseline"]["x0"] - st["x0"]) > 1e-9 * max(1.0, abs(st["x0"])):
            raise RuntimeError(f"{st['name']}: GPU x[0]={st['x0']!r} differs from the oracle's "
                               f"{out['cpu_baseline']['x0']!r}")
    return out


# ---------------------------------------------------------------------------------
# entry
# ---------------------------------------------------------------------------------

def launch_ranks(n, argv):
    """Start `n` rank processes of this script (fresh children: this process has made no
    GPU call), relay rank 0's JSON line, exit with the first failing rank's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective: stop them (by their own PIDs) and
    # report its exit code instead of hanging until the process-group timeout
    code = 0
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if failed:
            code = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        code = code or (p.returncode or 0)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    sys.exit(code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--per-step", type=int, default=0, help="circuits per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--force-collective", action="store_true",
                    help="cfg4 with the process group initialised and the all_gather run even at N = 1 "
                         "(backend nccl = RCCL, world_size 1: the collective path on a one-GPU box)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="streams of the extra concurrent-throughput figure (0: skip it)")
    ap.add_argument("--no-classes", action="store_true",
                    help="skip the kernel-trace child that measures roofline.by_class (it is one itself)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.force_collective) and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])

    if args.gpus == 1 and not args.force_collective and not args.no_classes and "WORLD_SIZE" not in os.environ:
        # BEFORE this process touches the GPU: the same workload in a child under the kernel tracer
        wname = args.workload or "cfg3"
        if wname in ("cfg3", "cfg5"):
            sizes = {"cfg3": (999999, 4995995), "cfg5": (1034717, 5100208)}[wname]
            LIVE_CLASSES[wname] = live_classes(wname, *sizes)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    # NODAL_BENCH_REHEARSE=1: every rank shares GPU 0 and the collectives run over gloo -- a
    # rehearsal of the launcher, the sharding and the in-loop gather on a one-GPU box (RCCL
    # refuses two ranks on one device).  Numbers from such a run are not benchmark results.
    rehearse = os.environ.get("NODAL_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    # (a launcher may give every rank ONE visible device: that device is 0 for all of them)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    saved_stdout = None
    if world > 1 or args.force_collective:
        # rank 0 prints ONE JSON line on stdout: whatever the communication libraries print there when they start
        # (RCCL's version banner, gloo's "connected to peers") goes to stderr instead
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist_mod.init_process_group("gloo")
        else:
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist = dist_mod

    # ONE headline workload at every N: config 3 (BASELINE.json's metric is "circuits/sec ... N-node resistor grid,
    # 1/2/4/8" -- one workload); --force-collective keeps config 4 as its default (the block gather on one GPU)
    name = args.workload or ("cfg4" if args.force_collective else "cfg3")
    per_step = args.per_step or CIRCUITS_PER_STEP[name]
    st = time_workload(name, rank, world, local, dist, args.steps, args.warmup, per_step, args.force_collective)
    head = summary(st, world, with_cpu=(rank == 0 and world == 1 and not args.no_cpu))
    out = {
        "metric": "circuits_per_sec",
        "value": head["circuits_per_sec"],
        "unit": "circuits/s",
        "n_gpus": world,
        "world_size": dist.get_world_size() if dist is not None else 1,
        "backend": (dist.get_backend() + (" (= RCCL)" if dist.get_backend() == "nccl" else " (rehearsal)"))
                   if dist is not None else None,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": st["elapsed"] / args.steps * 1e3,
        "ms_per_solve": head["ms_per_solve"],
        "higher_is_better": True,
        # what is scaled: the number of independent circuits (per-GPU work fixed: `circuits_per_gpu_per_step` each)
        "scaling": "weak",
        "scaling_what": ("single GPU: nothing is sharded" if world == 1 and dist is None else
                         ("the members of a value sweep: 128 per GPU as one block system" if name == "cfg4" else
                          "the number of independent circuits: every rank solves its own, x of each all_gathered")),
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": head["workload"], "circuits_per_gpu_per_step": st["circuits_per_step"],
                   "n": st["n"], "nnz": st["nnz"],
                   "parallelism": ("batch members sharded over ranks, all_gather of x"
                                   if name == "cfg4" and (world > 1 or args.force_collective)
                                   else (f"independent circuits x{world}, every x all_gathered" if dist is not None
                                         else "one GPU"))},
    }
    for key in ("phase_ms", "h2d_ms", "d2h_ms", "h2d_bytes", "d2h_bytes", "scaled_residual", "solver",
                "roofline", "assembly", "pcie_inclusive", "h2d_note", "cpu_baseline", "speedup_vs_cpu_baseline",
                "gather_ms_per_step", "gathered_ok"):
        if key in head:
            out[key] = head[key]
    if rank == 0 and world == 1 and name in ("cfg3", "cfg5") and not args.no_also:
        # the same circuits with the symbolic phases of the previous one kept (nodal_run(reuse_symbolic = 1):
        # stamping lists and, keyed on the same struct_epoch, the multigrid hierarchy's aggregates and
        # patterns; values, Galerkin sums and the solve are redone) -- a value sweep on one topology
        s3 = time_workload(name, 0, 1, local, None, max(2, args.steps // 4), 1, per_step, reuse=True)
        r3 = summary(s3, 1, with_cpu=False)
        out["reuse_symbolic"] = {k: r3[k] for k in ("circuits_per_sec", "ms_per_solve", "phase_ms", "solver",
                                                    "scaled_residual")}
    if rank == 0 and world == 1 and name != "cfg4" and args.concurrent > 1:
        per_stream = 12 if name != "cfg2" else 8
        conc = [concurrent_throughput(name, local, args.concurrent, per_stream)]
        if name in ("cfg3", "cfg5"):
            for streams in sorted({2, 3, args.concurrent}):
                conc.append(concurrent_throughput(name, local, streams, per_stream, reuse=True))
        out["concurrent"] = max(conc, key=lambda c: c["circuits_per_sec"])
        out["concurrent"]["all"] = [{k: c[k] for k in ("streams", "symbolic_phases_kept", "circuits_per_sec")}
                                    for c in conc]
    if rank == 0 and world == 1 and name == "cfg3" and not args.no_also:
        out["print_1e6_s"], out["netlist_read_1e6_s"] = print_solution_seconds(local)
    if world > 1 and name != "cfg4" and not args.no_also:
        # config 4 -- the value sweep that shards as one block system per rank, 128 members per GPU, gathered over
        # RCCL -- measured at every N beside the headline (every rank takes part; rank 0 reports)
        s4 = time_workload("cfg4", rank, world, local, dist, 6, 1, 128, False)
        r4 = summary(s4, world, with_cpu=False)
        out["also"] = {"cfg4": {k: r4[k] for k in ("workload", "circuits_per_sec", "ms_per_solve", "phase_ms", "solver",
                                                    "scaled_residual", "gather_ms_per_step", "gathered_ok") if k in r4}}
    if rank == 0 and world == 1 and not args.no_also and not args.force_collective:
        also = {}
        for other in ("cfg4", "cfg5", "cfg2"):
            if other == name:
                continue
            s2 = time_workload(other, 0, 1, local, None, {"cfg4": 6, "cfg5": 2, "cfg2": 1}[other], 1,
                               {"cfg4": 128, "cfg5": 4, "cfg2": 4}[other])
            also[other] = summary(s2, 1, with_cpu=not args.no_cpu)
            if other == "cfg4":  # config 4 IS a value sweep on one topology: the next 128 members with the block
                # system's symbolic phases kept (nodal_run_batch(reuse_symbolic = 1))
                s3 = time_workload(other, 0, 1, local, None, 6, 1, 128, reuse=True)
                r3 = summary(s3, 1, with_cpu=False)
                also[other]["reuse_symbolic"] = {k: r3[k] for k in ("circuits_per_sec", "ms_per_solve", "phase_ms", "solver",
                                                                     "scaled_residual")}
            if other == "cfg5":  # the general path with the symbolic phases kept (stamping lists, the presolved
                # netlist's lists, the hierarchy's patterns); the presolve's plan and rewrite are redone: they read values
                s3 = time_workload(other, 0, 1, local, None, 2, 1, 4, reuse=True)
                r3 = summary(s3, 1, with_cpu=False)
                also[other]["reuse_symbolic"] = {k: r3[k] for k in ("circuits_per_sec", "ms_per_solve", "phase_ms", "solver",
                                                                     "scaled_residual")}
        also["sparse_direct"] = direct_route_times(local)
        also["resistance_sweep"] = resistance_sweep_times(local)
        out["also"] = also
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        os.dup2(2, 1)  # (the libraries may speak again when the group goes)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
