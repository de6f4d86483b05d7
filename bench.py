#!/usr/bin/env python3
"""Benchmark of the hot path: component table resident in HBM -> solution x.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one pass of the hot path (symbolic + numeric assembly + solve, the
equivalent of the reference's `Circuit(netlist, sparse)` + `.solve()`,
nodal/nodal.py:306-336) over one batch of synthetic circuits per GPU.  Inputs
are uploaded before the timed region; nothing is cached between steps.

Workloads (BASELINE.json `configs`):
    cfg2  100x100 resistor grid, dense G, fp64 LU           (default, configs[1])
    cfg3  1000x1000 resistor grid, sparse CSR path
    cfg4  batch of 100x100 grids with per-member values, sparse path, shared
          symbolic phase inside each step
    cfg5  1000x1000 grid + 1% E + CCCS/VCVS, sparse path
With N > 1 every rank runs the same per-GPU work on its own members (weak
scaling, no data-path collective: independent circuits).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the
workload, timed with HIP events on the library's stream; `cpu_baseline` is the
oracle (the reference's algorithm restated, same numpy/scipy calls) timed on
this box's host cores; `also` carries the other single-GPU configurations.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_MFMA_PEAK_TF = 78.6    # vendor dense fp64 matrix peak
# measured on this hardware with tools/mfma_f64_peak.hip / tools/fp64_mix.hip
# (profiles/r01_mfma_f64_peak.txt, profiles/r01_fp64_mix.txt):
FP64_MFMA_444_TF = 73.0     # v_mfma_f64_4x4x4_4b_f64 -- the instruction gemm_f64.hip issues
FP64_MFMA_INSTR_TF = 36.2   # v_mfma_f64_16x16x4_f64, 138 cycles/instruction/wave
FP64_VALU_FMA_TF = 59.3     # v_fma_f64


def pmc_traffic(workload):
    """HBM bytes per launch of the workload's dominant kernel from the committed
    rocprofv3 PMC summary (FETCH_SIZE x2 + WRITE_SIZE, separate passes; see
    profiles/r01_pmc_traffic.json).  None if that workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["dominant"][workload]["hbm_bytes_per_launch"]
    except Exception:
        return None


def build_workload(name, rank, per_gpu):
    """Returns (table, values or None, dense, circuits_per_step, description)."""
    from nodal_amd import generators as gen
    if name == "cfg2":
        table = gen.grid_table(100)
        vals = np.ones((per_gpu, table.ncomp))
        for i in range(per_gpu):  # distinct members per rank, same topology
            member = rank * per_gpu + i
            if member > 0:
                vals[i, :-1] = gen.cfg4_values(member, 100)
        return table, vals, True, per_gpu, "grid(100) 1e4 nodes, dense G, fp64 LU"
    if name == "cfg4":
        from nodal_amd.batch import replicate_table
        table = gen.grid_table(100)
        vals = np.ones((per_gpu, table.ncomp))
        for i in range(per_gpu):
            vals[i, :-1] = gen.cfg4_values(rank * per_gpu + i, 100)
        # the shard's members are assembled and solved as ONE block-diagonal system
        return (replicate_table(table, vals), None, False, 1,
                f"batch of {per_gpu} grid(100) value sweeps per GPU, sparse CSR, block-diagonal")
    if name == "cfg3":
        return gen.grid_table(1000), None, False, 1, "grid(1000) 1e6 nodes, sparse CSR"
    if name == "cfg5":
        return gen.cfg5_table(1000), None, False, 1, "grid(1000)+1% E+CCCS/VCVS, sparse"
    raise SystemExit(f"unknown workload {name}")


def run_step(h, dense, members):
    for i in range(members):
        info = h.run(dense, member=i, reuse_symbolic=(i > 0))
        if info != 0:
            raise RuntimeError(f"solver reported info={info}")


def time_workload(name, rank, world, steps, warmup, per_gpu, dist, concurrent=1):
    """`concurrent` > 1 (independent-circuit workloads): the step's circuits are spread over that
    many handles, each driven by its own host thread -- independent solves overlap on the GPU
    (one circuit's latency-bound phases run beside another's bulk updates)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from nodal_amd import _ffi
    table, vals, dense, members, desc = build_workload(name, rank, per_gpu)
    concurrent = max(1, min(concurrent, members)) if name == "cfg2" else 1
    handles = []
    for _ in range(concurrent):
        hh = _ffi.Handle(torch.cuda.current_device())
        hh.upload(table)
        if vals is not None:
            hh.upload_values(vals)
        handles.append(hh)
    h = handles[0]
    pool = ThreadPoolExecutor(max_workers=concurrent) if concurrent > 1 else None

    def run_share(idx, first):
        # circuits idx, idx + concurrent, ... on handle idx; one symbolic assembly per step
        hh = handles[idx]
        for j, i in enumerate(range(idx, members, concurrent)):
            info = hh.run(dense, member=i, reuse_symbolic=not (first or (idx == 0 and j == 0)))
            if info != 0:
                raise RuntimeError(f"solver reported info={info}")

    def step(first=False):
        if pool is None:
            run_share(0, first)
        else:
            for f in [pool.submit(run_share, idx, first) for idx in range(concurrent)]:
                f.result()

    for w in range(warmup):
        step(first=(w == 0))
    if warmup == 0:
        step(first=True)  # every handle needs its symbolic phase once (untimed)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = kern_n = 0
    phase = np.zeros(3)
    for _ in range(steps):
        step()
        for hh in handles:
            ms, launches, alg = hh.kernel_stats()
            kern_ms += ms
            kern_n += launches
        phase += np.array(h.timings())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    resid = h.residual()
    x = h.download_x()
    iterations, amg_levels, _ = h.solve_info()
    if name == "cfg4":
        members = per_gpu  # circuits per step (one block-diagonal solve)
    stats = dict(elapsed=elapsed, members=members, dense=dense, desc=desc, table=table,
                 kern_ms=kern_ms, kern_n=kern_n, kern_alg=alg, resid=resid, x0=float(x[0]),
                 phase_ms=(phase / steps).tolist(), n=h.n, nnz=h.nnz, name=name,
                 iterations=iterations, amg_levels=amg_levels, concurrent=concurrent)
    for hh in handles:
        hh.close()
    if pool is not None:
        pool.shutdown()
    return stats


def roofline_of(stats):
    if stats["kern_n"] == 0:
        return None
    avg_s = stats["kern_ms"] / stats["kern_n"] * 1e-3
    if stats["dense"]:
        achieved = stats["kern_alg"] / avg_s / 1e12
        return {"bound": "mfma", "kernel": "gemm_sub_kernel (K=256 bulk update of the block elimination)",
                "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TF, "traffic": pmc_traffic(stats["name"]),
                "avg_launch_us": avg_s * 1e6, "launches_timed": stats["kern_n"],
                "alg_flops_per_launch": stats["kern_alg"],
                "measured_instruction_ceiling": {"v_mfma_f64_4x4x4_4b": FP64_MFMA_444_TF,
                                                 "v_mfma_f64_16x16x4": FP64_MFMA_INSTR_TF,
                                                 "v_fma_f64": FP64_VALU_FMA_TF, "unit": "TFLOP/s"}}
    achieved = stats["kern_alg"] / avg_s / 1e9
    return {"bound": "hbm", "kernel": "CSR-stream SpMV (pcg_spmv / spmv_kernel)", "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(stats["name"]), "avg_launch_us": avg_s * 1e6,
            "launches_timed": stats["kern_n"], "alg_bytes_per_launch": stats["kern_alg"]}


def cpu_baseline(name, table, members=1):
    """The oracle (reference algorithm restated; same numpy / scipy calls the
    reference makes) on this box's host cores, one circuit."""
    from oracle import nodal_oracle as oracle
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    t0 = time.perf_counter()
    G, A = oracle.assemble_fast(table)
    t_asm = time.perf_counter() - t0
    dense = name == "cfg2"
    t0 = time.perf_counter()
    if dense:
        x, _ = oracle.solve(G.toarray(), A, False)
        cores, what = blas_threads, "numpy.linalg.solve (LAPACK dgesv)"
    else:
        x, _ = oracle.solve(G.tocsr(), A, True)
        cores, what = 1, "scipy.sparse.linalg.spsolve (SuperLU)"
    t_solve = time.perf_counter() - t0
    return {"value": members / (t_asm + t_solve), "unit": "circuits/s", "cores": cores, "kind": "port",
            "sample": f"{members} circuit(s) of {name}: vectorised numpy stamping {t_asm:.2f} s + {what} "
                      f"{t_solve:.2f} s; the reference's own per-component Python stamping is "
                      "slower (BASELINE.md section 2)",
            "solve_only_circuits_per_s": members / t_solve, "x0": float(x[0])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--per-gpu", type=int, default=0, help="circuits per GPU per step")
    ap.add_argument("--concurrent", type=int, default=1,
                    help="cfg2: independent solves in flight per GPU (one handle + host thread each)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    per_gpu = args.per_gpu or {"cfg2": 2, "cfg4": 128}.get(args.workload, 1)
    st = time_workload(args.workload, rank, world, args.steps, args.warmup, per_gpu, dist, args.concurrent)
    circuits = st["members"] * args.steps * world
    out = {
        "metric": "circuits_per_sec",
        "value": circuits / st["elapsed"],
        "unit": "circuits/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": st["elapsed"] / args.steps * 1e3,
        "ms_per_solve": st["elapsed"] / (st["members"] * args.steps) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {st['desc']}", "circuits_per_gpu_per_step":
                   st["members"], "n": st["n"], "nnz": st["nnz"],
                   "parallelism": f"independent circuits x{world}",
                   "concurrent_solves_per_gpu": st["concurrent"]},
        "phase_ms": {"symbolic": st["phase_ms"][0], "numeric": st["phase_ms"][1],
                     "solve": st["phase_ms"][2]},
        "scaled_residual": st["resid"],
        "solver": {"iterations": st["iterations"], "amg_levels": st["amg_levels"]},
        "roofline": roofline_of(st),
    }
    if rank == 0 and world == 1:
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.workload, st["table"], st["members"])
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if not args.no_also:
            also = {}
            for other in ("cfg3", "cfg4", "cfg5"):
                if other == args.workload:
                    continue
                s2 = time_workload(other, 0, 1, 2, 1, {"cfg4": 128}.get(other, 1), None)
                also[other] = {"workload": s2["desc"],
                               "circuits_per_sec": s2["members"] * 2 / s2["elapsed"],
                               "ms_per_solve": s2["elapsed"] / (s2["members"] * 2) * 1e3,
                               "phase_ms": s2["phase_ms"], "scaled_residual": s2["resid"],
                               "solver": {"iterations": s2["iterations"], "amg_levels": s2["amg_levels"]},
                               "roofline": roofline_of(s2)}
            out["also"] = also
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
