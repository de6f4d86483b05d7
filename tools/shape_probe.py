"""Sweep of problem shapes through the sparse path on the GPU box (development aid): which hierarchy
each one gets, iterations, time, scaled residual.  Looks for cliffs next to the benchmark configs.

  python tools/shape_probe.py grid:1200 rgrid:1000 cfg5:700 batch:64x140 batch:1024x35 grid3:80
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen
from nodal_amd.batch import BatchSolver


def grid3_table(N, seed=0):
    """N^3 cube of unit-ish resistors, node 0 driven, last node grounded."""
    idx = np.arange(N ** 3).reshape(N, N, N)
    a = np.concatenate([idx[:-1].ravel(), idx[:, :-1].ravel(), idx[:, :, :-1].ravel()])
    b = np.concatenate([idx[1:].ravel(), idx[:, 1:].ravel(), idx[:, :, 1:].ravel()])
    vals = np.random.default_rng(seed).uniform(0.5, 2.0, a.size)
    return gen.passive_table(a, b, vals, 0, N ** 3 - 1)


def single(table, reps=3):
    h = _ffi.Handle(0)
    h.upload(table)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        info = h.run(False)
        h.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    it, lv, rr = h.solve_info()
    out = f"n {h.n} nnz {h.nnz}: {best:.2f} ms, info {info}, iterations {it}, levels {lv}, residual {h.residual():.1e}"
    h.close()
    return out


def main():
    for spec in sys.argv[1:]:
        kind, arg = spec.split(":", 1)
        if kind == "grid":
            out = single(gen.grid_table(int(arg)))
        elif kind == "rgrid":  # random resistances within a factor 4
            N = int(arg)
            out = single(gen.grid_table(N, np.random.default_rng(1).uniform(0.5, 2.0, gen.grid_resistor_count(N))))
        elif kind == "cfg5":  # cfg5:N or cfg5:N:seed
            parts = [int(v) for v in arg.split(":")]
            out = single(gen.cfg5_table(*parts))
        elif kind == "grid3":
            out = single(grid3_table(int(arg)))
        elif kind == "batch":
            reuse = arg.endswith("r")  # batch:4x1000r: symbolic phases kept from the first run on
            members, N = (int(v) for v in arg.rstrip("r").split("x"))
            table = gen.grid_table(N)
            vals = np.ones((members, table.ncomp))
            for i in range(members):
                vals[i, :-1] = gen.cfg4_values(i, N)
            s = BatchSolver(table, 0)
            s.upload_values(vals)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                s.run(sparse=True, reuse_symbolic=reuse, download=False)
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None or dt < best else best
            it, lv, rr = s.h.solve_info()
            out = (f"{members} x n {table.n}: {best:.2f} ms per shard = {members / best * 1e3:.0f} circuits/s, "
                   f"iterations {it}, levels {lv}, phases {['%.2f' % t for t in s.h.timings()]}")
        else:
            raise SystemExit(f"unknown spec {spec}")
        print(f"{spec:>16}: {out}", flush=True)


if __name__ == "__main__":
    main()
