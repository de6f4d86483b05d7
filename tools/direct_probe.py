"""Sparse direct route (csrc/sparse_direct.hip) on the GPU: accuracy against SuperLU and timings.
Usage: python tools/direct_probe.py [sizes ...]   (NODAL_TRACE=1 prints the phases)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi, generators as gen
from oracle import nodal_oracle as oracle


def run(name, table, ref=True):
    h = _ffi.Handle(0)
    h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)  # (a handle that is used alone, like bench.py's)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    out = []
    for rep in range(2):
        t0 = time.time()
        x, info, iters, rr = h.solve_sparse(method=_ffi.SPARSE_DIRECT)
        out.append((time.time() - t0) * 1e3)
    res = h.residual() if info == 0 else float("nan")
    err = float("nan")
    if ref and info == 0:
        G, A = oracle.assemble_fast(table)
        t0 = time.time()
        xo, _ = oracle.solve(G.tocsr(), A, True)
        tref = time.time() - t0
        err = np.abs(x - xo).max() / np.abs(xo).max()
    else:
        tref = float("nan")
    print(f"{name}: n={h.n} info={info} iters={iters} normwise {err:.2e} residual {res:.2e} "
          f"first {out[0]:.1f} ms, repeated {out[1]:.1f} ms (SuperLU {tref * 1e3:.0f} ms)", flush=True)
    h.close()


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [12, 40, 100]
    for N in sizes:
        run(f"grid({N})", gen.grid_table(N), ref=N <= 400)
        run(f"cfg5({N})", gen.cfg5_table(N), ref=N <= 400)
    run("ladder(20000)", gen.ladder_table(20000))
    run("tree(20000)", gen.binary_tree_table(20000))
