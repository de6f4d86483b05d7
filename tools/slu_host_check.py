"""Driver of tools/slu_host_check (host emulation of the sparse direct route): builds test systems with
the oracle, runs the checker, compares with SuperLU.  Runs without a GPU."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse.linalg as spla
from nodal_amd import generators as gen
from oracle import nodal_oracle as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "slu_host_check")


def run(name, table):
    G, A = oracle.assemble_fast(table)
    G = G.tocsr()
    G.sort_indices()
    n = G.shape[0]
    with tempfile.TemporaryDirectory() as d:
        m, xo = os.path.join(d, "m.bin"), os.path.join(d, "x.bin")
        with open(m, "wb") as f:
            np.array([n, G.nnz], dtype=np.int64).tofile(f)
            G.indptr.astype(np.int32).tofile(f)
            G.indices.astype(np.int32).tofile(f)
            G.data.astype(np.float64).tofile(f)
            A.astype(np.float64).tofile(f)
        t0 = time.time()
        r = subprocess.run([EXE, m, xo], capture_output=True, text=True)
        dt = time.time() - t0
        if r.returncode != 0:
            print(name, "FAILED rc", r.returncode, r.stderr[-500:])
            return
        x = np.fromfile(xo, dtype=np.float64)
    t0 = time.time()
    ref = spla.spsolve(G.tocsc(), A)
    dref = time.time() - t0
    err = np.abs(x - ref).max() / np.abs(ref).max()
    res = np.abs(G @ x - A).max() / (np.abs(G).sum(axis=1).max() * np.abs(x).max() + np.abs(A).max())
    print(f"{name}: n={n} nnz={G.nnz} normwise {err:.2e} scaled residual {res:.2e}  host {dt:.2f}s  superlu {dref:.2f}s")
    print("   ", r.stderr.strip().replace("\n", "\n    "))


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [12, 40]
    for N in sizes:
        run(f"grid({N})", gen.grid_table(N))
        run(f"cfg5({N})", gen.cfg5_table(N))
    run("ladder(3000)", gen.ladder_table(3000))
    run("tree(5000)", gen.binary_tree_table(5000))
