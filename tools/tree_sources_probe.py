"""A branching tree of resistors WITH sources (E at some leaves, a VCVS and a CCCS inside): the general sparse path
on a topology the multigrid hierarchies dislike; what the iteration does not solve goes to the direct route
(csrc/sparse_direct.hip), which orders tree-like parts leaves-first.  python tools/tree_sources_probe.py [nodes]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nodal_amd as n
from nodal_amd import _ffi
from nodal_amd.lowering import lower
from oracle import nodal_oracle as oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = random.Random(11)
rows = []
name = lambda k: "g" if k == 0 else str(k)
for k in range(1, N):
    rows.append([f"r{k}", "R", repr(rng.uniform(0.5, 2.0)), name(k), name((k - 1) // 2)])
leaves = [k for k in range(N // 2, N)]
for q, k in enumerate(rng.sample(leaves, max(4, N // 1000))):
    rows.append([f"e{q}", "E", repr(rng.uniform(1, 5)), f"s{q}", "g"])
    rows.append([f"rs{q}", "R", "1.0", f"s{q}", name(k)])
rows.append(["v0", "VCVS", "0.5", "x0", "g", name(5), name(11)])
rows.append(["rx0", "R", "2.0", "x0", name(23)])
rows.append(["f0", "CCCS", "0.3", name(40), "g", name(2), name(0 + 1), "r2"])
rows.append(["a0", "A", "1.0", name(N - 1), "g"])
t0 = time.time()
table = lower(n.Netlist.from_rows(rows))
print(f"{len(rows)} rows lowered in {time.time() - t0:.1f} s: K {table.K}, B {table.B}", flush=True)
h = _ffi.Handle(0)
h.upload(table)
h.assemble_symbolic()
assert h.assemble_numeric()[0] == _ffi.OK
for rep in range(2):
    t0 = time.time()
    x, info, iters, rr = h.solve_sparse()
    print(f"run {rep}: {(time.time() - t0) * 1e3:.1f} ms, info {info}, iterations {iters}, scaled residual {h.residual():.1e}", flush=True)
if N <= 300000:
    G, A = oracle.assemble_fast(table)
    t0 = time.time()
    xo, _ = oracle.solve(G.tocsr(), A, True)
    print(f"SuperLU {time.time() - t0:.2f} s; normwise distance {np.abs(x - xo).max() / np.abs(xo).max():.2e}")
h.close()
