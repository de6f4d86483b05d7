"""Resistance sweep on networks the low-degree elimination serves (ladders, trees): time per pair.
python tools/lowdeg_pairs_probe.py [npairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi, generators as gen

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, table in (("ladder(1e6)", gen.ladder_table(1000000)), ("tree(1e6)", gen.binary_tree_table(1000000)),
                    ("ladder(1e5)", gen.ladder_table(100000))):
    rng = np.random.RandomState(3)
    ia = rng.randint(0, table.K, size=npairs).astype(np.int32)
    ib = rng.randint(-1, table.K, size=npairs).astype(np.int32)
    ib[ib == ia] = -1
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    h.assemble_numeric()
    for rep in range(2):
        t0 = time.perf_counter()
        res, info = h.solve_pairs(ia, ib, False)
        dt = time.perf_counter() - t0
        print(f"{name}: {npairs} pairs in {dt * 1e3:.1f} ms = {dt / npairs * 1e3:.2f} ms per pair (info {info}, R[0] = {res[0]:.9f})", flush=True)
    h.close()
