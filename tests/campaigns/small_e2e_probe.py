"""Wall time of Circuit(netlist) + .solve() for small netlists (what a user of the reference's CLI sees),
HIP path against the CPU restatement of the reference (oracle): python tests/campaigns/small_e2e_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nodal_amd import generators as gen
from nodal_amd.netlist import Netlist
from nodal_amd.circuit import Circuit
from oracle import nodal_oracle as oracle


def rows_small():
    return [["r1", "R", "2", "1", "4"], ["r2", "R", "0.5", "1", "2"], ["r3", "R", "1", "1", "g"],
            ["e1", "E", "8", "4", "g"], ["a1", "A", "4", "1", "2"], ["d1", "CCCS", "2", "2", "g", "1", "4", "r1"]]


def timeit(f, reps):
    f()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


cases = {"doc-sized (n=5)": rows_small(), "cfg5(8) n=70": gen.cfg5_rows(8), "cfg5(16) n=267": gen.cfg5_rows(16),
         "grid(20) n=399": gen.grid_rows(20), "cfg5(22) n=501": gen.cfg5_rows(22)}
for name, rows in cases.items():
    nl = Netlist.from_rows(rows)
    for sparse in (False, True):
        def ours():
            return Circuit(nl, sparse=sparse).solve().result
        def ref():
            return oracle.solve_netlist(nl, sparse)[0]
        t_ours, t_ref = timeit(ours, 10), timeit(ref, 3)
        err = np.abs(np.asarray(ours()) - ref()).max()
        print(f"{name:18s} sparse={sparse!s:5s}: HIP path {t_ours:8.3f} ms, reference algorithm on the CPU {t_ref:8.3f} ms, "
              f"max difference {err:.1e}", flush=True)
