"""Development probe: repeated solves of the island cases of tests/test_gpu_parity.py on pooled handles."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nodal_amd as n
from nodal_amd import generators as gen
from tests.test_gpu_parity import _island_rows

rows = list(gen.grid_rows(72))[:-1]
rows.append(["e0", "E", "1", "1", "g"])
rows += [["ru1", "R", "1", "u1", "u2"], ["ru2", "R", "2", "u2", "u3"], ["ru3", "R", "3", "u3", "u4"],
         ["ru4", "R", "1", "u4", "u1"], ["ai", "A", "1", "u2", "u4"],
         ["dq", "VCCS", "0.5", "u1", "g", "u1", "g"]]
nl = n.Netlist.from_rows(rows)
others = [n.Netlist.from_rows(_island_rows(v)) for v in ("one", "two_terms")]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if rep % 3 == 1:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            n.Circuit(others[rep % 2], sparse=True).solve()
    circ = n.Circuit(nl, sparse=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sol = circ.solve()
    x = np.asarray(sol.result)
    ok = np.isfinite(x).all() and abs(x[nl.nodenum["u2"]] - 0.7142857142857142) < 1e-9
    if not ok:
        bad += 1
        print("rep", rep, "BAD", [str(i.message) for i in w], "iters", getattr(circ, "iterations", None),
              "relres", getattr(circ, "relative_residual", None), flush=True)
print("bad", bad)
