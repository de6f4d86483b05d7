"""Debugging aid: one case under NODAL_POISON with NaN probes (NODAL_NANCHECK=1); run on the GPU box.
Usage: python tools/poison_case.py vccs"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nodal_amd as n
from nodal_amd import generators as gen

def vccs():
    rows = list(gen.grid_rows(72))[:-1]
    rows.append(["e0", "E", "1", "1", "g"])
    rows += [["ru1", "R", "1", "u1", "u2"], ["ru2", "R", "2", "u2", "u3"], ["ru3", "R", "3", "u3", "u4"],
             ["ru4", "R", "1", "u4", "u1"], ["ai", "A", "1", "u2", "u4"],
             ["dq", "VCCS", "0.5", "u1", "g", "u1", "g"]]
    return rows

rows = {"vccs": vccs}[sys.argv[1]]()
nl = n.Netlist.from_rows(rows)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    sol = n.Circuit(nl, sparse=True).solve()
print("finite:", np.isfinite(sol.result).all(), "warnings:", [str(x.message) for x in w])
