"""Per-call wall time of the C ABI for a doc-sized circuit (n = 5): python tools/tiny_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nodal_amd as n
from nodal_amd import _ffi
from nodal_amd.lowering import lower

rows = [["r1", "R", "2", "1", "4"], ["r2", "R", "0.5", "1", "2"], ["r3", "R", "1", "1", "g"],
        ["e1", "E", "8", "4", "g"], ["a1", "A", "4", "1", "2"], ["d1", "CCCS", "2", "2", "g", "1", "4", "r1"]]
nl = n.Netlist.from_rows(rows)
t0 = time.perf_counter(); table = lower(nl); t_lower = time.perf_counter() - t0
h = _ffi.Handle(0)
acc = np.zeros(6)
reps = 200
for r in range(reps + 5):
    t = [time.perf_counter()]
    h.upload(table); t.append(time.perf_counter())
    h.assemble_symbolic(); t.append(time.perf_counter())
    h.assemble_numeric(); t.append(time.perf_counter())
    x, info = h.solve_dense(); t.append(time.perf_counter())
    G, A = h.export_dense(); t.append(time.perf_counter())
    if r >= 5:
        acc[:5] += np.diff(t)
t0 = time.perf_counter()
for r in range(reps):
    x = n.Circuit(nl).solve().result
acc[5] = time.perf_counter() - t0
print(f"lower {t_lower * 1e6:.0f} us; per call (us): upload {acc[0] / reps * 1e6:.0f}, symbolic {acc[1] / reps * 1e6:.0f}, "
      f"numeric {acc[2] / reps * 1e6:.0f}, solve_dense {acc[3] / reps * 1e6:.0f}, export_dense {acc[4] / reps * 1e6:.0f}; "
      f"Circuit(nl).solve() {acc[5] / reps * 1e6:.0f} us")
h.close()
