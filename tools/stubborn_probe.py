"""Config 5 (1e6-node grid, 1 % sources) plus three cascaded amplifier stages the presolve cannot substitute:
with the stages kept as branch equations of the reduced system (default) against round 2's behaviour
(NODAL_PRESOLVE_KEEP=0: the presolve declines, full-system FGMRES).  python tools/stubborn_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi, generators as gen
from nodal_amd.netlist import Netlist

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lab = lambda k: "g" if k == N * N - 1 else str(k + 1)  # noqa: E731
rows = list(gen.cfg5_rows(N))
a1 = rows.pop()
rows += [["w1", "VCVS", "0.5", "y1", "g", lab(3), lab(N + 7)], ["ry1", "R", "1", "y1", lab(2 * N + 5)],
         ["w2", "VCVS", "0.7", "y2", "g", "y1", lab(3 * N + 2)], ["ry2", "R", "1", "y2", lab(4 * N + 9)],
         ["w3", "VCVS", "-0.4", "y3", "g", "y2", "y1"], ["ry3", "R", "2", "y3", lab(5 * N + 1)], a1]
t0 = time.time()
nl = Netlist.from_rows(rows)
from nodal_amd.lowering import lower
table = lower(nl)
print(f"netlist of {len(rows)} rows lowered in {time.time() - t0:.1f} s: K {table.K}, B {table.B}", flush=True)
h = _ffi.Handle(0)
h.upload(table)
for rep in range(3):
    t0 = time.perf_counter()
    info = h.run(False)
    h.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    it, lv, rr = h.solve_info()
    print(f"run {rep}: {dt:.1f} ms, info {info}, iterations {it}, scaled residual {h.residual():.1e}", flush=True)
h.close()
