"""Dense solve of small non-passive systems (partial pivoting): panel kernel against the per-column
kernels.  python tools/dense_small_probe.py [side ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi
from nodal_amd import generators as gen

for side in [int(v) for v in sys.argv[1:]] or [3, 8, 16, 22, 31, 44]:
    table = gen.cfg5_table(side)
    line = f"cfg5({side}) n={table.K + table.B:5d}:"
    for panel in (1, 0):
        h = _ffi.Handle(0)
        h.set_option(_ffi.OPT_FORCE_PIVOTING, 1)
        h.set_option(_ffi.OPT_GEPP_PANEL, panel)
        h.upload(table)
        h.assemble_symbolic()
        h.assemble_numeric()
        best = 1e9
        for _ in range(5):
            h.assemble_numeric()
            t0 = time.perf_counter()
            x, info = h.solve_dense()
            best = min(best, time.perf_counter() - t0)
        line += f"  panel={panel}: {best * 1e3:8.3f} ms (info {info}, residual {h.residual():.1e})"
        h.close()
    # CPU reference point: LAPACK dgesv (numpy) on the matrix the library assembled
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    h.assemble_numeric()
    Gd, A = h.export_dense()
    h.close()
    t0 = time.perf_counter(); np.linalg.solve(Gd, A); t1 = time.perf_counter() - t0
    print(line + f"  numpy dgesv {t1 * 1e3:.3f} ms", flush=True)
