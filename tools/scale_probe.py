"""The sparse passive path under extreme scalings of the source and of the resistances (the cycle keeps its vectors
in f32: csrc/sagg.hip, cyc_t): iterations and scaled residual.  python tools/scale_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi, generators as gen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nres = gen.grid_resistor_count(N)
for src, rscale in ((1e37, 1.0), (1e39, 1.0), (1e60, 1.0), (1e-36, 1.0), (1e-40, 1.0), (1e-60, 1.0), (1.0, 1.0), (1e-15, 1.0), (1e-25, 1.0), (1e-32, 1.0), (1e20, 1.0), (1e30, 1.0), (1.0, 1e-9), (1.0, 1e9),
                    (1e-12, 1e9), (1e-30, 1e12), (1e25, 1e-12)):
    vals = np.full(nres, rscale)
    table = gen.grid_table(N, vals)
    table.value[-1] = src
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, rr = h.solve_sparse()
    print(f"source {src:8.0e} A, resistors {rscale:8.0e} ohm: info {info}, {iters} iterations, scaled residual {h.residual():.1e}, "
          f"max |x| {np.abs(x).max():.3e}, finite {bool(np.isfinite(x).all())}", flush=True)
    h.close()
