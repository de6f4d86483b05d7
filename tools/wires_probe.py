import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen
table = gen.grid_with_wires_table(40, 120)
h = _ffi.Handle(0)
h.upload(table); h.assemble_symbolic(); h.assemble_numeric()
x, info, iters, rr = h.solve_sparse()
print("info", info, "iters", iters, "resid", h.residual())
