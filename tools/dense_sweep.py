"""Development probe: the dense passive path on grids (banded) and on random graphs (no band:
every tile of the block elimination carries data) over a range of sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen


def random_graph_table(n, deg, seed):
    rng = np.random.default_rng(seed)
    a = np.concatenate([np.arange(n - 1), rng.integers(0, n, n * deg // 2)])
    b = np.concatenate([np.arange(1, n), rng.integers(0, n, n * deg // 2)])
    keep = a != b
    a, b = a[keep], b[keep]
    return gen.passive_table(a, b, rng.uniform(0.5, 2.0, a.size), 0, n - 1)


bad = []
for spec in sys.argv[1:] or ["r300", "r520", "r700", "r1100", "r1898", "r3000", "r6000", "g40", "g64"]:
    n = int(spec[1:])
    table = random_graph_table(n, 6, n) if spec[0] == "r" else gen.grid_table(n)
    h = _ffi.Handle(0)
    h.upload(table); h.assemble_symbolic(); h.assemble_numeric()
    x, info = h.solve_dense()
    r = h.residual()
    print(spec, h.n, info, "%.2e" % r, flush=True)
    if not (r <= 1e-13): bad.append((spec, h.n, r))
    h.close()
print("bad", bad)
