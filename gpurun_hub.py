import sys, time, numpy as np
sys.path.insert(0, '.')
from nodal_amd import _ffi, generators as gen
rng = np.random.default_rng(1)
side = 300
ga, gb, _ = gen._grid_arrays(side)
nn = side * side
for spokes_n in (0, 5000, 40000):
    hub = nn
    spokes = rng.choice(nn, spokes_n, replace=False) if spokes_n else np.zeros(0, dtype=np.int64)
    a = np.concatenate([ga, np.full(spokes_n, hub, dtype=np.int64), [0]])
    b = np.concatenate([gb, spokes.astype(np.int64), [nn + 1]])
    vals = rng.uniform(0.5, 2.0, len(a))
    table = gen.passive_table(a, b, vals, nn - 1, nn + 1)
    h = _ffi.Handle(0); h.upload(table); h.assemble_symbolic(); assert h.assemble_numeric()[0] == _ffi.OK
    for _ in range(2):
        t0 = time.perf_counter(); x, info, it, rr = h.solve_sparse(); dt = (time.perf_counter() - t0) * 1e3
    print(f"hub with {spokes_n} spokes: n={len(x)} info={info} iters={it} {dt:.1f} ms residual {h.residual():.1e}", flush=True)
    h.close()
