"""Config 3's circuit as members of ONE block-diagonal system (nodal_run_batch, the path of config 4): k value
sets of grid(1000) per launch sequence -- the coarse levels' launch floor is paid once per k circuits.
python tools/block_big_probe.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import generators as gen
from nodal_amd.batch import BatchSolver

N = int(os.environ.get("PROBE_N", "1000"))
table = gen.grid_table(N)
nres = gen.grid_resistor_count(N)
rng = np.random.RandomState(4)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    vals = np.ones((k, table.ncomp))
    vals[:, :nres] = rng.uniform(0.5, 2.0, size=(k, nres))
    with BatchSolver(table, 0) as s:
        s.upload_values(vals)
        for reuse in (False, False, True):
            t0 = time.perf_counter()
            s.run(sparse=True, reuse_symbolic=reuse, download=False)
            s.h.synchronize()
            dt = time.perf_counter() - t0
            it, lev, rr = s.h.solve_info()
            print(f"grid({N}) x {k} members, symbolic kept {reuse}: {dt * 1e3:8.2f} ms = {dt / k * 1e3:6.2f} ms per circuit "
                  f"({k / dt:6.1f} circuits/s), {it} iterations, {lev} levels, timings {np.round(s.h.timings(), 3)}", flush=True)
