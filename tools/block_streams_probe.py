"""Config 3's circuit as value-sweep members of block-diagonal systems (nodal_run_batch), several block systems in
flight: S threads x one BatchSolver each x k members per block.  python tools/block_streams_probe.py S k [reps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import generators as gen
from nodal_amd.batch import BatchSolver

S, k = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = int(os.environ.get("PROBE_N", "1000"))
table = gen.grid_table(N)
nres = gen.grid_resistor_count(N)
rng = np.random.RandomState(4)
if S >= 4 and "NODAL_STREAM_PRIORITY" not in os.environ:
    os.environ["NODAL_STREAM_PRIORITY"] = "normal"
solvers = []
for s in range(S):
    vals = np.ones((k, table.ncomp))
    vals[:, :nres] = rng.uniform(0.5, 2.0, size=(k, nres))
    b = BatchSolver(table, 0)
    b.upload_values(vals)
    b.run(sparse=True, reuse_symbolic=False, download=False)  # warm-up: buffers grow
    b.h.synchronize()
    solvers.append(b)
for reuse in (False, True):
    def work(b):
        for _ in range(reps):
            b.run(sparse=True, reuse_symbolic=reuse, download=False)
        b.h.synchronize()
    th = [threading.Thread(target=work, args=(b,)) for b in solvers]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    n = S * k * reps
    it, lev, rr = solvers[0].h.solve_info()
    print(f"{S} streams x {k} members x {reps} reps, symbolic kept {reuse}: {n / dt:7.1f} circuits/s ({dt / n * 1e3:.2f} ms per circuit), "
          f"{it} iterations, residual {solvers[0].h.residual():.1e}", flush=True)
for b in solvers:
    b.close()
