"""Config 4's per-GPU shard (128 x grid(100)) as ONE block system against k block systems of 128 / k members
solved concurrently (one handle, stream and host thread each): python tools/batch_conc_probe.py [k ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import generators as gen
from nodal_amd.batch import BatchSolver

table = gen.grid_table(100)
M = 128
vals = np.ones((M, table.ncomp))
for i in range(M):
    vals[i, :-1] = gen.cfg4_values(i, 100)
for k in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
    if k >= 4:
        os.environ["NODAL_STREAM_PRIORITY"] = "normal"
    per = M // k
    solvers = []
    for j in range(k):
        s = BatchSolver(table, 0)
        s.upload_values(vals[j * per:(j + 1) * per])
        s.run(sparse=True, download=False)
        solvers.append(s)
    for reuse in (False, True):
        reps = 6

        def work(s):
            for _ in range(reps):
                s.run(sparse=True, reuse_symbolic=reuse, download=False)

        threads = [threading.Thread(target=work, args=(s,)) for s in solvers]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in solvers:
            s.h.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{k} block system(s) of {per} members, symbolic phases kept {reuse}: {dt * 1e3:.2f} ms per 128 circuits = "
              f"{M / dt:.0f} circuits/s", flush=True)
    for s in solvers:
        s.close()
