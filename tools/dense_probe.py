"""cfg2 probe: grid(N) through the dense path (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h = _ffi.Handle(0)
h.upload(gen.grid_table(N))
for r in range(reps):
    t0 = time.perf_counter(); info = h.run(True); h.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"run {r}: {dt:.2f} ms wall, phases {['%.2f' % t for t in h.timings()]}, info {info}, kernel {h.kernel_stats()}")
x = h.download_x()
print("residual", h.residual(), "x0", x[0])
