"""Where the first solve on a fresh handle spends its time: python tools/first_solve_probe.py [case]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi
import tools.topologies as topo

name = sys.argv[1] if len(sys.argv) > 1 else "ladder1e5"
table = topo.CASES[name]()
for rep in range(3):
    t0 = time.perf_counter(); h = _ffi.Handle(0); t1 = time.perf_counter()
    h.upload(table); t2 = time.perf_counter()
    h.assemble_symbolic(); h.assemble_numeric(); t3 = time.perf_counter()
    x, info, iters, rr = h.solve_sparse(); t4 = time.perf_counter()
    x, info, iters, rr = h.solve_sparse(); t5 = time.perf_counter()
    h.close(); t6 = time.perf_counter()
    print(f"{name} handle {rep}: create {1e3*(t1-t0):.2f} upload {1e3*(t2-t1):.2f} assemble {1e3*(t3-t2):.2f} "
          f"first solve {1e3*(t4-t3):.2f} second {1e3*(t5-t4):.2f} close {1e3*(t6-t5):.2f} ms", flush=True)
