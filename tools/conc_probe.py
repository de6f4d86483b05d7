"""Throughput of C concurrent grid(1000) solves (one handle + host thread each): development aid."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen
C = int(sys.argv[1]); per = int(sys.argv[2]) if len(sys.argv) > 2 else 16
table = gen.grid_table(1000)
hs = []
for _ in range(C):
    h = _ffi.Handle(0); h.upload(table); h.run(False); hs.append(h)
def work(h):
    for _ in range(per):
        assert h.run(False) == 0
for rep in range(2):
    ts = [threading.Thread(target=work, args=(h,)) for h in hs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"C={C}: {C*per/dt:.1f} circuits/s, {dt/(C*per)*1e3:.2f} ms per circuit (throughput), {dt/per*1e3:.2f} ms per solve (latency)")
