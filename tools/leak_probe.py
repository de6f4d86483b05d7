"""Device memory across many handle lifetimes (every solver family once per lifetime): free memory must come
back.   python tools/leak_probe.py [rounds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nodal_amd import _ffi, generators as gen

hip = ctypes.CDLL("libamdhip64.so")


def free_mib():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value / 2**20


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tables = [("grid(300) sparse", gen.grid_table(300), False), ("cfg5(200) sparse", gen.cfg5_table(200), False),
          ("grid(60) dense", gen.grid_table(60), True), ("cfg5(20) dense", gen.cfg5_table(20), True)]
h0 = _ffi.Handle(0)  # (the runtime's own allocations happen with the first handle)
h0.close()
start = free_mib()
for r in range(rounds):
    for name, table, dense in tables:
        h = _ffi.Handle(0)
        h.upload(table)
        assert h.run(dense) == 0
        assert h.run(dense, 0, True) == 0
        h.close()
    print(f"round {r}: free device memory {free_mib():.0f} MiB ({free_mib() - start:+.0f} since the start)", flush=True)
assert abs(free_mib() - start) < 64, "device memory did not come back"
print("ok")
