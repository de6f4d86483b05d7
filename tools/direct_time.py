"""Repeated direct solve of cfg5(N) (analysis kept): wall time of nodal_solve_sparse(NODAL_SPARSE_DIRECT).
python tools/direct_time.py [N]   (NODAL_TRACE=1 adds the phases)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
table = gen.cfg5_table(N)
h = _ffi.Handle(0)
h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)
h.upload(table); h.assemble_symbolic(); h.assemble_numeric()
ms = []
for rep in range(4):
    t0 = time.perf_counter(); x, info, iters, rr = h.solve_sparse(method=_ffi.SPARSE_DIRECT, download=False); h.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
print(f"cfg5({N}): first {ms[0]:.1f} ms, repeated {min(ms[1:]):.1f} ms, info {info}, iterations {iters}, residual {h.residual():.2e}")
h.close()
