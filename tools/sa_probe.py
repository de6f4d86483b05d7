"""Quick probe of the sparse SPD path on the GPU box: grid(N) / cfg4 block, timings and
agreement with the golden samples (development aid, not part of the test suite)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nodal_amd import _ffi, generators as gen

def golden(name):
    for fn in ("synth_large.json", "synth.json"):
        d = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", fn)))
        for c in d["cases"]:
            if c["name"] == name:
                return c
    return None

def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "1000"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reuse = len(sys.argv) > 3 and sys.argv[3] == "reuse"  # symbolic phases kept from the first run on
    if what.startswith("cfg5"):
        N = int(what[5:] or 1000) if ":" in what else 1000
        table = gen.cfg5_table(N)
        gname = f"cfg5({N})"
    else:
        N = int(what)
        table = gen.grid_table(N)
        gname = f"grid({N})"
    h = _ffi.Handle(0)
    h.upload(table)
    for r in range(reps):
        t0 = time.perf_counter()
        info = h.run(False, 0, reuse and r > 0)
        h.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        it, lv, rr = h.solve_info()
        print(f"run {r}: {dt:.2f} ms wall, phases {['%.2f' % t for t in h.timings()]}, info {info}, "
              f"iterations {it}, levels {lv}, relres {rr:.2e}, kernel {h.kernel_stats()}")
    x = h.download_x()
    print("residual", h.residual(), "x0", x[0])
    g = golden(gname)
    if g:
        idx = np.array(g["x_idx"]); ref = np.array(g["x_sparse_samples"])
        print("normwise error vs reference samples", np.abs(x[idx] - ref).max() / g["x_sparse_absmax"])
    h.close()

if __name__ == "__main__":
    main()
