"""Equivalent-resistance sweep on the sparse path: time per pair, block iteration
(csrc/sagg_multi.h, 16 pairs per launch sequence) against one solve per pair.
python tools/pairs_probe.py N npairs"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 316
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if len(sys.argv) > 3:  # child: one mode
    from nodal_amd import _ffi, generators as gen
    table = gen.grid_table(N)
    rng = np.random.RandomState(3)
    ia = rng.randint(0, table.K, size=npairs).astype(np.int32)
    ib = rng.randint(-1, table.K, size=npairs).astype(np.int32)
    ib[ib == ia] = -1
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    h.assemble_numeric()
    for rep in range(2):
        t0 = time.perf_counter()
        res, info = h.solve_pairs(ia, ib, False)
        dt = time.perf_counter() - t0
        print(f"  grid({N}) [{sys.argv[3]}]: {npairs} pairs in {dt * 1e3:.1f} ms = {dt / npairs * 1e3:.2f} ms per pair "
              f"(info {info}, R[0] = {res[0]:.9f})", flush=True)
    np.save(f"/tmp/pairs_{sys.argv[3]}.npy", res)
    h.close()
else:
    modes = [("block", {"NODAL_PAIRS_DIRECT": "0"}), ("direct", {"NODAL_PAIRS_DIRECT": "1"})]
    if npairs <= 64:
        modes.append(("single", {"NODAL_PAIRS_BLOCK": "0", "NODAL_PAIRS_DIRECT": "0"}))
    for mode, env in modes:
        subprocess.run([sys.executable, __file__, str(N), str(npairs), mode], env=dict(os.environ, **env), check=True)
    a, d = np.load("/tmp/pairs_block.npy"), np.load("/tmp/pairs_direct.npy")
    print(f"  block vs factor-once (sparse LU, 16 columns per substitution): max relative difference "
          f"{np.abs(a - d).max() / np.abs(d).max():.2e}")
    if npairs <= 64:
        b = np.load("/tmp/pairs_single.npy")
        print(f"  block vs single: max relative difference {np.abs(a - b).max() / np.abs(b).max():.2e}")
