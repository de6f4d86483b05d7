"""Child process of tests/test_gpu_parity.py::test_schedule_variants_give_identical_bits.

Solves a fixed set of systems through the routes whose round-5 variants claim to change the schedule only -- columns as
16-bit offsets from the row at level 0 (NODAL_SA_D16), direction update and outer SpMV in one launch
(NODAL_SA_FUSE_DIR), the wide fronts of the direct route substituted in super steps (NODAL_DIRECT_SUPER) -- and prints
one SHA-256 over the bytes of every solution.  The parent runs it under each setting and compares the digests."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from nodal_amd import _ffi, generators as gen  # noqa: E402


def main():
    os.environ["NODAL_PAIRS_DIRECT"] = "1"  # (the pair sweep below through the sparse LU: sixteen columns per substitution)
    h = hashlib.sha256()
    rng = np.random.default_rng(5)
    # smoothed aggregation + flexible CG: fixed-width rows (grid) and rows with their lengths (a grid with diagonal
    # chords at every seventh node: rows of 5 to 9 entries, too uneven to pad)
    N = 260
    chords = list(gen.grid_rows(N))[:-1]
    for k in range(0, N * N - 2 * N - 3, 7):
        if (k % N) + 2 < N:
            chords.append([f"rc{k}", "R", "2", gen._label(k, N * N - 1), gen._label(k + 2 * N + 2, N * N - 1)])
            chords.append([f"rd{k}", "R", "3", gen._label(k, N * N - 1), gen._label(k + N + 2, N * N - 1)])
    chords.append(["a1", "A", "1", "1", "g"])
    from nodal_amd.lowering import lower
    from nodal_amd.netlist import Netlist
    for table in (gen.grid_table(200), lower(Netlist.from_rows(chords))):
        s = _ffi.Handle(0)
        s.upload(table)
        assert s.run(False) == 0
        h.update(s.download_x().tobytes())
        s.close()
    # presolve + FGMRES (the same level-0 kernels under a Krylov method of the general path)
    s = _ffi.Handle(0)
    s.upload(gen.cfg5_table(200))
    assert s.run(False) == 0
    h.update(s.download_x().tobytes())
    s.close()
    # direct route with fronts wide enough for the stepped substitution, one and sixteen right-hand sides
    s = _ffi.Handle(0)
    s.set_option(_ffi.OPT_EXTRA_STREAMS, 1)
    s.upload(gen.cfg5_table(420))
    s.assemble_symbolic()
    s.assemble_numeric()
    x, info, _it, _rr = s.solve_sparse(method=_ffi.SPARSE_DIRECT)
    assert info == 0
    h.update(np.asarray(x).tobytes())
    s.close()
    s = _ffi.Handle(0)
    table = gen.grid_table(420)
    s.upload(table)
    s.assemble_symbolic()
    s.assemble_numeric()
    ia = rng.integers(0, table.K, size=40).astype(np.int32)
    ib = rng.integers(0, table.K, size=40).astype(np.int32)
    ib[ib == ia] = -1
    res, info = s.solve_pairs(ia, ib, False)
    assert info == 0
    h.update(np.asarray(res).tobytes())
    s.close()
    print("variant digest", h.hexdigest())


if __name__ == "__main__":
    main()
