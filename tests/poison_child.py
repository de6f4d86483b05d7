"""Child process of tests/test_gpu_parity.py::test_reused_handle_with_every_scratch_buffer_poisoned.

Run with NODAL_POISON=2 in the environment (csrc/ctx.h): ONE device handle, singular and regular systems
of changing sizes through every solver path, each regular answer compared with the oracle.  Prints
"poison child ok" and exits 0 when all of them agree."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import nodal_amd as n  # noqa: E402
from nodal_amd import _ffi, generators as gen  # noqa: E402
from nodal_amd.lowering import lower  # noqa: E402
from oracle import nodal_oracle as oracle  # noqa: E402

TOL = 1e-9


def normwise(x, ref):
    scale = np.abs(ref).max()
    return np.abs(x - ref).max() / (scale if scale > 0 else 1.0)


def general_rows(N):
    rows = list(gen.grid_rows(N))[:-1]
    rows.append(["e1", "E", "5", "1", "g"])
    rows.append(["rv", "R", "3", "v1", "2"])
    rows.append(["d1", "VCVS", "0.5", "v1", "g", "3", "4"])
    return rows


def island(prefix, count):
    rows = [[f"{prefix}{i}", "R", "1", f"{prefix}n{i}", f"{prefix}n{i + 1}"] for i in range(count)]
    rows.append([f"{prefix}a", "A", "1", f"{prefix}n3", f"{prefix}n{count - 2}"])
    return rows


def main():
    assert os.environ.get("NODAL_POISON") == "2"
    rng = np.random.default_rng(21)
    graded = 10.0 ** rng.uniform(-1.5, 1.5, gen.grid_resistor_count(80))
    T = lambda rows: lower(n.Netlist.from_rows(rows))  # noqa: E731
    # (table, dense?, singular?)
    jobs = [
        (gen.grid_table(120), False, False),                                   # smoothed aggregation
        (T(list(gen.grid_rows(90)) + island("x", 300)), False, True),          # floating chain (low-degree elimination)
        (gen.grid_table(70), False, False),
        (gen.grid_table(80, graded), False, False),                            # plain aggregation, contrast mode
        (gen.cfg5_table(90), False, False),                                    # presolve + FGMRES
        (T(general_rows(85) + island("x", 150)), False, True),                 # general, singular
        (gen.cfg5_table(72), False, False),
        (gen.ladder_table(9000), False, False),                                # exact elimination
        (gen.ladder_table(7000, island=50), False, True),
        (gen.ladder_table(3000), False, False),
        (gen.grid_table(45), True, False),                                     # dense block elimination
        (T(list(gen.grid_rows(40)) + island("x", 40)), True, True),
        (gen.grid_table(31), True, False),
        (gen.cfg5_table(34), True, False),                                     # dense, presolve / pivoted LU
        (gen.cfg5_table(20), True, False),
        (gen.grid_table(130), False, False),
    ]
    h = _ffi.Handle(0)
    for k, (t, dense, singular) in enumerate(jobs):
        h.upload(t)
        for rep in range(2):
            info = h.run(dense)
            if singular:
                assert info > 0, (k, rep, info)
                continue
            assert info == 0, (k, rep, info)
            Go, Ao = oracle.assemble_fast(t)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                xo, _ = oracle.solve(Go.toarray() if dense else Go.tocsr(), Ao, not dense)
            err = normwise(h.download_x(), xo)
            assert err <= TOL, (k, rep, err)
            assert h.residual() <= 1e-12, (k, rep)
    # equivalent-resistance sweep and a value-sweep batch on the same handle
    t = gen.grid_table(60)
    h.upload(t)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    ia = np.array([0, 5, 100, 900], dtype=np.int32)
    ib = np.array([-1, 77, 3000, 901], dtype=np.int32)
    R, info = h.solve_pairs(ia, ib, dense=False)
    assert info == 0 and np.isfinite(R).all() and (R > 0).all()
    Go, Ao = oracle.assemble_fast(t)
    Gd = Go.toarray()
    for q in range(len(ia)):
        b = np.zeros(t.n)
        b[ia[q]] = 1.0
        if ib[q] >= 0:
            b[ib[q]] = -1.0
        e = np.linalg.solve(Gd, b)
        want = e[ia[q]] - (e[ib[q]] if ib[q] >= 0 else 0.0)
        assert abs(R[q] - want) <= TOL * abs(want), (q, R[q], want)
    vals = np.ones((9, t.ncomp))
    for b in range(9):
        vals[b, :-1] = gen.cfg4_values(b, 60)
    h.upload_values(vals)
    x, info = h.run_batch(0, 9)
    assert not np.any(info)
    for b in (0, 8):
        tb = t.truncated(t.ncomp)
        tb.value[:] = vals[b]
        Gb, Ab = oracle.assemble_fast(tb)
        assert normwise(x[b], oracle.solve(Gb.tocsr(), Ab, True)[0]) <= TOL, b
    h.close()
    print("poison child ok")


if __name__ == "__main__":
    main()
