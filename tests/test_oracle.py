"""Pins the oracle (oracle/nodal_oracle.py) against golden vectors produced by
the reference itself (tests/golden/make_golden.py)."""
import io

import numpy as np
import pytest

import nodal_amd as n
from nodal_amd import generators as gen
from nodal_amd.lowering import lower
from oracle import nodal_oracle as oracle
from tests.conftest import load_golden

CASES = [c for c in load_golden("cases.json") if "parse_error" not in c]
SYNTH = load_golden("synth.json")
EXC = {"ValueError": ValueError, "KeyError": KeyError, "AssertionError": AssertionError,
       "AttributeError": AttributeError, "NotImplementedError": NotImplementedError,
       "LinAlgError": np.linalg.LinAlgError, "ZeroDivisionError": ZeroDivisionError}


def parse(case):
    if case.get("raw_text") is not None:
        import csv
        return n.Netlist.from_rows(csv.reader(io.StringIO(case["raw_text"]), skipinitialspace=True))
    return n.Netlist.from_rows(case["rows"])


def normwise(x, ref):
    x, ref = np.asarray(x, float), np.asarray(ref, float)
    scale = np.abs(ref).max()
    return np.abs(x - ref).max() / (scale if scale > 0 else 1.0)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_oracle_matches_reference(case, mode):
    nl = parse(case)
    want = case[mode]
    sparse = mode == "sparse"
    if "error" in want and want["error"]["type"] not in ("LinAlgError", "UnconnectedCircuitError"):
        with pytest.raises(EXC[want["error"]["type"]]) as info:
            oracle.build_model(nl, sparse)
        assert [str(a) for a in info.value.args] == want["error"]["args"]
        return
    G, A, currents = oracle.build_model(nl, sparse)
    assert currents == case["currents"]
    assert A.tolist() == case["A"]  # bit-exact
    Gd = G.toarray() if sparse else G
    ii, jj, vv = case["G_coo"]
    ref = np.zeros_like(Gd)
    ref[ii, jj] = vv
    assert np.array_equal(Gd, ref)  # bit-exact
    if sparse:
        assert G.nnz == case["nnz_sparse"]
    if "error" in want:  # singular dense system
        with pytest.raises(np.linalg.LinAlgError):
            oracle.solve(G, A, sparse)
        return
    x, warns = oracle.solve(G, A, sparse)
    assert warns == want["warnings"]
    ref_x = np.array(want["x"])
    if np.isnan(ref_x).any():
        assert np.isnan(x).all()
    else:
        assert normwise(x, ref_x) <= 1e-12


def rows_of(genspec):
    kind = genspec[0]
    if kind == "grid":
        return list(gen.grid_rows(genspec[1]))
    if kind == "cfg4":
        return list(gen.grid_rows(genspec[1], gen.cfg4_values(genspec[2], genspec[1])))
    return gen.cfg5_rows(genspec[1], genspec[2])


@pytest.mark.parametrize("case", SYNTH, ids=[c["name"] for c in SYNTH])
def test_oracle_fast_assembly_and_solve(case):
    nl = n.Netlist.from_rows(rows_of(case["gen"]))
    assert nl.ground == case["ground"] and nl.nums == case["nums"]
    assert [list(i) for i in list(nl.nodenum.items())[:8]] == case["nodenum_head"]
    assert [list(i) for i in list(nl.nodenum.items())[-8:]] == case["nodenum_tail"]
    table = lower(nl)
    assert table.first_error is None
    G, A = oracle.assemble_fast(table)
    G.eliminate_zeros()
    assert G.nnz == case["nnz"]
    assert float(np.abs(G.data).sum()) == case["G_abs_sum"]
    assert float(A.sum()) == case["A_sum"]
    if "csr" in case:  # full matrices stored for the small ones: bit-exact
        import scipy.sparse as spsp
        indptr, indices, data = case["csr"]
        ref = spsp.csr_matrix((data, indices, indptr), shape=G.shape)
        assert (abs(G - ref)).nnz == 0
        assert A.tolist() == case["A"]
        if table.n <= 300:  # the per-component restatement agrees with the fast one
            Gs, As, _ = oracle.build_model(nl, True)
            assert (abs(Gs - G)).nnz == 0 and As.tolist() == A.tolist()
    x, _ = oracle.solve(G.tocsr(), A, True)
    idx = case["x_idx"]
    assert normwise(x[idx], case["x_sparse_samples"]) <= 1e-11
    assert abs(np.abs(x).max() - case["x_sparse_absmax"]) <= 1e-11 * case["x_sparse_absmax"]
