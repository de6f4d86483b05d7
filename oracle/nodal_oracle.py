"""ORACLE -- CPU restatement of Nodal.py's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (nodal_amd) never does and has no CPU fallback.

It restates, function by function, what the reference does between a parsed
netlist and the solution vector (all citations into /root/reference):

  build_model(netlist, sparse) ... Circuit.build_model, nodal/nodal.py:338-398,
                                   dispatching to write_R/A/E/VCVS/CCVS/CCCS,
                                   nodal/models.py:13-214 (VCCS -> write_VCVS,
                                   nodal/nodal.py:377-378).
  solve(G, A, sparse) ............ Circuit.solve, nodal/nodal.py:323-327.

The arithmetic of the solve lives in third-party code that is not under
/root/reference: numpy `np.linalg.solve` (LAPACK dgesv, OpenBLAS) and scipy
`scipy.sparse.linalg.spsolve` (SuperLU gssv, COLAMD).  scipy is un-pinned by the
reference (pyproject.toml:11); the versions present here are numpy 2.2.6 /
scipy 1.15.3.  The oracle calls exactly those two library functions, as the
reference does, so it inherits their algorithm instead of re-deriving it.

PINNING: tests/test_oracle.py checks this module against golden vectors
produced by running the reference itself in the build container
(tests/golden/make_golden.py -> tests/golden/*.json): G, A, currents, dense and
sparse solutions, and every error path, for all eleven doc/*.csv examples, 38
edge cases and the synthetic grids.  Parity of the sparse path is pinned by
those generated vectors only: the reference's own tests never exercise it.

`assemble_fast` is a vectorised restatement for netlists too large for the
per-component Python loop; tests pin it against `build_model`.
"""

import warnings

import numpy as np

ANOM = ("E", "VCVS", "VCCS", "CCVS", "CCCS")


class _Dok:
    """The subset of scipy.sparse.dok_matrix semantics the stamps rely on:
    missing entries read as 0.0 and an entry set to 0 is dropped."""

    def __init__(self, n):
        self.n = n
        self.d = {}

    def __getitem__(self, key):
        return self.d.get(key, 0.0)

    def __setitem__(self, key, value):
        if value:
            self.d[key] = float(value)
        elif key in self.d:
            del self.d[key]

    def tocsr(self):
        import scipy.sparse as spsp
        if not self.d:
            return spsp.csr_matrix((self.n, self.n), dtype=np.float64)
        keys = np.array(list(self.d.keys()), dtype=np.int64)
        vals = np.array(list(self.d.values()), dtype=np.float64)
        return spsp.csr_matrix((vals, (keys[:, 0], keys[:, 1])), shape=(self.n, self.n))


def build_model(netlist, sparse=False):
    """Returns [G, A, currents] exactly as Circuit.build_model does."""
    nums, anomnum = netlist.nums, netlist.anomnum
    comps, ground, nodenum = netlist.components, netlist.ground, netlist.nodenum
    K = nums["kcl"]
    n = K + nums["be"]
    G = _Dok(n) if sparse else np.zeros((n, n))
    A = np.zeros(n)
    currents = []

    def idx(label):
        return nodenum[label]

    for key in netlist.component_keys:  # file order (nodal/nodal.py:357)
        comp = comps[key]
        i = idx(comp.anode) if comp.anode != ground else None
        j = idx(comp.bnode) if comp.bnode != ground else None
        t = comp.type
        if t == "R":  # models.py:13-24
            try:
                g = 1 / comp.value
            except ZeroDivisionError:
                raise ValueError("Model error: resistors can't have null resistance")
            if i is not None:
                G[i, i] += g
            if j is not None:
                G[j, j] += g
            if i is not None and j is not None:
                G[i, j] -= g
                G[j, i] -= g
        elif t == "A":  # models.py:27-32
            if i is not None:
                A[i] += comp.value
            if j is not None:
                A[j] -= comp.value
        elif t == "E":  # models.py:35-50
            m = K + anomnum[comp.name]
            currents.append(comp.name)
            A[m] += comp.value
            _incidence(G, m, i, j, check=True)
        elif t in ("VCVS", "VCCS"):  # models.py:53-78 (both, nodal.py:377-380)
            currents.append(comp.name)
            m = K + anomnum[comp.name]
            _incidence(G, m, i, j, check=True)
            if comp.pos_control != ground:
                G[m, idx(comp.pos_control)] += -comp.value
            if comp.neg_control != ground:
                G[m, idx(comp.neg_control)] += comp.value
        elif t == "CCVS":  # models.py:109-158
            m = K + anomnum[comp.name]
            currents.append(comp.name)
            driver = _driver(comps, comp)
            _check_control(comp, driver)
            _incidence(G, m, i, j, check=False)
            if driver.type == "R":
                if comp.pos_control != ground:
                    G[m, idx(comp.pos_control)] = comp.value / driver.value
                if comp.neg_control != ground:
                    G[m, idx(comp.neg_control)] = -comp.value / driver.value
            else:
                _non_resistor_driver()
        elif t == "CCCS":  # models.py:161-214
            currents.append(comp.name)
            m = K + anomnum[comp.name]
            if i is not None:
                assert G[i, m] == 0
                G[i, m] = -1
            if j is not None:
                assert G[j, m] == 0
                G[j, m] = 1
            assert G[m, m] == 0
            G[m, m] = 1
            driver = _driver(comps, comp)
            if driver.type == "R":
                _check_control(comp, driver)
                if comp.pos_control != ground:
                    col = idx(comp.pos_control)
                    assert G[m, col] == 0
                    G[m, col] = +comp.value / driver.value
                if comp.neg_control != ground:
                    col = idx(comp.neg_control)
                    assert G[m, col] == 0
                    G[m, col] = -comp.value / driver.value
            else:
                _non_resistor_driver()
        elif t == "OPAMP":
            raise NotImplementedError
        else:
            raise ValueError(f"Unknown component type: {t}")
    if sparse:
        G = G.tocsr()
    return [G, A, currents]


def _incidence(G, m, i, j, check):
    """+-1 incidence entries of a branch equation (models.py:41-50, 64-72, 126-133)."""
    if i is not None:
        if check:
            assert G[m, i] == 0
        G[m, i] = 1
        G[i, m] = -1
    if j is not None:
        if check:
            assert G[m, j] == 0
        G[m, j] = -1
        G[j, m] = 1


def _driver(comps, comp):
    try:
        return comps[comp.driver]
    except KeyError:
        raise KeyError(f"Driving component {comp.driver} not found")


def _check_control(comp, driver):
    assert comp.pos_control is not None and comp.neg_control is not None
    assert (comp.pos_control == driver.anode and comp.neg_control == driver.bnode) or (
        comp.pos_control == driver.bnode and comp.neg_control == driver.anode)


def _non_resistor_driver():
    # models.py:146,200 read `c.NODE_TYPES_ANOM` off the Component argument
    raise AttributeError("'Component' object has no attribute 'NODE_TYPES_ANOM'")


def solve(G, A, sparse=False):
    """The two library calls of Circuit.solve (nodal/nodal.py:325,327).
    Returns (x, warning class names); raises numpy.linalg.LinAlgError like the
    dense reference path."""
    if sparse:
        import scipy.sparse.linalg as spspla
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            x = spspla.spsolve(G, A)
        return np.asarray(x, dtype=float), sorted({type(i.message).__name__ for i in w})
    return np.linalg.solve(G, A), []


def solve_netlist(netlist, sparse=False):
    G, A, currents = build_model(netlist, sparse)
    x, warns = solve(G, A, sparse)
    return x, G, A, currents, warns


# ---------------------------------------------------------------------------
# vectorised restatement for large tables (same summation order)
# ---------------------------------------------------------------------------

def assemble_fast(table):
    """CSR (scipy) and rhs from a lowered component table, for netlists with
    millions of components.  Accumulating entries are summed strictly in
    component order (k-th contribution of every entry added in the k-th pass),
    like the reference's sequential `+=`.  Supports tables in which the SET
    stamps of E / VCVS / CCVS / CCCS rows never coincide with another stamp
    (asserted), which holds for all synthetic benchmark netlists."""
    import scipy.sparse as spsp
    T, v = table.type.astype(np.int64), table.value
    a, b, cc, dd = (getattr(table, x).astype(np.int64) for x in "abcd")
    K, n = table.K, table.n
    m = K + table.k.astype(np.int64)
    comp = np.arange(table.ncomp, dtype=np.int64)
    rows, cols, vals, order, is_set = [], [], [], [], []

    def emit(mask, r, c_, val, slot, setflag):
        if mask.any():
            rows.append(r[mask]); cols.append(c_[mask])
            vals.append(np.broadcast_to(val, mask.shape)[mask])
            order.append(comp[mask] * 8 + slot)
            is_set.append(np.full(int(mask.sum()), setflag))

    isR = T == 0
    with np.errstate(divide="ignore"):
        g = np.where(isR, 1.0 / np.where(isR, v, 1.0), 0.0)
    emit(isR & (a >= 0), a, a, g, 0, False)
    emit(isR & (b >= 0), b, b, g, 1, False)
    both = isR & (a >= 0) & (b >= 0)
    emit(both, a, b, -g, 2, False)
    emit(both, b, a, -g, 3, False)
    branch = (T == 2) | (T == 3) | (T == 4)
    emit(branch & (a >= 0), m, a, 1.0, 0, True)
    emit(branch & (a >= 0), a, m, -1.0, 1, True)
    emit(branch & (b >= 0), m, b, -1.0, 2, True)
    emit(branch & (b >= 0), b, m, 1.0, 3, True)
    vc = T == 3
    emit(vc & (cc >= 0), m, cc, -v, 4, False)
    emit(vc & (dd >= 0), m, dd, v, 5, False)
    drv = table.drv.astype(np.int64)
    Rd = np.where(drv >= 0, v[np.maximum(drv, 0)], 1.0)
    cv = T == 4
    emit(cv & (cc >= 0), m, cc, v / Rd, 4, True)
    emit(cv & (dd >= 0), m, dd, (-v) / Rd, 5, True)
    cs = T == 5
    emit(cs & (a >= 0), a, m, -1.0, 0, True)
    emit(cs & (b >= 0), b, m, 1.0, 1, True)
    emit(cs, m, m, 1.0, 2, True)
    emit(cs & (cc >= 0), m, cc, v / Rd, 3, True)
    emit(cs & (dd >= 0), m, dd, (-v) / Rd, 4, True)

    A = np.zeros(n)
    isA, isE = T == 1, T == 2
    # rhs: sequential accumulation in component order
    ra = np.concatenate([a[isA & (a >= 0)], b[isA & (b >= 0)], m[isE]])
    va = np.concatenate([v[isA & (a >= 0)], -v[isA & (b >= 0)], v[isE]])
    oa = np.concatenate([comp[isA & (a >= 0)] * 2, comp[isA & (b >= 0)] * 2 + 1, comp[isE] * 2])
    if len(ra):
        rr, _, sv = _ordered_segment_sum(ra, np.zeros_like(ra), va, oa, n, 1)
        A[rr] = sv
    if not rows:
        return spsp.csr_matrix((n, n)), A
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals, order = np.concatenate(vals).astype(float), np.concatenate(order)
    is_set = np.concatenate(is_set)
    er, ec, ev = _ordered_segment_sum(rows, cols, vals, order, n, n, is_set)
    G = spsp.csr_matrix((ev, (er, ec)), shape=(n, n))
    return G, A


def _ordered_segment_sum(rows, cols, vals, order, nrows, ncols, is_set=None):
    key = rows * ncols + cols
    perm = np.lexsort((order, key))
    key, vals = key[perm], vals[perm]
    head = np.ones(len(key), dtype=bool)
    head[1:] = key[1:] != key[:-1]
    seg = np.cumsum(head) - 1
    starts = np.flatnonzero(head)
    rank = np.arange(len(key)) - starts[seg]
    if is_set is not None:
        s = is_set[perm]
        seglen = np.diff(np.append(starts, len(key)))
        # SET stamps must be alone in their entry for this fast path
        assert (seglen[seg[s]] == 1).all(), "SET stamp coincides with another stamp"
    out = np.zeros(len(starts))
    for r in range(int(rank.max()) + 1 if len(rank) else 0):
        sel = rank == r
        out[seg[sel]] += vals[sel]
    ukey = key[starts]
    return ukey // ncols, ukey % ncols, out
