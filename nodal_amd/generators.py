"""Deterministic synthetic netlists for the benchmark configurations.

The reference ships no generator; SURVEY.md section 8(d) defines these and
validated them against the reference at N <= 1000.  Every generator yields
CSV rows (lists of str) so the same rows can be fed to the reference (through
a temporary .csv) and to `Netlist.from_rows`.

  grid(N)  : N x N resistor grid, node k = r*N + c labelled str(k+1) except the
             last node, labelled "g"; rows `rh{r}_{c}` (k,k+1) then `rv{r}_{c}`
             (k,k+N) in `for r: for c:` order; last row `a1,A,1,1,g`.
  cfg4     : grid(100) topology, member b draws one value per resistor,
             0.5 * 4**u with u = random.Random(1000+b).random().
  cfg5     : grid(N) plus ~1% E, 0.5% CCCS, 0.5% VCVS rows, seed 5.
"""

import random


def _label(k, last):
    return "g" if k == last else str(k + 1)


def grid_resistor_count(N):
    return 2 * N * (N - 1)


def grid_rows(N, values=None):
    """Rows of grid(N).  `values` optionally gives one resistance per resistor
    in row order (defaults to "1")."""
    last = N * N - 1
    i = 0
    for r in range(N):
        for col in range(N):
            k = r * N + col
            if col + 1 < N:
                val = "1" if values is None else repr(float(values[i]))
                i += 1
                yield [f"rh{r}_{col}", "R", val, _label(k, last), _label(k + 1, last)]
            if r + 1 < N:
                val = "1" if values is None else repr(float(values[i]))
                i += 1
                yield [f"rv{r}_{col}", "R", val, _label(k, last), _label(k + N, last)]
    yield ["a1", "A", "1", "1", "g"]


def cfg4_values(member, N=100):
    """Resistances of batch member `member`: log-uniform in [0.5, 2) ohm.
    Only `.random()` of the stdlib Mersenne Twister is used, for stability
    across Python versions."""
    rng = random.Random(1000 + member)
    return [0.5 * 4.0 ** rng.random() for _ in range(grid_resistor_count(N))]


def cfg5_rows(N, seed=5):
    """grid(N) with voltage sources and dependent sources sprinkled in.

    Scanning nodes in k order, two draws (u, w) are made for EVERY node; nodes
    in the last column and the last node get no extra rows.  The extra rows go
    after all grid rows and before `a1`."""
    last = N * N - 1
    base = list(grid_rows(N))
    source = base.pop()  # a1 goes last
    rng = random.Random(seed)
    extra = []
    for k in range(N * N):
        u = rng.random()
        w = rng.random()
        r, col = divmod(k, N)
        if k == last or col == N - 1:
            continue
        here, right = _label(k, last), _label(k + 1, last)
        if u < 0.01:
            extra.append([f"es{k}", "E", repr(-5 + 10 * w), f"s{k}", "g"])
            extra.append([f"rs{k}", "R", "1", f"s{k}", here])
        elif u < 0.015:
            extra.append(
                [f"fc{k}", "CCCS", repr(0.1 + 0.4 * w), here, "g", here, right,
                 f"rh{r}_{col}"]
            )
        elif u < 0.02:
            extra.append([f"dv{k}", "VCVS", repr(0.1 + 0.4 * w), f"v{k}", "g", here, right])
            extra.append([f"rd{k}", "R", "1", f"v{k}", here])
    return base + extra + [source]


def write_csv(rows, path):
    with open(path, "w") as out:
        for row in rows:
            out.write(",".join(row) + "\n")


# ---------------------------------------------------------------------------
# Direct (vectorised) construction of the lowered component table.
#
# Parsing 2e6 CSV rows through the string front-end takes tens of seconds in
# Python; the benchmarks need the same table in well under a second.  These
# builders restate the front-end's numbering rules on integer node ids
# (first-appearance order, anode before bnode, control nodes do not create
# nodes, ground = the node labelled "g") and are pinned against
# `lower(Netlist.from_rows(...))` by tests/test_generators.py.
# ---------------------------------------------------------------------------

def _number_nodes(a_ids, b_ids, ground_id):
    """id -> nodenum index (ground -> -1) by first appearance as a lead."""
    import numpy as np
    leads = np.empty(2 * len(a_ids), dtype=np.int64)
    leads[0::2], leads[1::2] = a_ids, b_ids
    uniq, first = np.unique(leads, return_index=True)
    uniq = uniq[np.argsort(first, kind="stable")]
    uniq = uniq[uniq != ground_id]
    index = np.full(int(leads.max()) + 1, -2, dtype=np.int64)
    index[uniq] = np.arange(len(uniq))
    index[ground_id] = -1
    return index, len(uniq)


def _table(types, values, a_ids, b_ids, c_ids, d_ids, drv_rows, ground_id):
    import numpy as np
    from .lowering import ComponentTable
    index, K = _number_nodes(a_ids, b_ids, ground_id)
    anom = types >= 2
    table = ComponentTable(len(types), K, int(anom.sum()))
    table.type[:] = types
    table.value[:] = values
    table.a[:] = index[a_ids]
    table.b[:] = index[b_ids]
    has_ctl = c_ids >= 0
    table.c[has_ctl] = index[c_ids[has_ctl]]
    table.d[has_ctl] = index[d_ids[has_ctl]]
    table.drv[:] = drv_rows
    table.k[anom] = np.arange(int(anom.sum()))
    assert (table.a >= -1).all() and (table.b >= -1).all()
    assert (table.c >= -1).all() and (table.d >= -1).all()  # control nodes must be leads somewhere
    return table


def _grid_arrays(N):
    """Lead ids and row index of every grid resistor, in file order."""
    import numpy as np
    k = np.arange(N * N, dtype=np.int64)
    r, col = np.divmod(k, N)
    has_h, has_v = col + 1 < N, r + 1 < N
    # per node: [rh?, rv?] in that order -> position of each resistor in the file
    slot = np.zeros((N * N, 2), dtype=bool)
    slot[:, 0], slot[:, 1] = has_h, has_v
    pos = np.cumsum(slot.ravel()).reshape(N * N, 2) - 1
    nres = grid_resistor_count(N)
    a = np.empty(nres, dtype=np.int64)
    b = np.empty(nres, dtype=np.int64)
    a[pos[has_h, 0]], b[pos[has_h, 0]] = k[has_h], k[has_h] + 1
    a[pos[has_v, 1]], b[pos[has_v, 1]] = k[has_v], k[has_v] + N
    return a, b, pos


def grid_table(N, values=None):
    """Component table of grid(N) (== lower(Netlist.from_rows(grid_rows(N))))."""
    import numpy as np
    a, b, _ = _grid_arrays(N)
    nres = len(a)
    last = N * N - 1
    types = np.zeros(nres + 1, dtype=np.uint8)
    types[-1] = 1  # a1, A
    vals = np.ones(nres + 1)
    if values is not None:
        vals[:nres] = np.asarray(values, dtype=np.float64)
    a = np.append(a, 0)  # a1: 1 A from node "1" (id 0) to "g" (id last)
    b = np.append(b, last)
    none = np.full(nres + 1, -1, dtype=np.int64)
    return _table(types, vals, a, b, none, none, none, last)


def cfg5_table(N, seed=5):
    """Component table of cfg5 (== lower(Netlist.from_rows(cfg5_rows(N, seed))))."""
    import numpy as np
    ga, gb, pos = _grid_arrays(N)
    nres = len(ga)
    last = N * N - 1
    rng = random.Random(seed)
    S0, V0 = N * N, 2 * N * N  # ids of the extra nodes s{k}, v{k}
    t, v, a, b, cc, dd, drv = [], [], [], [], [], [], []

    def row(tt, vv, aa, bb, c_=-1, d_=-1, dr=-1):
        t.append(tt); v.append(vv); a.append(aa); b.append(bb)
        cc.append(c_); dd.append(d_); drv.append(dr)

    for k in range(N * N):
        u = rng.random()
        w = rng.random()
        if k == last or k % N == N - 1:
            continue
        if u < 0.01:
            row(2, -5 + 10 * w, S0 + k, last)
            row(0, 1.0, S0 + k, k)
        elif u < 0.015:
            row(5, 0.1 + 0.4 * w, k, last, k, k + 1, int(pos[k, 0]))
        elif u < 0.02:
            row(3, 0.1 + 0.4 * w, V0 + k, last, k, k + 1)
            row(0, 1.0, V0 + k, k)
    row(1, 1.0, 0, last)  # a1
    cat = lambda head, tail, dt: np.concatenate([head, np.asarray(tail, dtype=dt)])  # noqa: E731
    types = cat(np.zeros(nres, dtype=np.uint8), t, np.uint8)
    vals = cat(np.ones(nres), v, np.float64)
    none = np.full(nres, -1, dtype=np.int64)
    return _table(types, vals, cat(ga, a, np.int64), cat(gb, b, np.int64),
                  cat(none, cc, np.int64), cat(none, dd, np.int64), cat(none, drv, np.int64),
                  last)


# ---------------------------------------------------------------------------
# Chain-like passive networks (tests and tools/topologies.py): the cases the sparse path
# solves by exact elimination of low-degree nodes (csrc/lowdeg.hip) rather than by multigrid.
# ---------------------------------------------------------------------------

def passive_table(a, b, values, source_node, ground):
    """Resistors a[i]-b[i] plus a 1 A source from `source_node` to `ground` (largest id)."""
    import numpy as np
    n = len(a)
    types = np.zeros(n + 1, dtype=np.uint8)
    types[-1] = 1
    vals = np.append(np.asarray(values, dtype=np.float64), 1.0)
    a = np.append(np.asarray(a, dtype=np.int64), source_node)
    b = np.append(np.asarray(b, dtype=np.int64), ground)
    none = np.full(n + 1, -1, dtype=np.int64)
    return _table(types, vals, a, b, none, none, none, ground)


def ladder_table(n, seed=1, island=0):
    """n resistors in series, a shunt to ground at every tenth node; `island` > 0 adds a chain
    of that many nodes that touches nothing else (a floating sub-network: singular)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ground = n + 1 + island
    k = np.arange(n, dtype=np.int64)
    taps = np.arange(0, n + 1, 10, dtype=np.int64)
    a = [k, taps]
    b = [k + 1, np.full(len(taps), ground, dtype=np.int64)]
    vals = [rng.uniform(0.5, 2.0, n), rng.uniform(0.5e4, 2e4, len(taps))]
    if island > 1:
        isl = np.arange(n + 1, n + island, dtype=np.int64)
        a.append(isl)
        b.append(isl + 1)
        vals.append(rng.uniform(0.5, 2.0, len(isl)))
    return passive_table(np.concatenate(a), np.concatenate(b), np.concatenate(vals), n, ground)


def chain_table(n, seed=1):
    """n resistors in series, grounded at one end only (condition number ~ n^2)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ground = n + 1
    k = np.arange(n, dtype=np.int64)
    return passive_table(np.append(k, 0), np.append(k + 1, ground), rng.uniform(0.5, 2.0, n + 1), n, ground)


def binary_tree_table(n, seed=1):
    """Complete binary tree of n nodes, the root tied to ground, the source at the last leaf."""
    import numpy as np
    rng = np.random.default_rng(seed)
    child = np.arange(1, n, dtype=np.int64)
    return passive_table(np.append((child - 1) // 2, 0), np.append(child, n), rng.uniform(0.5, 2.0, n),
                         n - 1, n)


def grid_with_wires_table(side, wire, seed=1):
    """side x side grid; a wire of `wire` resistors hangs off every node of the top row and
    ends in a resistor to ground."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ga, gb, _ = _grid_arrays(side)
    nn = side * side
    a, b = [ga], [gb]
    nxt = nn
    for c in range(side):
        ids = np.arange(nxt, nxt + wire, dtype=np.int64)
        a.append(np.append(c, ids[:-1]))
        b.append(ids)
        nxt += wire
    ground = nxt
    ends = np.arange(nn + wire - 1, nxt, wire, dtype=np.int64)
    a.append(ends)
    b.append(np.full(len(ends), ground, dtype=np.int64))
    a, b = np.concatenate(a), np.concatenate(b)
    return passive_table(a, b, rng.uniform(0.5, 2.0, len(a)), nn - 1, ground)
