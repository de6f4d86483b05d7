"""Kernel-level checks on the MI355X (through the C ABI testing hooks)."""
import numpy as np
import pytest

from nodal_amd import _ffi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (128, 128, 16), (130, 70, 32), (257, 300, 128),
                                   (1000, 513, 256), (64, 1, 20), (3, 5, 7),
                                   # the 128 x 128 tile kernel (M N > 512 K) with few K chunks: the trailing updates of
                                   # the sparse direct route's wide fronts (sparse_direct.hip: K = panel width)
                                   (1000, 600, 16), (1000, 600, 32), (1100, 900, 48), (1500, 1400, 64), (900, 800, 20)])
def test_mfma_f64_gemm(M, N, K):
    """fp64 MFMA fragment layout and edge handling, against numpy in fp64.
    Asymmetric integer-valued data makes any row/col or k-order mix-up exact."""
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.randint(-8, 9, size=(M, K)).astype(float)
    B = rng.randint(-8, 9, size=(K, N)).astype(float)
    C = rng.randint(-8, 9, size=(M, N)).astype(float)
    h = _ffi.Handle(0)
    out = h.debug_gemm(A, B, C)
    assert np.array_equal(out, C - A @ B)  # integers: exact
    A, B, C = rng.randn(M, K), rng.randn(K, N), rng.randn(M, N)
    out = h.debug_gemm(A, B, C)
    ref = C - A @ B
    assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) * K
    h.close()


def test_dense_lu_paths_agree_on_passive_network():
    """On a resistor-only network the dense LU skips the pivot search (partial
    pivoting never interchanges on a column diagonally dominant matrix).  Forcing the
    tournament-pivoted path must give the same solution and choose identity pivots."""
    from nodal_amd import generators as gen
    table = gen.grid_table(50)  # n = 2499 > GEPP_MAX: exercises the blocked paths
    out = []
    for force in (0, 1):
        h = _ffi.Handle(0)
        h.set_option(_ffi.OPT_FORCE_PIVOTING, force)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info = h.solve_dense()
        assert info == 0 and h.residual() <= 1e-14
        out.append(x)
        h.close()
    assert np.abs(out[0] - out[1]).max() <= 1e-12 * np.abs(out[0]).max()


def test_dense_lu_tournament_on_general_matrix():
    """cfg5 has zero diagonals and non-symmetric rows: the tournament path must pivot."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    table = gen.cfg5_table(56)  # n ~ 3200
    h = _ffi.Handle(0)
    h.set_option(_ffi.OPT_FORCE_PIVOTING, 1)  # (the default route presolves: next test)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info = h.solve_dense()
    assert info == 0 and h.residual() <= 1e-14
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("side", [3, 8, 16, 22, 31, 44])  # n = 10 .. 1990: one row per thread up to 1024 rows
def test_dense_partial_pivoting_panel_kernel_matches_the_per_column_kernels(side):
    """Partial pivoting with the 32-column panel in registers (one launch per panel, the default) against
    the two-launches-per-column form: same pivots (idamax ties included: the +-1 incidence entries of the
    voltage sources tie all the time), same arithmetic, hence the same bits; and the oracle's dgesv."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    table = gen.cfg5_table(side)
    out = []
    for panel in (1, 0):
        h = _ffi.Handle(0)
        h.set_option(_ffi.OPT_FORCE_PIVOTING, 1)  # (above 512 unknowns the default route presolves)
        h.set_option(_ffi.OPT_GEPP_PANEL, panel)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info = h.solve_dense()
        assert info == 0 and h.residual() <= 1e-14
        out.append(x)
        h.close()
    assert np.array_equal(out[0], out[1])
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.toarray(), A, False)
    assert np.abs(out[0] - xo).max() <= 1e-10 * np.abs(xo).max()


def test_dense_partial_pivoting_panel_kernel_reports_the_zero_pivot_column():
    """Two voltage sources across the same node pair: dgesv stops at an exact zero pivot and both forms
    of the panel factorisation name the same column."""
    from nodal_amd.lowering import lower
    from nodal_amd.netlist import Netlist
    rows = [["r%d" % i, "R", "1", str(i), str(i + 1)] for i in range(1, 40)]
    rows += [["rg", "R", "1", "40", "g"], ["e1", "E", "1", "1", "g"], ["e2", "E", "2", "1", "g"]]
    table = lower(Netlist.from_rows(rows))
    infos = []
    for panel in (1, 0):
        h = _ffi.Handle(0)
        h.set_option(_ffi.OPT_GEPP_PANEL, panel)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        _, info = h.solve_dense()
        infos.append(info)
        h.close()
    assert infos[0] == infos[1] and infos[0] > 0


@pytest.mark.parametrize("side", [17, 23, 32, 46, 47, 50])  # n = 288 .. 2499; last blocks of 32 .. 255 columns
def test_dense_block_inverse_elimination_on_passive_network(side, monkeypatch):
    """Passive networks above GEPP_MAX: block elimination with inverted diagonal blocks
    (default) against the plain no-pivot LU (NODAL_DENSE_BLOCKINV=0) and the oracle."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    table = gen.grid_table(side)
    out = []
    # rank-4 MFMA inverse, LU, scalar inverse, 512-wide blocks (two levels of the Schur recursion)
    for flag, scalar, width in (("1", "0", "256"), ("0", "0", "256"), ("1", "1", "256"), ("1", "0", "512")):
        monkeypatch.setenv("NODAL_DENSE_BLOCKINV", flag)
        monkeypatch.setenv("NODAL_GJ_SCALAR", scalar)
        monkeypatch.setenv("NODAL_BI_WIDTH", width)
        h = _ffi.Handle(0)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info = h.solve_dense()
        assert info == 0 and h.residual() <= 1e-14
        out.append(x)
        h.close()
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    scale = np.abs(xo).max()
    assert np.abs(out[0] - xo).max() <= 1e-10 * scale
    assert np.abs(out[0] - out[1]).max() <= 1e-11 * scale
    assert np.abs(out[0] - out[2]).max() <= 1e-11 * scale
    assert np.abs(out[0] - out[3]).max() <= 1e-11 * scale


def test_dense_block_inverse_multiple_right_hand_sides():
    """nodal_solve_pairs on the dense passive path: the extra right-hand sides ride
    through the block elimination as extra columns."""
    from nodal_amd import generators as gen
    table = gen.grid_table(47)
    rng = np.random.RandomState(5)
    ia = rng.randint(0, table.K, size=7).astype(np.int32)
    ib = rng.randint(-1, table.K, size=7).astype(np.int32)
    ib[ib == ia] = -1
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    dense, info = h.solve_pairs(ia, ib, True)
    assert info == 0
    sparse, info = h.solve_pairs(ia, ib, False)
    assert info == 0
    assert np.abs(dense - sparse).max() <= 1e-9 * np.abs(sparse).max()
    h.close()


@pytest.mark.parametrize("side", [20, 50])  # n = 402 (one block row of 146) and 2502
def test_dense_passive_floating_island_is_reported_singular(side):
    """A resistor island without a path to ground makes G exactly singular.  The passive
    dense path eliminates without pivoting, where rounding can hide the zero pivot: the
    structural test must still report the system singular (reference: dgesv info > 0 ->
    LinAlgError -> UnconnectedCircuitError)."""
    import nodal_amd as n
    from nodal_amd import generators as gen
    rows = [list(r) for r in gen.grid_rows(side)]
    rows += [["Rx", "R", "5", "isl1", "isl2"], ["Ry", "R", "7", "isl2", "isl3"],
             ["Rz", "R", "3", "isl3", "isl1"], ["Ax", "A", "1", "isl1", "isl3"]]
    netlist = n.Netlist.from_rows(rows)
    with pytest.raises(n.UnconnectedCircuitError):
        n.Circuit(netlist, sparse=False).solve()


def test_dense_voltage_sources_take_presolve_and_block_elimination():
    """A large resistor network driven by voltage sources is not passive (branch rows with
    zero diagonals), but eliminating the voltage-defined branches leaves a passive network:
    the dense path then solves that one by block elimination and recovers node potentials and
    branch currents.  Same answer as the tournament-pivoted LU of the original matrix and as
    the oracle."""
    import random
    import nodal_amd as n
    from nodal_amd import generators as gen, lowering
    from oracle import nodal_oracle as oracle
    rng = random.Random(7)
    rows = [list(r) for r in gen.grid_rows(50)]
    nodes = [str(k) for k in rng.sample(range(2, 2400), 40)]
    for i, node in enumerate(nodes[:30]):
        rows.append([f"e{i}", "E", repr(1.0 + 0.1 * i), node, "g"])
    for i in range(5):  # floating sources between two grid nodes
        rows.append([f"ef{i}", "E", repr(0.5 + i), nodes[30 + 2 * i], nodes[31 + 2 * i]])
    table = lowering.lower(n.Netlist.from_rows(rows))
    assert table.B == 35 and table.K + table.B > 2048
    out = []
    for force in (0, 1):
        h = _ffi.Handle(0)
        h.set_option(_ffi.OPT_FORCE_PIVOTING, force)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info = h.solve_dense()
        assert info == 0 and h.residual() <= 1e-13
        out.append(x)
        h.close()
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    scale = np.abs(xo).max()
    assert np.abs(out[0] - xo).max() <= 1e-9 * scale
    assert np.abs(out[0] - out[1]).max() <= 1e-10 * scale


def _stacked_source_rows(side, dependent):
    """grid(side) plus chains of sources: a stack of three E sources on ground, an E between two
    of the stacked nodes' neighbours, a VCVS controlled by a stacked (eliminated) node and a
    CCVS -- every voltage-defined branch hangs in one of two source trees."""
    from nodal_amd import generators as gen
    rows = [list(r) for r in gen.grid_rows(side)]
    rows += [
        ["e1", "E", "2.0", "5", "g"],        # e(5) = 2
        ["e2", "E", "1.5", "40", "5"],       # stacked on a pivot: e(40) = e(5) + 1.5
        ["e3", "E", "-0.5", "41", "40"],     # second storey
        ["e4", "E", "0.75", "300", "301"],   # floating pair, its own tree
        ["e5", "E", "0.25", "302", "301"],   # shares the base node of e4
    ]
    if dependent:  # (the dense route needs a passive reduced network: independent sources only)
        rows += [
            ["v1", "VCVS", "3.0", "700", "g", "40", "900"],   # controlled by an eliminated node
            ["h1", "CCVS", "2.0", "800", "801", str(side + 2), str(side + 3), "rh1_1"],  # its driver's leads
            # CCCS sensing the current of a resistor that hangs on a source node (control = pivot "5")
            ["f1", "CCCS", "0.5", "1000", "g", "5", "6", "rh0_4"],
        ]
    return rows


@pytest.mark.parametrize("side,dense", [(50, True), (70, False)])
def test_presolve_resolves_chains_of_sources(side, dense, monkeypatch, capfd):
    """Stacked voltage sources and a dependent source controlled by an eliminated node: the
    presolve substitutes along the source trees, the reduced network is solved (dense block
    elimination / multigrid) and potentials and branch currents are recovered level by
    level.  Checked against the oracle and, through the trace, that the route was taken."""
    import nodal_amd as n
    from nodal_amd import lowering
    from oracle import nodal_oracle as oracle
    monkeypatch.setenv("NODAL_TRACE", "1")
    table = lowering.lower(n.Netlist.from_rows(_stacked_source_rows(side, not dense)))
    assert table.B == (5 if dense else 8)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    if dense:
        x, info = h.solve_dense()
    else:
        x, info = h.solve_sparse()[:2]
    assert info == 0 and h.residual() <= 1e-12
    err = capfd.readouterr().err
    assert "[presolve] accepted" in err, err[-400:]
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_dense_dependent_sources_take_presolve_and_optimistic_elimination(monkeypatch, capfd):
    """cfg5 (voltage sources + CCCS / VCVS) on the dense path: after the presolve the reduced
    matrix is a conductance matrix plus transconductance terms; the pivot-free block elimination
    is tried and accepted on the ORIGINAL system's residual."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    monkeypatch.setenv("NODAL_TRACE", "1")
    table = gen.cfg5_table(56)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info = h.solve_dense()
    assert info == 0 and h.residual() <= 1e-13
    assert "[presolve] accepted" in capfd.readouterr().err
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("side", [9, 12, 25, 40, 64])
def test_multigrid_path_on_small_networks(side):
    """Small passive networks take the direct solve by default; the multigrid-preconditioned
    CG is still exercised on them explicitly (tiny hierarchies, levels that all fit the LDS
    tail, the Jacobi-CG fallback below the multigrid's minimum size)."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    import random
    rng = random.Random(side)
    vals = [10.0 ** rng.uniform(-1, 1) for _ in range(gen.grid_resistor_count(side))]
    table = gen.grid_table(side, vals)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, relres = h.solve_sparse(method=_ffi.SPARSE_PCG)
    assert info == 0 and iters > 0 and h.residual() <= 1e-12
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("nu", ["112", "212", "312", "323", "133"])
def test_multigrid_sweep_counts_change_the_iteration_count_not_the_solution(nu, monkeypatch):
    """NODAL_SA_NU: Jacobi sweeps per side at level 0 / level 1 / deeper levels (1-3 each; the default is 212).
    Every choice is a symmetric preconditioner of the same flexible CG: the same solution as SuperLU, and more
    sweeps at level 0 never cost iterations."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    table = gen.grid_table(150)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)

    def solve(env):
        if env is None:
            monkeypatch.delenv("NODAL_SA_NU", raising=False)
        else:
            monkeypatch.setenv("NODAL_SA_NU", env)
        h = _ffi.Handle(0)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info, iters, relres = h.solve_sparse(method=_ffi.SPARSE_PCG)
        assert info == 0 and iters > 0 and h.residual() <= 1e-12
        h.close()
        return x, iters

    x, iters = solve(nu)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    x1, iters1 = solve("112")
    if nu[0] > "1" and nu[1:] == "12":
        assert iters < iters1


def _chainlike_table(kind):
    from nodal_amd import generators as gen
    return {"ladder": lambda: gen.ladder_table(30000),
            "chain": lambda: gen.chain_table(6000),
            "tree": lambda: gen.binary_tree_table(50000),
            "wires": lambda: gen.grid_with_wires_table(40, 120)}[kind]()


@pytest.mark.parametrize("kind", ["ladder", "chain", "tree", "wires"])
def test_sparse_low_degree_elimination(kind, monkeypatch, capfd):
    """Ladders, chains, trees and grids with dangling wires: the sparse passive path removes
    the nodes with one or two neighbours exactly, round by round (csrc/lowdeg.hip), and hands
    what is left to the direct solve or the multigrid.  Same answer as SuperLU and as the
    multigrid on the unreduced network."""
    from oracle import nodal_oracle as oracle
    table = _chainlike_table(kind)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    scale = np.abs(xo).max()
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, relres = h.solve_sparse()
    assert "[lowdeg]" in capfd.readouterr().err
    assert info == 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * scale
    monkeypatch.setenv("NODAL_LOWDEG", "0")
    x1, info1, iters1, _ = h.solve_sparse()
    assert "[lowdeg]" not in capfd.readouterr().err
    assert info1 == 0 and iters1 > 0
    assert np.abs(x1 - xo).max() <= 1e-9 * scale
    h.close()


@pytest.mark.parametrize("island", [2, 3, 700])
def test_sparse_low_degree_elimination_reports_floating_chain(island):
    """A chain that touches nothing else collapses to a single node without neighbours and
    without a path to ground: singular, like the structural test of the multigrid path says."""
    from nodal_amd import generators as gen
    table = gen.ladder_table(20000, island=island)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, relres = h.solve_sparse()
    assert info > 0 and np.isnan(x).all()
    h.close()
    # regular without the island
    table = gen.ladder_table(20000)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, relres = h.solve_sparse()
    assert info == 0 and np.isfinite(x).all()
    h.close()


def test_dense_entry_point_eliminates_low_degree_nodes_too(monkeypatch, capfd):
    """`Circuit(netlist)` is dense by default: a passive ladder is still reduced by the exact
    elimination before anything is formed densely, with the same answer as LAPACK on the full
    matrix; a floating chain is reported singular."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    table = gen.ladder_table(3000)
    G, A = oracle.assemble_fast(table)
    xo = np.linalg.solve(G.toarray(), A)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info = h.solve_dense()
    assert "[lowdeg]" in capfd.readouterr().err
    assert info == 0 and np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    monkeypatch.setenv("NODAL_LOWDEG", "0")
    x1, info1 = h.solve_dense()
    assert info1 == 0 and np.abs(x1 - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()
    monkeypatch.delenv("NODAL_LOWDEG")
    h = _ffi.Handle(0)
    h.upload(gen.ladder_table(3000, island=40))
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info = h.solve_dense()
    assert info > 0
    h.close()


@pytest.mark.parametrize("seed", range(6))
def test_sparse_low_degree_elimination_on_random_networks(seed):
    """Random trees with a few extra edges (cycles), subdivided edges (wires), parallel
    resistors and several ties to ground: every mix of one- and two-neighbour nodes, fill that
    lands on existing entries, rounds that stop early.  Against SuperLU."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1500, 12000))
    parent = np.array([rng.integers(max(0, i - 1 - int(rng.integers(0, 50))), i) for i in range(1, n)],
                      dtype=np.int64)
    a = [parent, ]
    b = [np.arange(1, n, dtype=np.int64)]
    extra = int(n * rng.uniform(0.0, 0.3))
    ea, eb = rng.integers(0, n, extra), rng.integers(0, n, extra)
    keep = ea != eb
    a.append(ea[keep]); b.append(eb[keep])
    dup = rng.integers(0, n - 1, n // 20)          # parallel resistors on tree edges
    a.append(parent[dup]); b.append(dup + 1)
    ground = n
    ties = rng.integers(0, n, max(1, n // 200))
    a.append(ties); b.append(np.full(len(ties), ground, dtype=np.int64))
    a, b = np.concatenate(a), np.concatenate(b)
    vals = 10.0 ** rng.uniform(-1.5, 1.5, len(a))
    table = gen.passive_table(a, b, vals, int(rng.integers(0, n)), ground)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, relres = h.solve_sparse()
    assert info == 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("dense", [False, True])
def test_pair_sweep_on_a_ladder_uses_the_elimination(dense, monkeypatch, capfd):
    """Equivalent resistances between nodes of a long ladder (nodal_solve_pairs): every pair
    goes through the exact elimination instead of hundreds of multigrid iterations (sparse)
    or a dense factorisation of the whole ladder (dense); same numbers as SuperLU."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    import scipy.sparse.linalg as spla
    table = gen.ladder_table(20000 if not dense else 6000)
    G, _ = oracle.assemble_fast(table)
    lu = spla.splu(G.tocsc())
    n = G.shape[0]
    ia = np.array([0, 17, n - 1, 5], dtype=np.int32)
    ib = np.array([n - 1, 4000, -1, 6], dtype=np.int32)   # -1: the ground node
    want = []
    for a, b in zip(ia, ib):
        rhs = np.zeros(n)
        rhs[a] += 1.0
        if b >= 0:
            rhs[b] -= 1.0
        x = lu.solve(rhs)
        want.append(x[a] - (x[b] if b >= 0 else 0.0))
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    got, info = h.solve_pairs(ia, ib, dense)
    assert "[lowdeg]" in capfd.readouterr().err
    assert info == 0
    assert np.abs(got - np.array(want)).max() <= 1e-9 * np.abs(want).max()
    h.close()


@pytest.mark.parametrize("side,wire", [(40, 120), (80, 60)])
def test_floating_island_survives_the_elimination_rounds(side, wire):
    """A floating 30 x 30 grid next to a grid with dangling wires: the rounds shorten the wires,
    the island (no low-degree nodes) stays, and whichever solver gets the remainder -- the
    dense one with its LDS connectivity check (<= 4096 unknowns left) or the multigrid with
    the inherited 'touches ground' flags -- must report the network singular."""
    from nodal_amd import generators as gen
    rng = np.random.default_rng(3)
    ga, gb, _ = gen._grid_arrays(side)
    nn = side * side
    a, b = [ga], [gb]
    nxt = nn
    for c in range(side):
        ids = np.arange(nxt, nxt + wire, dtype=np.int64)
        a.append(np.append(c, ids[:-1])); b.append(ids)
        nxt += wire
    ends = np.arange(nn + wire - 1, nxt, wire, dtype=np.int64)
    ia_, ib_, _ = gen._grid_arrays(30)
    island0 = nxt
    ground = island0 + 900
    for tie in (False, True):
        aa = a + [ends, ia_ + island0] + ([np.array([island0 + 17])] if tie else [])
        bb = b + [np.full(len(ends), ground, dtype=np.int64), ib_ + island0] + ([np.array([ground])] if tie else [])
        aa, bb = np.concatenate(aa), np.concatenate(bb)
        table = gen.passive_table(aa, bb, rng.uniform(0.5, 2.0, len(aa)), nn - 1, ground)
        h = _ffi.Handle(0)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info, iters, relres = h.solve_sparse()
        if tie:   # island tied to ground: regular
            assert info == 0 and np.isfinite(x).all() and h.residual() <= 1e-12
        else:
            assert info > 0 and np.isnan(x).all()
        h.close()


def test_low_degree_elimination_structure_is_reused_across_a_value_sweep(monkeypatch, capfd):
    """One ladder topology, several sets of resistances on one handle: the eliminated sets and
    the patterns of the reduced matrices are kept from the first member on (no '[lowdeg]'
    build lines afterwards), the values are not -- every member against SuperLU.  A different
    circuit uploaded to the same handle rebuilds everything."""
    from nodal_amd import generators as gen
    from nodal_amd.lowering import ComponentTable  # noqa: F401  (table type of the generators)
    from oracle import nodal_oracle as oracle
    import copy
    rng = np.random.default_rng(9)
    table = gen.ladder_table(12000)
    members = 4
    vals = np.tile(table.value, (members, 1))
    for m in range(1, members):
        vals[m, :-1] = table.value[:-1] * 10.0 ** rng.uniform(-1.0, 1.0, table.ncomp - 1)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    h.upload_values(vals)
    for m in range(members):
        assert h.assemble_numeric(m)[0] == _ffi.OK
        capfd.readouterr()
        x, info, iters, _ = h.solve_sparse()
        err = capfd.readouterr().err
        assert ("[lowdeg]" in err) == (m == 0)
        tm = copy.copy(table)
        tm.value = vals[m]
        G, A = oracle.assemble_fast(tm)
        xo, _ = oracle.solve(G.tocsr(), A, True)
        assert info == 0 and np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    # another circuit of the same size class on the same handle
    other = gen.binary_tree_table(12002)
    h.upload(other)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    assert "[lowdeg]" in capfd.readouterr().err
    G, A = oracle.assemble_fast(other)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    assert info == 0 and np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("shape", ["star", "double_star", "caterpillar", "ring"])
def test_sparse_low_degree_elimination_extreme_shapes(shape):
    """Networks that collapse to one or two unknowns in a single round (a hub with thousands of
    leaves), that alternate leaves and a spine, or that are one big cycle."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    rng = np.random.default_rng(5)
    m = 5000
    if shape == "star":            # hub 0, leaves 1..m, hub tied to ground
        a = np.append(np.zeros(m, dtype=np.int64), 0)
        b = np.append(np.arange(1, m + 1, dtype=np.int64), m + 1)
        src, ground = m, m + 1
    elif shape == "double_star":   # two hubs joined by a resistor, each with m/2 leaves
        half = m // 2
        a = np.concatenate([np.zeros(half, dtype=np.int64), np.ones(half, dtype=np.int64), [0], [1]])
        b = np.concatenate([np.arange(2, 2 + half), np.arange(2 + half, 2 + 2 * half), [1], [2 + 2 * half]])
        src, ground = 2, 2 + 2 * half
    elif shape == "caterpillar":   # spine 0..m-1, one leaf per spine node, both ends grounded
        spine = np.arange(m - 1, dtype=np.int64)
        a = np.concatenate([spine, np.arange(m, dtype=np.int64), [0], [m - 1]])
        b = np.concatenate([spine + 1, np.arange(m, 2 * m, dtype=np.int64), [2 * m], [2 * m]])
        src, ground = 2 * m - 1, 2 * m
    else:                          # ring of m nodes, one tie to ground
        ring = np.arange(m, dtype=np.int64)
        a = np.append(ring, 0)
        b = np.append((ring + 1) % m, m)
        src, ground = m // 2, m
    table = gen.passive_table(a, b, rng.uniform(0.5, 2.0, len(a)), src, ground)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    for _ in range(2):  # second solve: cached elimination structure
        x, info, iters, relres = h.solve_sparse()
        assert info == 0 and h.residual() <= 1e-12
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


@pytest.mark.parametrize("decades", [4, 6])
def test_multigrid_block_smoother_on_high_contrast_grid(decades, monkeypatch, capfd):
    """Resistances spread log-uniformly over several decades: many nodes hang on one dominant
    link, point Jacobi cannot damp errors that are constant on such strongly coupled clusters
    inside an aggregate, and the setup switches to Jacobi over the aggregates' diagonal blocks.
    Same answer as SuperLU, and far fewer iterations than with point Jacobi."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    side = 90
    rng = np.random.default_rng(decades)
    vals = 10.0 ** rng.uniform(-decades / 2, decades / 2, gen.grid_resistor_count(side))
    table = gen.grid_table(side, vals)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    assert "[amg] blocks:" in capfd.readouterr().err  # contrast mode (chosen by the setup, or forced)
    assert info == 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    monkeypatch.setenv("NODAL_AMG_BLOCK", "0")
    x0, info0, iters0, _ = h.solve_sparse()
    assert info0 == 0 and iters0 > 2 * iters
    assert np.abs(x0 - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_multigrid_block_smoother_forced_on_uniform_and_general_networks(monkeypatch):
    """NODAL_AMG_BLOCK=1: the block smoother on networks that would not select it -- a uniform
    grid (multigrid CG) and config 5 in miniature (presolve + FGMRES preconditioned by the
    multigrid on the node block)."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    monkeypatch.setenv("NODAL_AMG_BLOCK", "1")
    for table in (gen.grid_table(80), gen.cfg5_table(90)):
        G, A = oracle.assemble_fast(table)
        xo, _ = oracle.solve(G.tocsr(), A, True)
        h = _ffi.Handle(0)
        h.upload(table)
        h.assemble_symbolic()
        assert h.assemble_numeric()[0] == _ffi.OK
        x, info, iters, _ = h.solve_sparse()
        assert info == 0 and iters > 0 and h.residual() <= 1e-12
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
        h.close()


def test_multigrid_contrast_mode_with_a_hub(monkeypatch, capfd):
    """A hub tied by strong links to 300 nodes of a grid: those nodes refuse weak matches and
    join the hub's pair, whose aggregate then exceeds the 32 nodes a dense smoother block may
    have -- it keeps point Jacobi (diagonal fallback) while the rest of the network gets
    blocks."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    side = 80
    rng = np.random.default_rng(21)
    ga, gb, _ = gen._grid_arrays(side)
    nn = side * side
    hub = nn
    spokes = rng.choice(nn, 300, replace=False)
    a = np.concatenate([ga, np.full(300, hub, dtype=np.int64), [0]])
    b = np.concatenate([gb, spokes.astype(np.int64), [nn + 1]])
    vals = np.concatenate([rng.uniform(0.5, 2.0, len(ga)), rng.uniform(0.005, 0.02, 300), [1.0]])  # ohms
    table = gen.passive_table(a, b, vals, nn - 1, nn + 1)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    err = capfd.readouterr().err
    import re
    m = re.search(r"\[amg\] blocks: \d+ aggregates, \d+ larger than 16, (\d+) larger than 32, largest (\d+)", err)
    assert m and int(m.group(1)) >= 1 and int(m.group(2)) > 32  # level 0: the hub's aggregate
    assert info == 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_expander_like_network_keeps_the_jacobi_preconditioner(monkeypatch, capfd):
    """Random long-range connections: the first Galerkin product of the multigrid setup fills in
    (9 -> ~60 entries per row) and every level below would be nearly dense.  The setup notices,
    keeps a one-level 'hierarchy' and the (well-conditioned) network is solved by Jacobi-
    preconditioned CG in a few dozen iterations."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    rng = np.random.default_rng(12)
    n = 6000  # (SuperLU fills in catastrophically on such graphs: dense LAPACK is the oracle here)
    a = rng.integers(0, n, 3 * n)
    b = rng.integers(0, n, 3 * n)
    keep = a != b
    ring = np.arange(n)
    a = np.concatenate([a[keep], ring, [0]])
    b = np.concatenate([b[keep], (ring + 1) % n, [n]])
    table = gen.passive_table(a, b, rng.uniform(0.5, 2.0, len(a)), n // 2, n)
    G, A = oracle.assemble_fast(table)
    xo = np.linalg.solve(G.toarray(), A)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    err = capfd.readouterr().err
    import os
    if os.environ.get("NODAL_AMG_BLOCK") is None:  # (a forced smoother mode coarsens differently)
        assert "expander-like" in err
    assert info == 0 and 0 < iters < 200 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_hub_node_with_thousands_of_neighbours():
    """A node tied to 4000 nodes of a grid: its row block exceeds what the CSR-stream kernels
    stage at once (chunked walk, the hub's row summed by the whole workgroup), its aggregate
    has thousands of members (restriction through the same pattern), and its coarse images
    keep their levels out of the LDS tail."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    side = 120
    rng = np.random.default_rng(33)
    ga, gb, _ = gen._grid_arrays(side)
    nn = side * side
    spokes = rng.choice(nn, 4000, replace=False).astype(np.int64)
    a = np.concatenate([ga, np.full(4000, nn, dtype=np.int64), [0]])
    b = np.concatenate([gb, spokes, [nn + 1]])
    table = gen.passive_table(a, b, rng.uniform(0.5, 2.0, len(a)), nn - 1, nn + 1)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info, iters, _ = h.solve_sparse()
    assert info == 0 and iters > 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_multigrid_contrast_mode_on_anisotropic_grid(monkeypatch, capfd):
    """Vertical resistors 1000 x the horizontal ones: every node has two strong links and two
    weak ones (none carries 0.9 of the diagonal, but the links are graded 1000 : 1), the setup
    must pick the contrast mode, which needs a fraction of the iterations of point Jacobi."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    side = 90
    ga, gb, _ = gen._grid_arrays(side)
    vals = np.where(gb == ga + 1, 1.0, 1000.0)
    table = gen.grid_table(side, vals)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    assert "[amg] blocks:" in capfd.readouterr().err
    assert info == 0 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    monkeypatch.setenv("NODAL_AMG_BLOCK", "0")
    x0, info0, iters0, _ = h.solve_sparse()
    assert info0 == 0 and iters0 > 2 * iters
    h.close()


def test_a_few_near_shorts_select_the_contrast_mode(monkeypatch, capfd):
    """A uniform grid in which 60 resistors are replaced by 0.1 milli-ohm wires: a fraction of a
    percent of the nodes, but each pair joined by a short is a near-null mode point Jacobi cannot
    damp -- the setup counts them (not their share) and picks the contrast mode."""
    from nodal_amd import generators as gen
    from oracle import nodal_oracle as oracle
    side = 100
    rng = np.random.default_rng(8)
    vals = rng.uniform(0.5, 2.0, gen.grid_resistor_count(side))
    vals[rng.choice(len(vals), 60, replace=False)] = 1e-4  # (micro-ohms would put kappa beyond the 1e-9 bar)
    table = gen.grid_table(side, vals)
    G, A = oracle.assemble_fast(table)
    xo, _ = oracle.solve(G.tocsr(), A, True)
    monkeypatch.setenv("NODAL_TRACE", "1")
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    capfd.readouterr()
    x, info, iters, _ = h.solve_sparse()
    assert "[amg] blocks:" in capfd.readouterr().err
    assert info == 0 and iters <= 80 and h.residual() <= 1e-12
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def _random_graph_table(n, deg, seed):
    """A connected passive network without any band structure: a path through all nodes plus
    n * deg / 2 random edges, resistances in [0.5, 2)."""
    from nodal_amd import generators as gen
    rng = np.random.default_rng(seed)
    a = np.concatenate([np.arange(n - 1), rng.integers(0, n, n * deg // 2)])
    b = np.concatenate([np.arange(1, n), rng.integers(0, n, n * deg // 2)])
    keep = a != b
    a, b = a[keep], b[keep]
    return gen.passive_table(a, b, rng.uniform(0.5, 2.0, a.size), 0, n - 1)


@pytest.mark.parametrize("n", [520, 700, 1898, 3000, 7000])
def test_dense_symmetric_block_elimination_on_unbanded_networks(n):
    """The symmetric block elimination (csrc/block_elim.hip: only the upper block triangle is updated,
    A22 -= V^T W with the transposed-operand GEMM) on matrices WITHOUT a band: on a grid most tiles of
    the trailing matrix stay zero and a tile the update skips by mistake goes unnoticed -- a random
    graph fills every one of them.  n = 700: the last diagonal block spans two 128-tiles; n = 7000:
    512-wide blocks first, 256-wide ones after.  Against numpy.linalg.solve (reference
    nodal/nodal.py:327) up to n = 3000, by the scaled residual beyond."""
    from oracle import nodal_oracle as oracle
    table = _random_graph_table(n, 6, n)
    h = _ffi.Handle(0)
    h.upload(table)
    h.assemble_symbolic()
    assert h.assemble_numeric()[0] == _ffi.OK
    x, info = h.solve_dense()
    assert info == 0 and h.residual() <= 1e-14
    if n <= 3000:
        G, A = oracle.assemble_fast(table)
        xo, _ = oracle.solve(G.toarray(), A, False)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    h.close()


def test_a_host_wait_that_runs_out_says_where_and_what():
    """Round 5: every host wait of the library is bounded (csrc/wait.hip).  With a bound of a few microseconds
    the first wait of a large solve runs out: NODAL_E_HIP, the wait site (file:line) and the last kernel enqueued
    on the stream in nodal_last_error, the handle refuses further calls (tests/wait_child.py)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NODAL_WAIT_TIMEOUT_S="0.000002")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "wait_child.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "wait child ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
