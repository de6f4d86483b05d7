"""Host-side analysis of the sparse direct route (csrc/slu_analyse.h: row matching, nested dissection, symbolic
factorisation over supernodes) without a GPU: tools/slu_host_check.cpp runs the numeric phase on the host with
the very structures the HIP kernels use and must reproduce SuperLU's answer (reference nodal/nodal.py:325)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from nodal_amd import generators as gen
from nodal_amd.lowering import lower
import nodal_amd as n
from oracle import nodal_oracle as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path_factory.mktemp("slu") / "slu_host_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", exe, os.path.join(ROOT, "tools", "slu_host_check.cpp")])
    return exe


def _run(exe, tmp_path, table):
    G, A = oracle.assemble_fast(table)
    G = G.tocsr()
    G.sort_indices()
    m, xo = str(tmp_path / "m.bin"), str(tmp_path / "x.bin")
    with open(m, "wb") as f:
        np.array([G.shape[0], G.nnz], dtype=np.int64).tofile(f)
        G.indptr.astype(np.int32).tofile(f)
        G.indices.astype(np.int32).tofile(f)
        G.data.astype(np.float64).tofile(f)
        A.astype(np.float64).tofile(f)
    r = subprocess.run([exe, m, xo], capture_output=True, text=True)
    return r, G, A, xo


@pytest.mark.parametrize("name", ["grid(3)", "grid(40)", "cfg5(40)", "cfg5(70)", "ladder", "tree", "every type"])
def test_host_emulation_of_the_direct_route_matches_superlu(checker, tmp_path, name):
    if name.startswith("grid"):
        table = gen.grid_table(int(name[5:-1]))
    elif name.startswith("cfg5"):
        table = gen.cfg5_table(int(name[5:-1]))
    elif name == "ladder":
        table = gen.ladder_table(2000)
    elif name == "tree":
        table = gen.binary_tree_table(1500)
    else:  # doc/test_1's kind: every component type, zero diagonals in the branch rows
        rows = list(gen.grid_rows(12))[:-1]
        rows += [["e1", "E", "5", "1", "g"], ["rx", "R", "2", "x1", "7"], ["v1", "VCVS", "0.5", "x1", "g", "20", "21"],
                 ["ry", "R", "3", "y1", "9"], ["h1", "CCVS", "0.4", "y1", "g", "1", "2", "rh0_0"],
                 ["f1", "CCCS", "0.3", "30", "g", "2", "1", "rh0_0"], ["a1", "A", "1", "40", "g"]]
        table = lower(n.Netlist.from_rows(rows))
    r, G, A, xo = _run(checker, tmp_path, table)
    assert r.returncode == 0, r.stderr[-1500:]  # (also: nothing for the sanitizers to report)
    assert "perturbed" in r.stderr and " 0 perturbed pivots" in r.stderr, r.stderr[-400:]
    x = np.fromfile(xo, dtype=np.float64)
    ref, _ = oracle.solve(G, A, True)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()


def test_structurally_singular_matrix_is_reported(checker, tmp_path):
    """A branch row without any entry (an E source between the ground node and itself) has no column to be
    matched to: the analysis says so (exit code 3) instead of producing an ordering."""
    rows = list(gen.grid_rows(8)) + [["e0", "E", "1", "g", "g"]]
    r, *_ = _run(checker, tmp_path, lower(n.Netlist.from_rows(rows)))
    assert r.returncode == 3 and "structurally singular" in r.stderr


def test_threaded_in_place_dissection_equals_the_sequential_form(tmp_path):
    """csrc/slu_analyse.h dissects in place and hands independent halves to other threads; the elimination order
    and the supernode boundaries must be those of the sequential, stack-driven form it replaced
    (tools/nd_reference.h), whatever the schedule: random graphs with several components, chords and hubs, a
    grid and config 5's topology, with the threading threshold lowered so that small pieces take threads too."""
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "nd_compare")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tools", "nd_compare.cpp")])
    files = []
    for name, table in (("grid", gen.grid_table(90)), ("cfg5", gen.cfg5_table(60))):
        G, _ = oracle.assemble_fast(table)
        G = G.tocsr()
        G.sort_indices()
        path = str(tmp_path / (name + ".bin"))
        with open(path, "wb") as f:
            np.array([G.shape[0], G.nnz], dtype=np.int64).tofile(f)
            G.indptr.astype(np.int32).tofile(f)
            G.indices.astype(np.int32).tofile(f)
            G.data.astype(np.float64).tofile(f)
        files.append(path)
    for env in ({}, {"NODAL_ND_PAR": "40", "NODAL_ND_DEPTH": "6"}, {"NODAL_ND_DEPTH": "0"}):
        r = subprocess.run([exe, "random"] + files, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "DIFFERENT" not in r.stdout and r.stdout.count("identical") == 14
