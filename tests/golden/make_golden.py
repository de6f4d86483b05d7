#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (Nodal.py v1.3.0).

Runs only in the build container, where the reference is mounted read-only at
/root/reference; the outputs (JSON fixtures in this directory) are data and
travel to the GPU box, the reference does not.  Usage:

    python tests/golden/make_golden.py            # small + medium cases
    python tests/golden/make_golden.py --large    # adds grid(1000) / cfg5(1000)

Every case stores the netlist rows, the reference's integer outputs (ground,
nodenum, anomnum, component order), its G / A, and its dense and sparse
solutions, or the exception it raised.  The netlist rows of the reference's
doc/*.csv examples are (c) 2018 Enrico Miccoli, MIT licence; they are stored
here as test vectors (comment lines dropped).
"""

import argparse
import csv
import io
import json
import os
import sys
import tempfile
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402

import nodal as ref  # noqa: E402  (the reference)
import nodal.equiv as ref_equiv  # noqa: E402

assert os.path.abspath(ref.__file__).startswith(REF), ref.__file__

from nodal_amd import generators as gen  # noqa: E402

DOC = [
    "1.6.1", "buffer", "netlist", "opmodel_amplifier", "opmodel_voltage_buffer",
    "resistive_1", "resistive_2", "resistive_3", "test_1", "unconnected_0",
    "unconnected_1",
]


def rows_of_doc(name):
    with open(f"{REF}/doc/{name}.csv") as f:
        rows = [r for r in csv.reader(f, skipinitialspace=True)]
    return [r for r in rows if r != [] and not r[0].startswith("#")]


def write_tmp(rows):
    f = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False)
    for r in rows:
        f.write(",".join(r) + "\n")
    f.close()
    return f.name


def exc_info(e):
    return {"type": type(e).__name__, "args": [str(a) for a in e.args]}


def solve_both(netlist, want_matrix):
    out = {}
    for mode in ("dense", "sparse"):
        rec = {}
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                circ = ref.Circuit(netlist, sparse=(mode == "sparse"))
                if mode == "dense":
                    G = np.array(circ.G)
                    A = np.array(circ.A)
                    out["currents"] = list(circ.currents)
                    out["A"] = A.tolist()
                    ii, jj = np.nonzero(G)
                    out["G_coo"] = [ii.tolist(), jj.tolist(), G[ii, jj].tolist()]
                    if want_matrix:
                        out["G_dense"] = G.tolist()
                else:
                    csr = circ.G
                    out["nnz_sparse"] = int(csr.nnz)
                sol = circ.solve()
                rec["x"] = [float(v) for v in sol.result]
                rec["str"] = str(sol)
                rec["warnings"] = sorted({type(x.message).__name__ for x in w})
        except Exception as e:  # noqa: BLE001 - the exception IS the golden value
            rec["error"] = exc_info(e)
        out[mode] = rec
    return out


def netlist_record(nl):
    return {
        "ground": nl.ground,
        "degrees": [[k, v] for k, v in nl.degrees.items()],
        "nodenum": [[k, v] for k, v in nl.nodenum.items()],
        "anomnum": [[k, v] for k, v in nl.anomnum.items()],
        "component_keys": list(nl.component_keys),
        "nums": dict(nl.nums),
        "is_connected": bool(ref.is_connected(nl)),
    }


def run_case(name, rows, want_matrix=True, raw_text=None):
    case = {"name": name, "rows": rows}
    if raw_text is not None:
        case["raw_text"] = raw_text
        f = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False)
        f.write(raw_text)
        f.close()
        path = f.name
    else:
        path = write_tmp(rows)
    try:
        try:
            nl = ref.Netlist(path)
        except Exception as e:  # noqa: BLE001
            case["parse_error"] = exc_info(e)
            return case
        case.update(netlist_record(nl))
        case.update(solve_both(nl, want_matrix))
    finally:
        os.unlink(path)
    return case


def small_cases():
    cases = []
    for name in DOC:
        cases.append(run_case("doc/" + name, rows_of_doc(name)))

    R = lambda n, v, a, b: [n, "R", v, a, b]  # noqa: E731
    extra = {
        "zero_ohm": [R("r1", "1", "1", "g"), R("r2", "0", "1", "2"), ["a1", "A", "1", "2", "g"]],
        "neg_zero_ohm": [R("r1", "-0.0", "1", "g"), ["a1", "A", "1", "1", "g"]],
        "missing_driver": [R("r1", "1", "1", "g"), ["d1", "CCCS", "2", "1", "g", "1", "g", "nope"]],
        "missing_driver_ccvs": [R("r1", "1", "1", "g"), ["d1", "CCVS", "2", "2", "g", "1", "g", "nope"], R("r2", "1", "2", "g")],
        "e_driven_cccs": [R("r1", "1", "1", "g"), ["e1", "E", "1", "1", "g"], ["d1", "CCCS", "2", "2", "g", "1", "g", "e1"], R("r2", "1", "2", "g")],
        "a_driven_ccvs": [R("r1", "1", "1", "g"), ["a1", "A", "1", "1", "g"], ["d1", "CCVS", "2", "2", "g", "1", "g", "a1"], R("r2", "1", "2", "g")],
        "opamp": [R("r1", "1", "1", "g"), ["q1", "OPAMP", "1", "1", "g", "2", "3"]],
        "two_e_same_pair": [R("r1", "1", "1", "g"), ["e1", "E", "1", "1", "g"], ["e2", "E", "2", "1", "g"]],
        "duplicate_r": [R("r1", "1", "1", "g"), R("r1", "4", "1", "2"), R("r2", "1", "2", "g"), ["a1", "A", "1", "1", "g"]],
        "duplicate_e": [R("r1", "1", "1", "g"), ["e1", "E", "1", "1", "g"], ["e1", "E", "2", "1", "g"]],
        "self_loop_r": [R("r1", "1", "1", "g"), R("r2", "3", "1", "1"), ["a1", "A", "1", "1", "g"]],
        "self_loop_r_bits": [R("r1", "0.3", "1", "g"), R("r2", "1e-17", "1", "1"), R("r3", "0.7", "1", "2"), R("r4", "3", "2", "g"), ["a1", "A", "1", "1", "g"]],
        "no_g": [R("r1", "1", "a", "b"), R("r2", "1", "b", "c"), R("r3", "1", "c", "a"), R("r4", "1", "b", "d"), ["a1", "A", "1", "a", "d"]],
        "e_same_node": [R("r1", "1", "1", "g"), ["e1", "E", "1", "1", "1"]],
        "e_both_ground": [R("r1", "1", "1", "g"), ["e1", "E", "1", "g", "g"], ["a1", "A", "1", "1", "g"]],
        "control_node_missing": [R("r1", "1", "1", "g"), ["d1", "VCVS", "2", "2", "g", "zz", "g"], R("r2", "1", "2", "g")],
        "vccs_standalone": [R("r1", "2", "1", "g"), ["a1", "A", "1", "1", "g"], ["d1", "VCCS", "3", "2", "g", "1", "g"], R("r2", "5", "2", "g")],
        "vcvs_ctrl_on_leads": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["a1", "A", "1", "1", "g"], ["d1", "VCVS", "0.25", "1", "2", "1", "2"]],
        "vcvs_ctrl_same": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["a1", "A", "1", "1", "g"], ["d1", "VCVS", "0.3", "2", "g", "1", "1"]],
        "ccvs_ctrl_on_leads": [R("r1", "2", "1", "2"), R("r2", "3", "2", "g"), R("r3", "1", "1", "g"), ["a1", "A", "1", "1", "g"], ["d1", "CCVS", "0.5", "1", "2", "1", "2", "r1"]],
        "ccvs_reversed_ctrl": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["a1", "A", "1", "1", "g"], ["d1", "CCVS", "0.5", "2", "g", "g", "1", "r1"]],
        "ccvs_ctrl_mismatch": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["d1", "CCVS", "0.5", "2", "g", "2", "g", "r1"]],
        "cccs_ctrl_on_anode": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), R("r3", "1", "1", "2"), ["a1", "A", "1", "1", "g"], ["d1", "CCCS", "0.5", "2", "g", "1", "g", "r1"]],
        "cccs_ctrl_same": [R("r1", "2", "1", "1"), R("r2", "3", "1", "g"), ["d1", "CCCS", "0.5", "1", "g", "1", "1", "r1"]],
        "duplicate_ccvs": [R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["a1", "A", "1", "1", "g"], ["d1", "CCVS", "0.5", "2", "g", "1", "g", "r1"], ["d1", "CCVS", "0.7", "2", "g", "1", "g", "r1"]],
        "driver_defined_later": [["d1", "CCCS", "2", "2", "g", "1", "g", "r1"], R("r1", "2", "1", "g"), R("r2", "3", "2", "g"), ["a1", "A", "1", "1", "g"]],
        "current_only_a": [R("r1", "2", "1", "g"), ["a1", "A", "1.5", "g", "1"], ["a2", "A", "0.25", "1", "g"]],
        "floating_island": [R("r1", "1", "1", "g"), R("r2", "1", "2", "3"), ["a1", "A", "1", "1", "g"]],
        "opmodel_short_row": [["q1", "OPMODEL", "1", "2", "g", "3"]],
        "short_row": [["aaaaa"]],
        "unknown_type": [["v1", "VoltageSource", "5", "1", "2"]],
        "bad_float": [["r1", "R", "one_ohm", "1", "2"]],
        "too_many": [["r1", "R", "5", "1", "2", "3"]],
        "empty_netlist": [],
        "trailing_blank_label": [R("r1", "1", "1 ", "g"), R("r2", "1", "1", "g"), ["a1", "A", "1", "1", "g"]],
    }
    for name, rows in extra.items():
        cases.append(run_case("edge/" + name, rows))

    raw = {
        "comments_and_blanks": "# header\n\nr1, R, 1, 1, 2\n# mid\nr2, R, 1, 2, g\n\na1, A, 1, 1, g\n",
        "whitespace_only_line": "r1,R,1,1,g\n   \na1,A,1,1,g\n",
        "empty_first_field": ",R,1,1,g\n",
        "quoted_fields": 'r1,R,1,"1","g"\n"a1",A,2,1,g\n',
    }
    for name, text in raw.items():
        cases.append(run_case("raw/" + name, None, raw_text=text))
    return cases


def equiv_cases():
    out = []
    for name in ("resistive_1", "resistive_2", "resistive_3"):
        nl = ref.Netlist(f"{REF}/doc/{name}.csv")
        rec = {"name": "doc/" + name, "rows": rows_of_doc(name), "a": "1", "b": "g"}
        for sparse in (False, True):
            rec["sparse" if sparse else "dense"] = float(
                ref_equiv.equivalent_resistance(nl, "1", "g", sparse=sparse))
        out.append(rec)
    # error paths of equivalent_resistance
    nl = ref.Netlist(f"{REF}/doc/1.6.1.csv")
    try:
        ref_equiv.equivalent_resistance(nl, "1", "g")
    except Exception as e:  # noqa: BLE001
        out.append({"name": "doc/1.6.1", "rows": rows_of_doc("1.6.1"), "a": "1", "b": "g", "error": exc_info(e)})
    nl = ref.Netlist(f"{REF}/doc/resistive_1.csv")
    try:
        ref_equiv.equivalent_resistance(nl, "1", "nope")
    except Exception as e:  # noqa: BLE001
        out.append({"name": "doc/resistive_1", "rows": rows_of_doc("resistive_1"), "a": "1", "b": "nope", "error": exc_info(e)})
    # non-"g" ground pair on resistive_3
    nl = ref.Netlist(f"{REF}/doc/resistive_3.csv")
    out.append({"name": "doc/resistive_3", "rows": rows_of_doc("resistive_3"), "a": "2", "b": "3",
                "dense": float(ref_equiv.equivalent_resistance(nl, "2", "3")),
                "sparse": float(ref_equiv.equivalent_resistance(nl, "2", "3", sparse=True))})
    for N in (4, 10, 32):
        rows = [r for r in gen.grid_rows(N)][:-1]  # resistors only
        path = write_tmp(rows)
        nl = ref.Netlist(path)
        os.unlink(path)
        out.append({"name": f"grid({N})", "gen": ["grid_resistors", N], "a": "1", "b": "g",
                    "dense": float(ref_equiv.equivalent_resistance(nl, "1", "g")),
                    "sparse": float(ref_equiv.equivalent_resistance(nl, "1", "g", sparse=True))})
    return out


SAMPLE = 64


def sample_idx(n):
    rng = np.random.RandomState(12345)
    idx = np.unique(np.concatenate([[0, 1, n // 2, n - 2, n - 1], rng.randint(0, n, SAMPLE)]))
    return idx.astype(int).tolist()


def synth_case(name, genspec, rows, full, dense_ok):
    """Medium/large synthetic case.  `full`: store whole vectors."""
    t0 = time.time()
    path = write_tmp(rows)
    nl = ref.Netlist(path)
    os.unlink(path)
    rec = {"name": name, "gen": genspec, "ground": nl.ground, "nums": dict(nl.nums),
           "ncomp": len(nl.component_keys)}
    K = nl.nums["kcl"]
    # node numbering: store a digest (full list would be large) + first/last
    items = list(nl.nodenum.items())
    rec["nodenum_head"] = items[:8]
    rec["nodenum_tail"] = items[-8:]
    rec["anomnum_head"] = list(nl.anomnum.items())[:8]
    xs = {}
    for mode in (["dense", "sparse"] if dense_ok else ["sparse"]):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            circ = ref.Circuit(nl, sparse=(mode == "sparse"))
            if mode == "sparse":
                csr = circ.G.copy()
                csr.sum_duplicates()
                csr.sort_indices()
                csr.eliminate_zeros()
                rec["nnz"] = int(csr.nnz)
                rec["G_abs_sum"] = float(np.abs(csr.data).sum())
                rec["G_diag_sum"] = float(csr.diagonal().sum())
                rec["A_sum"] = float(np.sum(circ.A))
                rec["A_nnz"] = int(np.count_nonzero(circ.A))
                if full:
                    rec["csr"] = [csr.indptr.tolist(), csr.indices.tolist(), csr.data.tolist()]
                    rec["A"] = np.asarray(circ.A).tolist()
            x = np.asarray(circ.solve().result, dtype=float)
            if mode == "sparse":
                r = csr @ x - circ.A
                rec["ref_residual_inf"] = float(np.abs(r).max())
        xs[mode] = x
    x = xs["sparse"]
    idx = sample_idx(len(x))
    rec["x_idx"] = idx
    rec["x_sparse_samples"] = x[idx].tolist()
    rec["x_sparse_sum"] = float(x.sum())
    rec["x_sparse_absmax"] = float(np.abs(x).max())
    rec["e1"] = float(x[nl.nodenum["1"]])  # = R_eq(1,g) for the plain grid
    if full:
        rec["x_sparse"] = x.tolist()
    if "dense" in xs:
        xd = xs["dense"]
        rec["x_dense_samples"] = xd[idx].tolist()
        rec["dense_vs_sparse_normwise"] = float(np.abs(xd - x).max() / np.abs(x).max())
        if full:
            rec["x_dense"] = xd.tolist()
    rec["ref_seconds"] = round(time.time() - t0, 2)
    print(f"  {name}: n={len(x)} {rec['ref_seconds']} s", flush=True)
    return rec


def synth_cases(large):
    out = []
    out.append(synth_case("grid(3)", ["grid", 3], list(gen.grid_rows(3)), True, True))
    out.append(synth_case("grid(10)", ["grid", 10], list(gen.grid_rows(10)), True, True))
    out.append(synth_case("grid(32)", ["grid", 32], list(gen.grid_rows(32)), False, True))
    out.append(synth_case("cfg4(10,b=0)", ["cfg4", 10, 0], list(gen.grid_rows(10, gen.cfg4_values(0, 10))), True, True))
    out.append(synth_case("cfg4(10,b=7)", ["cfg4", 10, 7], list(gen.grid_rows(10, gen.cfg4_values(7, 10))), True, True))
    out.append(synth_case("cfg5(16)", ["cfg5", 16, 5], gen.cfg5_rows(16), True, True))
    out.append(synth_case("cfg5(32)", ["cfg5", 32, 5], gen.cfg5_rows(32), False, True))
    out.append(synth_case("cfg5(64)", ["cfg5", 64, 5], gen.cfg5_rows(64), False, True))
    out.append(synth_case("grid(100)", ["grid", 100], list(gen.grid_rows(100)), False, False))
    out.append(synth_case("cfg4(100,b=3)", ["cfg4", 100, 3], list(gen.grid_rows(100, gen.cfg4_values(3, 100))), False, False))
    out.append(synth_case("cfg5(100)", ["cfg5", 100, 5], gen.cfg5_rows(100), False, False))
    if large:
        out.append(synth_case("grid(316)", ["grid", 316], list(gen.grid_rows(316)), False, False))
        out.append(synth_case("grid(1000)", ["grid", 1000], list(gen.grid_rows(1000)), False, False))
        out.append(synth_case("cfg5(1000)", ["cfg5", 1000, 5], gen.cfg5_rows(1000), False, False))
    return out


def dump(obj, name):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only-large", action="store_true")
    args = ap.parse_args()
    meta = {
        "reference": "EnricoMiccoli/nodal v" + ref.__version__,
        "numpy": np.__version__,
        "scipy": __import__("scipy").__version__,
        "generated_by": "tests/golden/make_golden.py",
    }
    if args.only_large:
        big = [
            synth_case("grid(316)", ["grid", 316], list(gen.grid_rows(316)), False, False),
            synth_case("grid(1000)", ["grid", 1000], list(gen.grid_rows(1000)), False, False),
            synth_case("cfg5(1000)", ["cfg5", 1000, 5], gen.cfg5_rows(1000), False, False),
        ]
        dump({"meta": meta, "cases": big}, "synth_large.json")
        return
    dump({"meta": meta, "cases": small_cases()}, "cases.json")
    dump({"meta": meta, "cases": equiv_cases()}, "equiv.json")
    dump({"meta": meta, "cases": synth_cases(args.large)}, "synth.json")


if __name__ == "__main__":
    main()
