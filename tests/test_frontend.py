"""Host front-end vs the reference's outputs (golden vectors) and the
reference's own unit-test groups (reference tests.py:125-216)."""
import io

import pytest

import nodal_amd as n
from nodal_amd import equiv
from tests.conftest import load_golden

CASES = load_golden("cases.json")


def parse(case):
    if case.get("raw_text") is not None:
        import csv
        return n.Netlist.from_rows(csv.reader(io.StringIO(case["raw_text"]), skipinitialspace=True))
    return n.Netlist.from_rows(case["rows"])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_front_end_matches_reference(case):
    if "parse_error" in case:
        exc = {"ValueError": ValueError, "IndexError": IndexError}[case["parse_error"]["type"]]
        with pytest.raises(exc) as info:
            parse(case)
        assert [str(a) for a in info.value.args] == case["parse_error"]["args"]
        return
    nl = parse(case)
    assert nl.ground == case["ground"]
    assert [[k, v] for k, v in nl.degrees.items()] == case["degrees"]  # order matters
    assert [[k, v] for k, v in nl.nodenum.items()] == case["nodenum"]
    assert [[k, v] for k, v in nl.anomnum.items()] == case["anomnum"]
    assert nl.component_keys == case["component_keys"]
    assert nl.nums == case["nums"]
    assert n.is_connected(nl) == case["is_connected"]


def test_netlist_from_file(tmp_path):
    path = tmp_path / "c.csv"
    path.write_text("# comment\n\nr1, R, 1, 1, 2\nr2, R, 1, 2, g\n")
    nl = n.Netlist(str(path))
    assert nl.nodenum == {"1": 0, "2": 1} and nl.ground == "g"
    with pytest.raises(FileNotFoundError):
        n.Netlist(str(tmp_path / "missing.csv"))


# --- reference tests.py:125-185 (InputTesters) ---------------------------------
BAD = ["aaaaa", "v1,VCVS,5,1,2", "v1,VCCS,5,1,2", "v1,CCVS,5,1,2", "v1,CCCS,5,1,2",
       "q1,OPMODEL,0,2,g,3", "v1,VCVS,5,1,2,1,1,1", "r1,R,5,1,2,3", "r1,A,5,1,2,3",
       "r1,E,5,1,2,3", "q1,OPMODEL,1,2,g,3,1,5", "v1,VoltageSource,5,1,2", "r1,R,one_ohm,1,2"]
GOOD = ["r1,R,2,1,4", "r2,R,2,1,g", "r3,R,0.5,1,2", "e1,E,8,4,g", "a1,A,4,1,2",
        "d1,CCCS,2,2,g,1,g,r2", "Ri,R,1e7,1,3", "Ro,R,1e1,1,2", "vs,E,10,3,g",
        "d1,VCVS,1e5,2,g,3,1", "q1,OPMODEL,1,2,g,3,1", "q1,OPMODEL,0,2,g,3,2"]


@pytest.mark.parametrize("row", BAD)
def test_check_input_rejects(row):
    with pytest.raises(ValueError):
        n.Component.check_input(None, row.split(","))


@pytest.mark.parametrize("row", GOOD)
def test_check_input_accepts(row):
    n.Component.check_input(None, row.split(","))


def test_check_input_empty_and_comment():
    n.Component.check_input(None, [])
    n.Component.check_input(None, "# This is a comment")


# --- reference tests.py:188-202 (GroundNode) -------------------------------------
@pytest.mark.parametrize("deg,expected", [
    ({"g": 1}, "g"), ({"g": 1, "a": 10, "b": 2}, "g"), ({"1": 1}, "1"),
    ({"3": 1, "a": 10, "b": 2}, "a"), ({"1": 1, "2": 1}, "1"), ({"3": 1, "a": 10, "b": 10}, "a")])
def test_find_ground_node(deg, expected):
    assert n.find_ground_node(deg) == expected


def test_build_opmodel_rows():
    rows = n.build_opmodel(["q1", "OPMODEL", "1", "2", "g", "3", "1"])
    assert rows == [["q1_ri", "R", "10000000.0", "3", "1"], ["q1_ro", "R", "10", "q1_internal_node", "2"],
                    ["q1_vcvs", "VCVS", "100000.0", "q1_internal_node", "g", "3", "1"],
                    ["q1_rf", "R", "1", "1", "2"]]
    assert len(n.build_opmodel(["q1", "OPMODEL", "0", "2", "g", "3", "2"])) == 3
    with pytest.raises(AssertionError):
        n.build_opmodel(["q1", "OPMODEL", "0", "2", "g", "3", "1"])


# --- reference tests.py:16-36 (check_resistive) ----------------------------------
def test_check_resistive():
    by_name = {c["name"]: c for c in CASES}
    expect = {"doc/resistive_1": True, "doc/resistive_2": True, "doc/1.6.1": False,
              "doc/netlist": False, "doc/opmodel_amplifier": False}
    for name, want in expect.items():
        assert equiv.check_resistive(parse(by_name[name])) is want


def test_equivalent_resistance_argument_errors():
    by_name = {c["name"]: c for c in CASES}
    with pytest.raises(ValueError, match="Network is not resistive"):
        equiv.equivalent_resistance(parse(by_name["doc/1.6.1"]), "1", "g")
    with pytest.raises(KeyError, match="Node `nope` not found in netlist"):
        equiv.equivalent_resistance(parse(by_name["doc/resistive_1"]), "1", "nope")


def test_circuit_rejects_non_netlist():
    with pytest.raises(TypeError, match="Input isn't a netlist"):
        n.Circuit("not a netlist")
