"""The vectorised table builders equal the string front-end + lowering."""
import numpy as np
import pytest

import nodal_amd as n
from nodal_amd import generators as gen
from nodal_amd.lowering import lower

FIELDS = ("type", "value", "a", "b", "c", "d", "drv", "k")


def same(t1, t2):
    assert (t1.K, t1.B, t1.ncomp) == (t2.K, t2.B, t2.ncomp)
    for f in FIELDS:
        assert np.array_equal(getattr(t1, f), getattr(t2, f)), f


@pytest.mark.parametrize("N", [2, 3, 7, 20])
def test_grid_table(N):
    same(lower(n.Netlist.from_rows(gen.grid_rows(N))), gen.grid_table(N))
    vals = gen.cfg4_values(5, N)
    same(lower(n.Netlist.from_rows(gen.grid_rows(N, vals))), gen.grid_table(N, vals))


@pytest.mark.parametrize("N", [8, 32, 64])
def test_cfg5_table(N):
    same(lower(n.Netlist.from_rows(gen.cfg5_rows(N))), gen.cfg5_table(N))


def test_survey_counts():
    # validated sizes of the generators
    nl = n.Netlist.from_rows(gen.cfg5_rows(32))
    assert (nl.nums["kcl"], nl.nums["be"]) == (1038, 20)
    nl = n.Netlist.from_rows(gen.cfg5_rows(64))
    assert (nl.nums["kcl"], nl.nums["be"]) == (4142, 63)
    t = gen.grid_table(100)
    assert (t.K, t.B, t.ncomp) == (9999, 0, 19801)


def test_cfg4_values_are_stable():
    v = gen.cfg4_values(0, 10)
    assert len(v) == 180 and all(0.5 <= x < 2 for x in v)
    assert v[0] == 0.5 * 4.0 ** __import__("random").Random(1000).random()


def test_lowering_records_host_errors():
    rows = [["r1", "R", "1", "1", "g"], ["d1", "CCCS", "2", "1", "g", "1", "g", "nope"]]
    t = lower(n.Netlist.from_rows(rows))
    row, exc, probe = t.first_error
    assert row == 1 and isinstance(exc, KeyError) and probe
    rows = [["r1", "R", "0", "1", "g"]]
    row, exc, probe = lower(n.Netlist.from_rows(rows)).first_error
    assert row == 0 and isinstance(exc, ValueError) and not probe
