"""The drop-in surface: `import nodal` (reference nodal/__init__.py:1-3) and the two console
scripts of the reference's pyproject.toml:14-16 resolve to the MI355X implementation."""
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_import_nodal_exports_the_reference_names():
    import nodal
    import nodal_amd
    assert nodal.__version__ == "1.3.0"
    # reference nodal/nodal.py: every public name `from .nodal import *` brings in
    for name in ("Netlist", "Circuit", "Solution", "Component", "UnconnectedCircuitError",
                 "find_ground_node", "build_opmodel", "is_connected"):
        assert getattr(nodal, name) is getattr(nodal_amd, name)
    for sub in ("nodal", "solver", "equiv", "constants"):
        importlib.import_module(f"nodal.{sub}")
    import nodal.constants as c
    assert c.OPMODEL_RI == 1e7 and c.OPMODEL_RO == 10 and c.OPMODEL_GAIN == 1e5  # reference constants.py:36-38
    assert "VCCS" in c.NODE_TYPES_ANOM and "OPMODEL" in c.NODE_TYPES


def test_console_scripts_point_at_the_reference_entry_points():
    text = open(os.path.join(ROOT, "pyproject.toml")).read()
    scripts = dict(re.findall(r'^(nodal-[a-z]+) = "([\w.:]+)"', text, flags=re.M))
    assert scripts == {"nodal-solver": "nodal.solver:main", "nodal-resistance": "nodal.equiv:main"}
    for target in scripts.values():
        module, func = target.split(":")
        assert callable(getattr(importlib.import_module(module), func))


def test_reference_style_caller_parses_without_a_gpu(tmp_path):
    """The front end half of reference nodal/solver.py:24 runs anywhere; Circuit needs the GPU."""
    import nodal as n
    path = tmp_path / "c.csv"
    path.write_text("r1,R,2,1,g\na1,A,1,1,g\n")
    netlist = n.Netlist(str(path))
    assert netlist.ground == "g" and netlist.nums["kcl"] == 1


@pytest.mark.gpu
def test_reference_solver_script_runs_unchanged(tmp_path, capsys):
    """The body of the reference's solver.main (nodal/solver.py:16-31) written against
    `import nodal as n`, on the reference's own doc/1.6.1 netlist."""
    import nodal as n
    from tests.conftest import load_golden
    from nodal_amd import generators as gen
    case = next(c for c in load_golden("cases.json") if c["name"] == "doc/1.6.1")
    path = tmp_path / "1.6.1.csv"
    gen.write_csv(case["rows"], str(path))
    for sparse in (False, True):
        netlist = n.Netlist(str(path))
        circuit = n.Circuit(netlist, sparse=sparse)
        solution = circuit.solve()
        print(solution)
        assert capsys.readouterr().out == case["sparse" if sparse else "dense"]["str"] + "\n"
    from nodal.solver import main
    main(["-s", str(path)])
    assert capsys.readouterr().out == case["sparse"]["str"] + "\n"
