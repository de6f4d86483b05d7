"""Drop-in import name of the reference package (EnricoMiccoli/nodal v1.3.0): `import nodal`
resolves to the MI355X implementation in `nodal_amd`, so that scripts written against the
reference -- `import nodal as n; n.Netlist(path); n.Circuit(netlist, sparse).solve()`
(reference nodal/solver.py:5,24-27) -- run unchanged.  Nothing is implemented here."""
from nodal_amd import __version__  # noqa: F401
from nodal_amd import *  # noqa: F401,F403
from nodal_amd import (  # noqa: F401  (the names the reference's `from .nodal import *` exports)
    Circuit,
    Component,
    Netlist,
    Solution,
    UnconnectedCircuitError,
    build_opmodel,
    find_ground_node,
    is_connected,
)
