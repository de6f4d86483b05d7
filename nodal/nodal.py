"""`nodal.nodal` of the reference (nodal/nodal.py): the same names, from nodal_amd."""
from nodal_amd.circuit import Circuit, Solution  # noqa: F401
from nodal_amd.netlist import (  # noqa: F401
    Component,
    Netlist,
    UnconnectedCircuitError,
    build_opmodel,
    find_ground_node,
    is_connected,
)
