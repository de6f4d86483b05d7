"""MI355X-native modified-nodal-analysis solver with Nodal.py's Python API."""
__version__ = "1.3.0"
from .netlist import (  # noqa: F401
    Component,
    Netlist,
    UnconnectedCircuitError,
    build_opmodel,
    find_ground_node,
    is_connected,
)
from .circuit import Circuit, Solution  # noqa: E402,F401
