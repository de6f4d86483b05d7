"""Lowering: Netlist (strings) -> structure-of-arrays component table.

The table is what crosses the C ABI (`nodal_upload_components`,
include/nodal_hip.h) and what the HIP stamping kernel reads with coalesced
loads.  One table row per entry of `netlist.component_keys`, in file order:

    type  u8   device type code (constants.TYPE_CODE; VCCS -> VCVS code)
    value f64  component value as parsed by float()
    a, b  i32  nodenum of first / second lead, -1 = ground
    c, d  i32  nodenum of the control nodes, -1 = ground or not applicable
    drv   i32  table row of the driving resistor (CCVS/CCCS), else -1
    k     i32  anomnum index (branch-equation row K+k), else -1

The host-side checks reproduce, in component order, the exceptions that the
reference raises while stamping and that depend only on strings or on a single
parsed value (reference nodal/nodal.py:357-390, nodal/models.py:13-214):
null resistance, OPAMP, missing driver, control/driver node mismatch,
non-resistor driver, unknown control node.  Stamp
collisions (the reference's `assert G[i, j] == 0`) depend on matrix contents
and are detected on the device.
"""

import numpy as np

from . import constants as c


class ComponentTable:
    """SoA component table plus the sizes the kernels need."""

    __slots__ = ("type", "value", "a", "b", "c", "d", "drv", "k", "K", "B",
                 "ncomp", "first_error")

    def __init__(self, ncomp, K, B):
        self.ncomp, self.K, self.B = ncomp, K, B
        # large columns live in pinned host memory (copied to the GPU by DMA at link rate);
        # ordinary numpy arrays when small or when no HIP device is there
        from ._ffi import host_empty
        self.type = host_empty(ncomp, np.uint8)
        self.type[:] = 0
        self.value = host_empty(ncomp, np.float64)
        self.value[:] = 0.0
        for name in ("a", "b", "c", "d", "drv", "k"):
            col = host_empty(ncomp, np.int32)
            col[:] = -1
            setattr(self, name, col)
        # (row index, exception instance, probe) of the first host-detected
        # stamping error, or None.  Rows after that index are not valid; the
        # row itself is a valid "probe" row (control entries stripped) iff
        # `probe`: the reference checks that row's incidence-entry collisions
        # BEFORE it reaches the host-detected error, so the device must too.
        self.first_error = None

    @property
    def n(self):
        return self.K + self.B

    def truncated(self, ncomp):
        """A copy holding only the first `ncomp` rows (used to let the device
        look for an earlier stamp collision before a host error is raised)."""
        t = ComponentTable(ncomp, self.K, self.B)
        for name in ("type", "value", "a", "b", "c", "d", "drv", "k"):
            getattr(t, name)[:] = getattr(self, name)[:ncomp]
        return t


def _node_index(label, ground, nodenum):
    if label == ground:
        return c.GROUND
    return nodenum[label]  # KeyError(label), as the reference's nodenum[...] does


def lower(netlist):
    """Build the component table of `netlist`.

    Never raises for stamping errors: the first one is recorded in
    `table.first_error` so the caller can order it against device-detected
    collisions (see circuit.Circuit.build_model)."""
    keys = netlist.component_keys
    comps = netlist.components
    ground, nodenum, anomnum = netlist.ground, netlist.nodenum, netlist.anomnum
    nums = netlist.nums
    table = ComponentTable(len(keys), nums["kcl"], nums["be"])
    first_row = {}
    for row, key in enumerate(keys):
        first_row.setdefault(key, row)

    for row, key in enumerate(keys):
        comp = comps[key]  # a duplicated name resolves to the LAST definition
        try:
            ia = _node_index(comp.anode, ground, nodenum)
            ib = _node_index(comp.bnode, ground, nodenum)
            ctype = comp.type
            if ctype == "OPAMP":
                raise NotImplementedError
            if ctype not in c.TYPE_CODE:
                raise ValueError(f"Unknown component type: {ctype}")
            table.type[row] = c.TYPE_CODE[ctype]
            table.value[row] = comp.value
            table.a[row], table.b[row] = ia, ib
            if ctype == "R":
                if comp.value == 0:
                    raise ValueError("Model error: resistors can't have null resistance")
            elif ctype in c.NODE_TYPES_ANOM:
                table.k[row] = anomnum[comp.name]
                if ctype in c.NODE_TYPES_DEP:
                    # the reference records these on the component as a side effect
                    comp.cnode, comp.dnode = comp.pos_control, comp.neg_control
                if ctype in c.NODE_TYPES_CC:
                    _lower_current_controlled(table, row, comp, comps, first_row,
                                              ground, nodenum)
                elif ctype in c.NODE_TYPES_DEP:
                    table.c[row] = _node_index(comp.pos_control, ground, nodenum)
                    table.d[row] = _node_index(comp.neg_control, ground, nodenum)
        except (KeyError, ValueError, AssertionError, AttributeError,
                NotImplementedError, ZeroDivisionError) as exc:
            probe = comp.type in ("VCVS", "VCCS", "CCCS") and table.k[row] >= 0
            table.c[row] = table.d[row] = table.drv[row] = -1
            table.first_error = (row, exc, probe)
            break
    return table


def _lower_current_controlled(table, row, comp, comps, first_row, ground, nodenum):
    """CCVS / CCCS rows (reference nodal/models.py:109-158, 161-214)."""
    try:
        driver = comps[comp.driver]
    except KeyError:
        raise KeyError(f"Driving component {comp.driver} not found")
    if comp.type == "CCCS" and driver.type != "R":
        _raise_non_resistor_driver()
    assert comp.pos_control is not None and comp.neg_control is not None
    assert (comp.pos_control == driver.anode and comp.neg_control == driver.bnode) or (
        comp.pos_control == driver.bnode and comp.neg_control == driver.anode
    )
    if driver.type != "R":
        _raise_non_resistor_driver()
    ic = _node_index(comp.pos_control, ground, nodenum)
    id_ = _node_index(comp.neg_control, ground, nodenum)
    if driver.value == 0 and (ic != c.GROUND or id_ != c.GROUND):
        # the reference computes value / driver.value with Python floats
        raise ZeroDivisionError("float division by zero")
    table.c[row], table.d[row] = ic, id_
    table.drv[row] = first_row[comp.driver]


def _raise_non_resistor_driver():
    # The reference evaluates `c.NODE_TYPES_ANOM` on the Component argument
    # (nodal/models.py:146,200), so every non-resistor driver ends here.
    raise AttributeError("'Component' object has no attribute 'NODE_TYPES_ANOM'")
