"""Netlist front-end: CSV rows -> components, node numbering, ground choice.

This is the input contract of the hot path (SURVEY.md section 8a rows A1-A4,
A14).  It stays on the host and works on strings; its integer outputs
(`nodenum`, `anomnum`, `ground`, `component_keys`) must be bit-exact with the
reference because they decide where every stamp lands in G.

Behaviour mirrored from the reference (file:line into /root/reference):
  * Component / check_input ........ nodal/nodal.py:112-178
  * Netlist.process_component ...... nodal/nodal.py:222-257
  * Netlist.read_netlist ........... nodal/nodal.py:259-296
  * find_ground_node ............... nodal/nodal.py:30-42
  * build_opmodel .................. nodal/nodal.py:45-85
  * is_connected ................... nodal/nodal.py:88-105
"""

import csv
import logging
import os
from collections import deque

from . import constants as c

logging.basicConfig(level=logging.ERROR)


FAST_PARSE_MIN_BYTES = 1 << 18  # files this large use the vectorised reader


class UnconnectedCircuitError(Exception):
    """Raised by Circuit.solve() when floating nodes make G singular."""


def find_ground_node(degrees):
    """Pick the reference node: "g" if present, else the first node (in
    insertion order) of maximal degree (reference nodal/nodal.py:30-42)."""
    if "g" in degrees:
        return "g"
    best, best_deg = None, None
    for node, deg in degrees.items():
        if best_deg is None or deg > best_deg:  # strict: first maximum wins
            best, best_deg = node, deg
    if best is None:
        # same exception type/message as max() over an empty dict
        raise ValueError("max() arg is an empty sequence")
    return best


def build_opmodel(data):
    """Expand an OPMODEL row into its equivalent rows, in the order
    [input R, output R, VCVS, (feedback R)] (reference nodal/nodal.py:45-85).

    Row layout: [name, "OPMODEL", feedback ohms, out, gnd, non-inverting,
    inverting].  A feedback value of exactly the string "0" means a direct
    wire, which requires inverting == out.
    """
    name, rf = data[c.NCOL], data[c.VCOL]
    out, gnd, pos, neg = data[c.ACOL], data[c.BCOL], data[c.CCOL], data[c.DCOL]
    inner = f"{name}_internal_node"
    rows = [
        [f"{name}_ri", "R", str(c.OPMODEL_RI), pos, neg],
        [f"{name}_ro", "R", str(c.OPMODEL_RO), inner, out],
        [f"{name}_vcvs", "VCVS", str(c.OPMODEL_GAIN), inner, gnd, pos, neg],
    ]
    if rf != "0":
        rows.append([f"{name}_rf", "R", rf, neg, out])
    else:
        assert neg == out
    return rows


class Component:
    """One electrical component parsed from a CSV row.

    Attributes: name, type, value (float), anode, bnode, pos_control,
    neg_control and, for dependent sources, driver (reference
    nodal/nodal.py:130-148).  Raises ValueError on a malformed row.
    """

    def __init__(self, data):
        self.check_input(data)
        self.name = data[c.NCOL]
        self.type = data[c.TCOL]
        self.value = float(data[c.VCOL])
        self.anode = data[c.ACOL]
        self.bnode = data[c.BCOL]
        self.pos_control = None
        self.neg_control = None
        if self.type in c.NODE_TYPES_DEP:
            self.pos_control = data[c.CCOL]
            self.neg_control = data[c.DCOL]
            # only dependent sources carry the attribute at all
            self.driver = data[c.PCOL] if self.type in c.NODE_TYPES_CC else None

    def check_input(self, data):
        """Validate a row; callable unbound with self=None as the reference's
        tests do (reference tests.py:10-11, nodal/nodal.py:150-178)."""
        nfields = len(data)
        if nfields == 0 or data[0][0] == "#":
            return
        key = data[c.NCOL]
        assert type(key) == str
        if nfields < 5:
            raise ValueError(f"Missing arguments for component {key}")
        ctype = data[c.TCOL]
        if ctype not in c.NODE_TYPES:
            raise ValueError(f"Unknown type {ctype} for component {key}")
        expected = c.NODE_ARGS_NUMBER[ctype]
        if nfields != expected:
            raise ValueError(
                f"Wrong number of arguments for component {key}: "
                f"expected {expected}, got {nfields}"
            )
        try:
            float(data[c.VCOL])
        except ValueError:
            raise ValueError(
                "Bad input: expected a number for component value "
                f"of {key}, got {data[c.VCOL]} instead"
            )


class Netlist:
    """Reads a netlist from a .csv file.

    Attributes (same names and meaning as the reference, nodal/nodal.py:181-220):
    nums, degrees, anomnum, components, component_keys, ground, nodenum,
    opmodel_equivalents.  `Netlist.from_rows(rows)` builds one from already
    split rows (used by the synthetic generators and tests).
    """

    def __init__(self, path):
        self._reset()
        self.read_netlist(path)

    def _reset(self):
        self.nums = {"components": 0, "anomalies": 0, "be": 0, "kcl": 0, "opamps": 0}
        self.degrees = {}
        self.anomnum = {}
        self.components = {}
        self.component_keys = []
        self.ground = None
        self.nodenum = {}
        self.opmodel_equivalents = []

    def __getattr__(self, name):
        # natively read netlists build their public containers when somebody asks for them (fastparse.materialise)
        if self.__dict__.get("_fast") and "_names_blob" in self.__dict__:
            from . import fastparse
            if name in fastparse.LAZY_ATTRIBUTES:
                return fastparse.materialise(self, name)
        # vectorised netlists build their name -> row map only when somebody asks for it
        if name == "_row_of" and self.__dict__.get("_fast"):
            row_of = dict(zip(self.component_keys, range(len(self.component_keys))))
            self.__dict__["_row_of"] = row_of
            return row_of
        raise AttributeError(name)

    def __deepcopy__(self, memo):
        """equivalent_resistance deep-copies the netlist before adding its probe source.  A
        vectorised netlist keeps its big columns in arrays that are REPLACED, never mutated,
        when a row is appended (fastparse.append_row), so the copy shares them and only the
        small mutable containers are duplicated (2e6 rows: 0.3 s instead of 8 s)."""
        import copy
        if not self.__dict__.get("_fast"):
            clone = self.__class__.__new__(self.__class__)
            memo[id(self)] = clone
            for key, value in self.__dict__.items():
                clone.__dict__[key] = copy.deepcopy(value, memo)
            return clone
        clone = self.__class__.__new__(self.__class__)
        memo[id(self)] = clone
        shared = {"_raw", "_line_off", "_line_len", "_tidx", "_df", "_name", "_type", "_value", "_nfields",
                  "_acode", "_bcode", "_is_anom", "_node_index", "_names_blob", "_labels_blob", "_deg",
                  "_names_list", "_labels"}
        for key, value in self.__dict__.items():
            if key == "_row_of":
                continue  # rebuilt on demand
            if key in shared:
                clone.__dict__[key] = value
            elif key == "components":
                from . import fastparse
                comps = fastparse.LazyComponents(clone)
                comps._cache = dict(value._cache)
                clone.__dict__[key] = comps
            elif isinstance(value, dict):
                clone.__dict__[key] = dict(value)
            elif isinstance(value, list):
                clone.__dict__[key] = [list(v) if isinstance(v, list) else v for v in value]
            else:
                clone.__dict__[key] = copy.deepcopy(value, memo)
        return clone

    @classmethod
    def from_rows(cls, rows):
        self = cls.__new__(cls)
        self._reset()
        self._ingest(rows)
        return self

    def process_component(self, data):
        """Register one row: component table, degrees, anomalous numbering
        (reference nodal/nodal.py:222-257)."""
        if data == [] or data[0][0] == "#":
            return
        if getattr(self, "_fast", False):
            from . import fastparse
            try:
                return fastparse.append_row(self, data)
            except fastparse.Irregular:
                self._demote()
        if data[c.TCOL] == "OPMODEL":
            # macro rows are queued and only registered after every file row
            self.opmodel_equivalents.extend(build_opmodel(data))
            return
        comp = Component(data)
        key = data[c.NCOL]
        self.component_keys.append(key)  # duplicates keep two key entries
        self.components[key] = comp  # ... but the later object wins
        self.nums["components"] += 1
        leads = (data[c.ACOL], data[c.BCOL])
        fresh = [node for node in leads if node not in self.degrees]
        if data[c.TCOL] in c.NODE_TYPES_ANOM:
            self.anomnum[key] = self.nums["anomalies"]
            self.nums["anomalies"] += 1
        for node in fresh:  # first appearance order, anode before bnode
            self.degrees[node] = 0
        for node in leads:
            self.degrees[node] += 1

    def read_netlist(self, path):
        """Parse the file at `path` (reference nodal/nodal.py:259-296).

        Large regular files go through the vectorised reader of fastparse.py, which
        produces the same attributes; anything it does not recognise is re-read by the
        exact row-by-row parser below."""
        try:
            infile = open(path, "r")
        except FileNotFoundError:
            logging.error(f"File '{path}' not found.")
            raise
        with infile:
            if os.path.getsize(path) >= FAST_PARSE_MIN_BYTES:
                from . import fastparse
                try:
                    fastparse.read_fast(self, path)
                    return
                except fastparse.Irregular:
                    self._reset()
            self._ingest(csv.reader(infile, skipinitialspace=True))

    def _row_fields(self, row):
        from . import fastparse
        return fastparse.row_fields(self, row)

    def _demote(self):
        """Turn a vectorised netlist back into a row-by-row one (rare: a row was added
        that needs the reference's duplicate / macro bookkeeping)."""
        rows = [self._row_fields(i) for i in range(len(self._name))]
        ground, nodenum, nums = self.ground, self.nodenum, dict(self.nums)
        self._fast = False
        self._reset()
        for data in rows:
            self.process_component(data)
        # numbering is NOT recomputed by process_component in the reference either
        self.ground, self.nodenum = ground, nodenum
        self.nums["kcl"], self.nums["be"] = nums["kcl"], nums["be"]

    def _ingest(self, rows):
        for data in rows:
            self.process_component(data)
        for data in self.opmodel_equivalents:
            self.process_component(data)
        self.ground = find_ground_node(self.degrees)
        self.nodenum = {}
        for node in self.degrees:
            if node != self.ground:
                self.nodenum[node] = len(self.nodenum)
        assert len(self.nodenum) == len(self.degrees) - 1
        self.nums["kcl"] = len(self.nodenum)
        self.nums["be"] = self.nums["anomalies"]


def is_connected(netlist):
    """True iff every node is reachable from ground through component leads
    (control nodes do not count; reference nodal/nodal.py:88-105).

    Same answer as the reference's BFS, but with an O(V+E) visited set
    instead of its O(V^2) list membership test.
    """
    if getattr(netlist, "_fast", False):
        return _is_connected_fast(netlist)
    neighbours = {node: set() for node in netlist.degrees}
    for comp in netlist.components.values():
        neighbours[comp.anode].add(comp.bnode)
        neighbours[comp.bnode].add(comp.anode)
    assert len(neighbours) == len(netlist.degrees)
    seen = {netlist.ground}
    frontier = deque(seen)
    while frontier:
        for nxt in neighbours[frontier.popleft()]:
            if nxt not in seen:
                seen.add(nxt)
                frontier.append(nxt)
    return len(seen) == len(netlist.degrees)


def _is_connected_fast(netlist):
    """is_connected for a vectorised netlist: union-find by repeated min-label
    propagation over the lead pairs (numpy), no per-component Python objects."""
    import numpy as np
    a, b = netlist._acode.astype(np.int64), netlist._bcode.astype(np.int64)
    n = len(netlist.degrees)
    label = np.arange(n, dtype=np.int64)
    while True:
        la, lb = label[a], label[b]
        lo = np.minimum(la, lb)
        new = label.copy()
        np.minimum.at(new, la, lo)
        np.minimum.at(new, lb, lo)
        new = new[new]  # pointer jumping
        if np.array_equal(new, label):
            break
        label = new
    while True:  # full compression
        nxt = label[label]
        if np.array_equal(nxt, label):
            break
        label = nxt
    return bool((label == label[0]).all()) if n else True
