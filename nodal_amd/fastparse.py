"""Vectorised netlist front-end for large, regular CSV files.

The reference parses row by row into Python objects and dicts (reference
nodal/nodal.py:222-296): 19.7 s for the 2e6-row grid(1000) netlist, two orders of
magnitude more than the GPU path needs for the same circuit.  `read_fast` produces
the SAME observable state -- `component_keys`, `degrees`, `ground`, `nodenum`,
`anomnum`, `nums`, `components` -- from whole-column operations:

  * pandas' C reader splits the file (skipinitialspace, quoted fields, ragged rows);
  * `factorize` on the interleaved (anode, bnode) labels yields the nodes in order of
    first appearance, anode before bnode -- exactly the dict insertion order the
    reference's node numbering depends on;
  * `components` is a lazy mapping that builds `Component` objects on demand.

Anything irregular -- malformed rows, OPMODEL macros, duplicated names, fields that
need the reference's exact exception -- raises `Irregular`, and `Netlist` falls back
to the exact row-by-row parser, which then reports the problem with the reference's
own message.  `lower_fast` is the matching vectorised lowering (the component table
of lowering.py) for such a netlist.
"""

from collections.abc import Mapping

import numpy as np

from . import constants as c

try:
    import pandas as pd
except Exception:  # pragma: no cover - pandas is optional
    pd = None


class Irregular(Exception):
    """The file needs the exact row-by-row parser."""


_NARGS = c.NODE_ARGS_NUMBER


class LazyComponents(Mapping):
    """name -> Component, materialised on demand from the parsed columns."""

    def __init__(self, netlist):
        self._nl = netlist
        self._cache = {}

    def __deepcopy__(self, memo):  # equivalent_resistance deep-copies the netlist
        import copy
        clone = LazyComponents(memo.get(id(self._nl), self._nl))
        clone._cache = copy.copy(self._cache)
        return clone

    def __getitem__(self, key):
        comp = self._cache.get(key)
        if comp is None:
            row = self._nl._row_of.get(key)
            if row is None:
                raise KeyError(key)
            from .netlist import Component
            comp = Component(self._nl._row_fields(row))
            self._cache[key] = comp
        return comp

    def __iter__(self):
        return iter(self._nl.component_keys)  # names are unique on a vectorised netlist

    def __len__(self):
        return len(self._nl.component_keys)


# ---- native tokenizer (csrc/fastcsv.cpp -> libnodal_csv.so), optional -------------------------

_TYPE_NAMES = np.array(["R", "A", "E", "VCVS", "VCCS", "CCVS", "CCCS"], dtype=object)
_CSV_REASONS = {1: "quoted fields", 2: "whitespace-only line", 3: "empty first field",
                4: "row with more than 8 fields", 5: "unknown or macro type", 6: "wrong number of arguments",
                7: "component value spelling", 8: "duplicated component names", 9: "no components",
                10: "carriage return inside a line", 11: "out of memory"}
_csv_lib = None


def _load_csv_lib():
    global _csv_lib
    if _csv_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnodal_csv.so")
        if not os.path.exists(path):
            _csv_lib = False
            return False

        class Result(C.Structure):
            _fields_ = [("nrows", C.c_int64), ("nnodes", C.c_int64), ("status", C.c_int32),
                        ("bad_line", C.c_int64), ("line_off", C.POINTER(C.c_int64)),
                        ("line_len", C.POINTER(C.c_int32)), ("type_idx", C.POINTER(C.c_uint8)),
                        ("nfields", C.POINTER(C.c_uint8)), ("value", C.POINTER(C.c_double)),
                        ("acode", C.POINTER(C.c_int32)), ("bcode", C.POINTER(C.c_int32)),
                        ("names_blob", C.POINTER(C.c_char)), ("names_bytes", C.c_int64),
                        ("labels_blob", C.POINTER(C.c_char)), ("labels_bytes", C.c_int64)]

        lib = C.CDLL(path)
        lib.nodal_csv_parse.restype = C.c_int
        lib.nodal_csv_parse.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Result)]
        lib.nodal_csv_free.restype = None
        lib.nodal_csv_free.argtypes = [C.POINTER(Result)]
        lib.Result = Result
        lib.nodal_repr_double.restype = C.c_int
        lib.nodal_repr_double.argtypes = [C.c_double, C.c_char_p]
        lib.nodal_format_lines.restype = C.c_int
        lib.nodal_format_lines.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                           C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.nodal_csv_free_buffer.restype = None
        lib.nodal_csv_free_buffer.argtypes = [C.c_void_p]
        _csv_lib = lib
    return _csv_lib


def native_potential_lines(netlist, values):
    """The "e(name) \t= value" lines of Solution.__str__ (reference nodal/nodal.py:422-434: names in sorted() order,
    values as str(np.float64) prints them) for a natively read netlist, made by libnodal_csv.so on the host threads from
    the tokenizer's label blob; None when that does not apply (the caller formats them in Python)."""
    d = netlist.__dict__
    if not d.get("_fast") or "_labels_blob" not in d:
        return None
    if "nodenum" in d and len(d["nodenum"]) != d["_nlabels"] - 1:
        return None  # (somebody changed the dict: its own contents are what counts)
    lib = _load_csv_lib()
    if not lib:
        return None
    import ctypes as C
    index_of = np.ascontiguousarray(d["_node_index"][: d["_nlabels"]], dtype=np.int64)
    values = np.ascontiguousarray(values, dtype=np.float64)
    if len(values) < d["_nlabels"] - 1:
        return None
    out, out_len = C.c_void_p(), C.c_int64(0)
    blob = d["_labels_blob"]
    status = lib.nodal_format_lines(blob, len(blob), d["_nlabels"], index_of.ctypes.data_as(C.POINTER(C.c_int64)),
                                    values.ctypes.data_as(C.POINTER(C.c_double)), b"e(", C.byref(out), C.byref(out_len))
    if status != 0:
        return None
    try:
        return C.string_at(out, out_len.value).decode()
    finally:
        lib.nodal_csv_free_buffer(out)


def _finish_fast(nl, name, type_names, value, nfields, acode, bcode, labels):
    """Common tail of both readers: the Netlist attributes from the parsed columns."""
    deg = np.bincount(np.concatenate((acode, bcode)), minlength=len(labels))
    nl._fast = True
    nl._name, nl._type, nl._value = name, type_names, value
    nl._nfields = nfields
    nl._acode, nl._bcode = acode, bcode
    nl.component_keys = name.tolist()
    nl.__dict__.pop("_row_of", None)  # name -> row, built on first use (Netlist.__getattr__)
    nl.components = LazyComponents(nl)
    nl.degrees = dict(zip(labels, deg.tolist()))
    is_anom = np.isin(type_names.astype(str), c.NODE_TYPES_ANOM) if not hasattr(nl, "_tidx") \
        else (nl._tidx >= 2)  # everything after R, A owns a branch current
    nl._is_anom = is_anom
    anom_rows = np.flatnonzero(is_anom)
    nl.anomnum = dict(zip(name[anom_rows].tolist(), range(len(anom_rows))))
    nl.nums["components"] = int(len(name))
    nl.nums["anomalies"] = int(len(anom_rows))
    # ground: "g" if present, else the first node of maximal degree
    if "g" in nl.degrees:
        nl.ground = "g"
        gcode = labels.index("g")
    else:
        gcode = int(np.argmax(deg))
        nl.ground = labels[gcode]
    order = labels[:gcode] + labels[gcode + 1:]
    nl.nodenum = dict(zip(order, range(len(order))))
    node_index = np.arange(len(labels), dtype=np.int64)
    node_index[gcode + 1:] -= 1
    node_index[gcode] = -1
    nl._node_index = node_index  # label code -> nodenum index (-1 = ground)
    nl.nums["kcl"] = len(order)
    nl.nums["be"] = nl.nums["anomalies"]
    return nl


def _read_native(netlist, path, lib):
    import ctypes as C
    with open(path, "rb") as f:
        raw = f.read()
    res = lib.Result()
    status = lib.nodal_csv_parse(raw, len(raw), C.byref(res))
    try:
        if status != 0:
            raise Irregular(f"{_CSV_REASONS.get(status, status)} (line {res.bad_line + 1})")
        n, m = res.nrows, res.nnodes

        def arr(ptr, count):  # one copy out of the library's buffers (freed below)
            return np.array(np.ctypeslib.as_array(ptr, shape=(count,)))

        line_off = arr(res.line_off, n)
        line_len = arr(res.line_len, n)
        tidx = arr(res.type_idx, n)
        nfields = arr(res.nfields, n)
        value = arr(res.value, n)
        acode = arr(res.acode, n)
        bcode = arr(res.bcode, n)
        names_blob = C.string_at(res.names_blob, res.names_bytes)
        labels_blob = C.string_at(res.labels_blob, res.labels_bytes)
        try:  # (validation only: the strings are made when somebody asks for them)
            if not names_blob.isascii():
                names_blob.decode()
            if not labels_blob.isascii():
                labels_blob.decode()
        except UnicodeDecodeError:
            raise Irregular("not UTF-8")
        if names_blob.count(b"\n") + 1 != n or labels_blob.count(b"\n") + 1 != m:
            raise Irregular("line structure")
    finally:
        lib.nodal_csv_free(C.byref(res))
    nl = netlist
    nl._df = None
    nl._raw, nl._line_off, nl._line_len = raw, line_off, line_len
    nl._tidx = tidx
    return _finish_native(nl, names_blob, labels_blob, n, m, value, nfields, acode, bcode)


# ---- the public containers of a natively read netlist, built when somebody asks for them --------------------
# Round 5: `Netlist(path)` of the 2e6-row grid(1000) file spent as long building Python objects -- two million name
# strings, a million labels, the degrees and nodenum dicts -- as tokenizing the file, and `Circuit(netlist)` +
# `.solve()` read none of them (the lowering works on the integer columns).  The native reader keeps the two string
# blobs and the columns; `component_keys`, `degrees`, `nodenum`, `anomnum` (and the private name array) are made on
# first access (Netlist.__getattr__), with exactly the contents the eager build had: same keys, same insertion
# order, same values.  `ground`, `nums` and the integer columns are there at once.
LAZY_ATTRIBUTES = ("component_keys", "degrees", "nodenum", "anomnum", "_name", "_labels", "_names_list")


def _finish_native(nl, names_blob, labels_blob, n, m, value, nfields, acode, bcode):
    deg = np.bincount(acode, minlength=m) + np.bincount(bcode, minlength=m)
    nl._fast = True
    nl._names_blob, nl._labels_blob = names_blob, labels_blob
    nl._nrows_file, nl._nlabels = int(n), int(m)
    nl._deg = deg
    nl._type, nl._value = _TYPE_NAMES[nl._tidx], value
    nl._nfields = nfields
    nl._acode, nl._bcode = acode, bcode
    for key in LAZY_ATTRIBUTES + ("_row_of",):
        nl.__dict__.pop(key, None)
    nl.components = LazyComponents(nl)
    is_anom = nl._tidx >= 2  # everything after R, A owns a branch current
    nl._is_anom = is_anom
    nanom = int(np.count_nonzero(is_anom))
    nl.nums["components"] = int(n)
    nl.nums["anomalies"] = nanom
    # ground: "g" if present, else the first node of maximal degree (labels are unique: one whole-label match at most)
    lb = labels_blob
    if lb == b"g" or lb.startswith(b"g\n"):
        gcode = 0
    else:
        at = lb.find(b"\ng\n")
        if at < 0 and lb.endswith(b"\ng"):
            at = len(lb) - 2
        gcode = lb.count(b"\n", 0, at + 1) if at >= 0 else -1
    if gcode >= 0:
        nl.ground = "g"
    else:
        gcode = int(np.argmax(deg))
        nl.ground = nl._labels[gcode]
    nl._gcode = gcode
    node_index = np.arange(m, dtype=np.int64)
    node_index[gcode + 1:] -= 1
    node_index[gcode] = -1
    nl._node_index = node_index  # label code -> nodenum index (-1 = ground)
    nl.nums["kcl"] = m - 1
    nl.nums["be"] = nanom
    return nl


def materialise(nl, name):
    """One of LAZY_ATTRIBUTES of a natively read netlist (called by Netlist.__getattr__, once per attribute)."""
    d = nl.__dict__
    if name == "_names_list":
        value = d["_names_blob"].decode().split("\n")
    elif name == "_labels":
        value = d["_labels_blob"].decode().split("\n")
    elif name == "component_keys":
        value = list(nl._names_list)
    elif name == "_name":
        value = np.empty(d["_nrows_file"], dtype=object)
        value[:] = nl._names_list
    elif name == "degrees":
        value = dict(zip(nl._labels, d["_deg"].tolist()))
    elif name == "nodenum":
        labels, gcode = nl._labels, d["_gcode"]
        order = labels[:gcode] + labels[gcode + 1:]
        value = dict(zip(order, range(len(order))))
    elif name == "anomnum":
        rows = np.flatnonzero(d["_is_anom"][: d["_nrows_file"]])
        if len(rows) == 0:
            value = {}
        else:
            names = nl._names_list
            value = {names[r]: k for k, r in enumerate(rows.tolist())}
    else:
        raise AttributeError(name)
    d[name] = value
    return value


def read_fast(netlist, path):
    """Fill `netlist` (a Netlist whose `_reset()` has run) from the CSV at `path`."""
    import gc
    # millions of small objects (names, labels, dict entries) are created and all survive:
    # the cyclic collector would only rescan them over and over
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _read_fast(netlist, path)
    finally:
        if was_enabled:
            gc.enable()


def _read_fast(netlist, path):
    lib = _load_csv_lib()
    if lib:
        return _read_native(netlist, path, lib)
    if pd is None:
        raise Irregular("pandas not available")
    # fields per line, counted on the raw bytes: pandas pads short rows silently, the
    # reference rejects them, so the count has to come from the text itself
    with open(path, "rb") as f:
        raw = f.read()
    if b'"' in raw:
        raise Irregular("quoted fields")  # commas inside quotes: let csv.reader decide
    buf = np.frombuffer(raw, dtype=np.uint8)
    ends = np.flatnonzero(buf == 10)
    if len(buf) and (len(ends) == 0 or ends[-1] != len(buf) - 1):
        ends = np.append(ends, len(buf))  # last line without a newline
    starts = np.concatenate(([0], ends[:-1] + 1))
    length = ends - starts
    # segment reductions over the byte buffer (reduceat on an empty segment returns the
    # element at its start, so empty lines are masked out explicitly)
    seg = np.minimum(starts, len(buf) - 1)
    line_commas = np.add.reduceat((buf == 44).view(np.uint8), seg, dtype=np.int32)
    nonblank = ((buf != 32) & (buf != 9) & (buf != 13) & (buf != 10)).view(np.uint8)
    content = np.maximum.reduceat(nonblank, seg)
    content[length == 0] = 0
    line_commas[length == 0] = 0
    only_cr = (length == 1) & (buf[np.minimum(starts, len(buf) - 1)] == 13)
    if ((content == 0) & (length > 0) & ~only_cr).any():
        raise Irregular("whitespace-only line")  # the reference raises IndexError there
    line_fields = line_commas[content > 0] + 1
    if len(line_fields) == 0 or line_fields.max() > 8:
        raise Irregular("row with more than 8 fields")  # (or a comment line full of commas)
    try:
        df = pd.read_csv(path, header=None, names=list(range(8)), dtype=str, engine="c",
                         skipinitialspace=True, keep_default_na=False, na_values=[],
                         skip_blank_lines=True, quotechar='"')
    except Exception as exc:  # tokenising error: let the exact parser explain it
        raise Irregular(str(exc))
    if len(df) == 0 or len(df) != len(line_fields):
        raise Irregular("line structure")
    if (df[0].str.len() == 0).any():
        raise Irregular("empty first field")
    # comment rows: first field starts with '#'
    keep = (df[0].str.slice(0, 1) != "#").to_numpy()
    if not keep.all():
        df = df[keep].reset_index(drop=True)
        line_fields = line_fields[keep]
    if len(df) == 0:
        raise Irregular("no components")
    name = df[0].to_numpy(dtype=object)
    nfields = line_fields
    ctype = df[1].to_numpy(dtype=object)
    kinds, kind_code = np.unique(ctype.astype(str), return_inverse=True)
    for kd in kinds:
        if kd not in _NARGS or kd in ("OPMODEL", "OPAMP"):
            raise Irregular(f"type {kd}")
    expected = np.array([_NARGS[kd] for kd in kinds])[kind_code]
    if not (nfields == expected).all():
        raise Irregular("wrong number of arguments")
    try:
        value = df[2].astype(np.float64).to_numpy()
    except Exception:
        raise Irregular("bad component value")
    if not pd.Index(name).is_unique:
        raise Irregular("duplicated component names")

    an = df[3].to_numpy(dtype=object)
    bn = df[4].to_numpy(dtype=object)
    leads = np.empty(2 * len(df), dtype=object)
    leads[0::2], leads[1::2] = an, bn
    codes, labels = pd.factorize(leads, sort=False)  # first-appearance order
    deg = np.bincount(codes, minlength=len(labels))
    labels = labels.tolist()

    nl = netlist
    nl._fast = True
    nl._df = df
    nl._name, nl._type, nl._value = name, ctype, value
    nl._nfields = nfields
    nl._acode, nl._bcode = codes[0::2], codes[1::2]
    nl.component_keys = name.tolist()
    nl._row_of = dict(zip(nl.component_keys, range(len(name))))
    nl.components = LazyComponents(nl)
    nl.degrees = dict(zip(labels, deg.tolist()))
    is_anom = np.isin(ctype.astype(str), c.NODE_TYPES_ANOM)
    nl._is_anom = is_anom
    anom_rows = np.flatnonzero(is_anom)
    nl.anomnum = dict(zip(name[anom_rows].tolist(), range(len(anom_rows))))
    nl.nums["components"] = int(len(name))
    nl.nums["anomalies"] = int(len(anom_rows))
    # ground: "g" if present, else the first node of maximal degree
    if "g" in nl.degrees:
        nl.ground = "g"
        gcode = labels.index("g")
    else:
        gcode = int(np.argmax(deg))
        nl.ground = labels[gcode]
    order = [lab for i, lab in enumerate(labels) if i != gcode]
    nl.nodenum = dict(zip(order, range(len(order))))
    node_index = np.arange(len(labels), dtype=np.int64)
    node_index[gcode + 1:] -= 1
    node_index[gcode] = -1
    nl._node_index = node_index  # label code -> nodenum index (-1 = ground)
    nl.nums["kcl"] = len(order)
    nl.nums["be"] = nl.nums["anomalies"]
    return nl


def row_fields(netlist, row):
    """The CSV fields of one row as the list the row-by-row parser would have seen."""
    if netlist._df is None:  # native tokenizer: split the stored line again
        nfile = len(netlist._line_off)
        if row >= nfile:
            return list(netlist._extra_rows[row - nfile])
        o = int(netlist._line_off[row])
        line = netlist._raw[o:o + int(netlist._line_len[row])].decode()
        return [f.lstrip(" ") for f in line.split(",")]
    if row >= len(netlist._df):
        return list(netlist._extra_rows[row - len(netlist._df)])
    return netlist._df.iloc[row].tolist()[: int(netlist._nfields[row])]


def append_row(netlist, data):
    """process_component() on a fast netlist (equivalent_resistance adds its probe
    source this way).  Same bookkeeping as the reference's method, on the columns."""
    from .netlist import Component
    nl = netlist
    comp = Component(data)  # validates, raises the reference's ValueErrors
    key = data[c.NCOL]
    if key in nl._row_of or data[c.TCOL] in ("OPMODEL", "OPAMP"):
        raise Irregular("needs the row-by-row bookkeeping")
    if not hasattr(nl, "_extra_rows"):
        nl._extra_rows = []
        nl._label_code = {lab: i for i, lab in enumerate(nl.degrees)}
    row = len(nl._name)
    nl._extra_rows.append(list(data))
    nl.component_keys.append(key)
    nl._row_of[key] = row
    nl.components._cache[key] = comp
    nl.nums["components"] += 1
    codes = []
    for lab in (data[c.ACOL], data[c.BCOL]):
        if lab not in nl.degrees:
            nl.degrees[lab] = 0
            nl._label_code[lab] = len(nl._label_code)
            nl._node_index = np.append(nl._node_index, -2)  # not in nodenum: KeyError later
        codes.append(nl._label_code[lab])
    for lab in (data[c.ACOL], data[c.BCOL]):
        nl.degrees[lab] += 1
    anom = data[c.TCOL] in c.NODE_TYPES_ANOM
    if anom:
        nl.anomnum[key] = nl.nums["anomalies"]
        nl.nums["anomalies"] += 1
    nl._name = np.append(nl._name, key)
    nl._type = np.append(nl._type, data[c.TCOL])
    nl._value = np.append(nl._value, comp.value)
    nl._acode = np.append(nl._acode, codes[0])
    nl._bcode = np.append(nl._bcode, codes[1])
    nl._is_anom = np.append(nl._is_anom, anom)


def lower_fast(netlist):
    """Vectorised equivalent of lowering.lower for a netlist read by `read_fast`."""
    from .lowering import ComponentTable, lower
    nl = netlist
    tidx = getattr(nl, "_tidx", None)
    if tidx is not None and len(tidx) == len(nl._type):
        # the native reader's type indices (no component was added since): a table lookup instead of
        # two sorts of 2e6 strings (0.3 s of the 0.5 s this function took at 1e6 nodes)
        tcode = np.array([c.TYPE_CODE[k] for k in _TYPE_NAMES.tolist()], dtype=np.uint8)[tidx]
    else:
        kinds = nl._type.astype(str)
        codes = np.array([c.TYPE_CODE[k] for k in np.unique(kinds)])
        _, inv = np.unique(kinds, return_inverse=True)
        tcode = codes[inv].astype(np.uint8)
    table = ComponentTable(len(tcode), nl.nums["kcl"], nl.nums["be"])
    if int(nl._is_anom.sum()) != nl.nums["be"]:
        raise Irregular("branch added after numbering")
    table.type[:] = tcode
    table.value[:] = nl._value
    table.a[:] = nl._node_index[nl._acode]
    table.b[:] = nl._node_index[nl._bcode]
    unknown = np.flatnonzero((table.a == -2) | (table.b == -2))
    if len(unknown):  # a lead that is not in nodenum (added after numbering): KeyError
        row = int(unknown[0])
        fields = row_fields(nl, row)
        lab = fields[c.ACOL] if table.a[row] == -2 else fields[c.BCOL]
        table.first_error = (row, KeyError(lab), False)
        return table
    # nums["be"] may lag behind anomalies added after numbering, exactly as in the reference
    table.k[nl._is_anom] = np.arange(int(nl._is_anom.sum()))
    # first string-level stamping error in file order, as lower() records it
    bad_r = np.flatnonzero((tcode == c.T_R) & (nl._value == 0))
    dep = np.flatnonzero(tcode >= c.T_VCVS)
    if len(dep):
        # dependent sources are few: reuse the exact per-row logic on them
        from .lowering import _lower_current_controlled, _node_index
        first_row = nl._row_of
        for row in dep.tolist():
            if len(bad_r) and bad_r[0] < row:
                break
            comp = nl.components[nl._name[row]]
            try:
                comp.cnode, comp.dnode = comp.pos_control, comp.neg_control
                if comp.type in c.NODE_TYPES_CC:
                    _lower_current_controlled(table, row, comp, nl.components, first_row,
                                              nl.ground, nl.nodenum)
                else:
                    table.c[row] = _node_index(comp.pos_control, nl.ground, nl.nodenum)
                    table.d[row] = _node_index(comp.neg_control, nl.ground, nl.nodenum)
            except (KeyError, ValueError, AssertionError, AttributeError, NotImplementedError,
                    ZeroDivisionError) as exc:
                probe = comp.type in ("VCVS", "VCCS", "CCCS")
                table.c[row] = table.d[row] = table.drv[row] = -1
                table.first_error = (row, exc, probe)
                return table
    if len(bad_r):
        table.first_error = (int(bad_r[0]),
                             ValueError("Model error: resistors can't have null resistance"), False)
    return table
