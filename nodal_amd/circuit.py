"""Circuit and Solution: the reference's Python seam over the HIP hot path.

Mirrors reference nodal/nodal.py:299-434 (`Circuit(netlist, sparse=False)`,
`.solve()`, `Solution`) with the same names, argument meaning and error
behaviour.  What the reference does with numpy/scipy on the host happens here
on an MI355X through libnodal_hip.so:

    Circuit.__init__ -> build_model -> lowering.lower (host, strings -> table)
                                    -> nodal_upload_components
                                    -> nodal_assemble_symbolic / _numeric  (HIP)
    Circuit.solve    -> nodal_solve_dense | nodal_solve_sparse            (HIP)

`Circuit.G` / `Circuit.A` are exported from the device on first access (numpy
array for the dense path, scipy CSR for the sparse path) so that code written
against the reference's attributes keeps working without paying a device->host
copy on the hot path.
"""

import logging
import os
import warnings

import numpy as np

from . import _ffi
from . import constants as c
from .lowering import lower
from .netlist import Netlist, UnconnectedCircuitError, is_connected

try:  # same warning class the reference's spsolve call emits
    from scipy.sparse.linalg import MatrixRankWarning
except Exception:  # pragma: no cover - scipy is optional at run time
    class MatrixRankWarning(UserWarning):
        pass


def default_device():
    return int(os.environ.get("NODAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))


# Device contexts are expensive to create (four HIP streams, one of them CU-masked, a dozen
# events: ~30 ms) and cheap to reuse (their buffers grow on demand), so a Circuit borrows one
# from this pool and hands it back when it is garbage collected.
# (The pool is touched from Circuit.__del__, i.e. from whatever thread the collector runs on, while the lanes of a
# long resistance sweep build Circuits on threads of their own (equiv.py): one lock around every look at it.  A
# re-entrant one: a collection triggered inside the locked region may run another Circuit's __del__ on this thread.)
import threading

_IDLE_HANDLES = {}
_MAX_IDLE = 4
_POOL_LOCK = threading.RLock()


def _acquire_handle(device):
    with _POOL_LOCK:
        idle = _IDLE_HANDLES.get(device)
        while idle:
            h = idle.pop()
            if not h.closed:
                return h
    return _ffi.Handle(device)


def _release_handle(device, h):
    if h is None or h.closed:
        return
    with _POOL_LOCK:
        idle = _IDLE_HANDLES.setdefault(device, [])
        if len(idle) < _MAX_IDLE:
            idle.append(h)
            return
    h.close()


class Circuit:
    """Builds the linear system G e = A of a Netlist on the GPU.

    Attributes: netlist, sparse, G, A, currents (reference nodal/nodal.py:306-311).
    """

    def __init__(self, netlist, sparse=False, device=None):
        if not isinstance(netlist, Netlist):
            raise TypeError("Input isn't a netlist")
        self.netlist = netlist
        self.sparse = sparse
        self._device = default_device() if device is None else device
        self._handle = None
        self._G = self._A = None
        self.currents = self.build_model()

    @classmethod
    def _clone_of(cls, other):
        """A second device context holding the same assembled system as `other` (no second lowering of the
        netlist): the lanes of a long equivalent-resistance sweep (equiv.py)."""
        self = cls.__new__(cls)
        self.netlist, self.sparse, self._device = other.netlist, other.sparse, other._device
        self._handle = None
        self._G = self._A = None
        self.table = other.table
        self.currents = other.currents
        self._assemble(self.table)
        return self

    def __del__(self):
        try:
            handle, self._handle = self._handle, None
            _release_handle(self._device, handle)
        except Exception:  # interpreter shutdown
            pass

    # -- assembly ----------------------------------------------------------
    def build_model(self):
        """Lower the netlist, upload the component table and assemble G, A on
        the device (reference nodal/nodal.py:338-398).  Returns `currents`."""
        nl = self.netlist
        table = self._lower(nl)
        self.table = table
        if table.first_error is not None:
            row, exc, probe = table.first_error
            # the reference would have hit an earlier stamp collision first
            prefix = table.truncated(row + (1 if probe else 0))
            if prefix.ncomp:
                self._assemble(prefix)
            raise exc
        self._assemble(table)
        if getattr(nl, "_fast", False):
            if not nl._is_anom.any():  # (no branch unknowns: the names are not even looked at)
                return []
            return nl._name[nl._is_anom].tolist()
        comps = nl.components
        return [key for key in nl.component_keys if comps[key].type in c.NODE_TYPES_ANOM]

    @staticmethod
    def _lower(nl):
        if getattr(nl, "_fast", False):
            from . import fastparse
            try:
                return fastparse.lower_fast(nl)
            except fastparse.Irregular:
                nl._demote()
        return lower(nl)

    def _assemble(self, table):
        if self._handle is None:
            self._handle = _acquire_handle(self._device)
        h = self._handle
        h.upload(table)
        h.assemble_symbolic()
        status, bad = h.assemble_numeric(0)
        if status == _ffi.E_ZERO_RESISTANCE:
            raise ValueError("Model error: resistors can't have null resistance")
        if status == _ffi.E_STAMP_COLLISION:
            raise AssertionError  # the reference's bare `assert G[i, j] == 0`

    # -- reference attributes, exported lazily -------------------------------
    @property
    def G(self):
        if self._G is None:
            h = self._handle
            if self.sparse:
                indptr, indices, data, rhs = h.export_csr()
                import scipy.sparse as spsp
                self._G = spsp.csr_matrix((data, indices, indptr), shape=(h.n, h.n))
                # The reference stamps into a dok_matrix, which never stores an exact zero (a +g / -g
                # pair, a zero gain): its `G.tocsr()` has no such entries (reference nodal/nodal.py:
                # 396-397).  The device keeps them -- the pattern is the topology's, whatever the
                # values -- so they are dropped from the exported copy only.
                self._G.eliminate_zeros()
                self._A = rhs
            else:
                self._G, self._A = h.export_dense()
        return self._G

    @property
    def A(self):
        if self._A is None:
            self._A = self._handle.export_csr()[3]
        return self._A

    # -- solve ---------------------------------------------------------------
    def solve(self):
        """Solve G e = A on the device (reference nodal/nodal.py:313-336).

        Raises numpy.linalg.LinAlgError when the dense system is singular and
        the circuit is connected, UnconnectedCircuitError when it is not.  The
        sparse path does not raise on a matrix that is singular by construction
        (floating sub-network, loop of voltage sources, exact zero pivot of a
        small system): like the reference's spsolve call it warns
        (MatrixRankWarning) and returns NaNs.  A large general system whose
        iteration does not converge and that is not singular by construction
        raises NodalHipError (never a wrong answer)."""
        h = self._handle
        if self.sparse:
            e, info, self.iterations, self.relative_residual = h.solve_sparse()
            if info > 0:
                warnings.warn("Matrix is exactly singular", MatrixRankWarning, stacklevel=2)
        else:
            e, info = h.solve_dense()
            if info > 0:
                if not is_connected(self.netlist):
                    logging.error("Model error: unconnected circuit")
                    raise UnconnectedCircuitError
                logging.error("Model error: matrix is singular")
                raise np.linalg.LinAlgError("Singular matrix")
        return Solution(e, self.netlist, self.currents)

    def scaled_residual(self):
        """||G x - A||_inf / (||G||_inf ||x||_inf + ||A||_inf) of the last
        solution, computed on the device."""
        return self._handle.residual()


class Solution:
    """Result of Circuit.solve(): `result[0:K]` node potentials in nodenum
    order, `result[K:K+B]` branch currents in anomnum order; printable
    (reference nodal/nodal.py:401-434)."""

    def __init__(self, result, netlist, currents):
        self.result = result
        self._netlist = netlist
        self.nums = netlist.nums
        self.currents = currents
        self.ground = netlist.ground

    # (the netlist's own dicts, as in the reference -- nodal/nodal.py:416-420 aliases them -- but looked at only when
    # somebody does: a natively read netlist builds them on first access)
    @property
    def nodenum(self):
        return self._netlist.nodenum

    @property
    def anomnum(self):
        return self._netlist.anomnum

    def __str__(self):
        # values as str(np.float64) prints them (shortest round-trip repr), names in
        # lexicographic string order: reference nodal/nodal.py:422-434.  tolist() +
        # repr() gives the same digits as formatting np.float64 one by one, without a
        # numpy scalar object per line.
        result = np.asarray(self.result, dtype=np.float64)
        lines = [f"Ground node: {self.ground}"]
        # a natively read netlist: the million "e(...)" lines by the host library, from the label blob (round 5; the
        # same text, tests/test_frontend.py)
        from . import fastparse
        block = fastparse.native_potential_lines(self._netlist, result) if len(result) >= 20000 else None
        values = None
        if block is not None:
            if block:
                lines.append(block)
        else:
            values = result.tolist()
            nodenum = self.nodenum
            lines += [f"e({name}) \t= {values[nodenum[name]]!r}" for name in sorted(nodenum)]
        offset, anomnum = self.nums["kcl"], self.anomnum
        if anomnum:
            if values is None:
                values = result.tolist()
            lines += [f"i({name}) \t= {values[offset + anomnum[name]]!r}" for name in sorted(anomnum)]
        return "\n".join(lines)
