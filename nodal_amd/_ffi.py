"""ctypes binding of libnodal_hip.so (include/nodal_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible
when a handle is created, this raises.  The library itself is loadable on a
machine without a GPU (symbol checks only).
"""

import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnodal_hip.so")

OK, E_INVALID, E_HIP, E_ZERO_RESISTANCE, E_STAMP_COLLISION, E_SINGULAR, E_NOMEM, \
    E_UNSUPPORTED = range(8)
SPARSE_AUTO, SPARSE_PCG, SPARSE_DENSIFY, SPARSE_LU, SPARSE_DIRECT = range(5)
OPT_FORCE_PIVOTING = 1
OPT_GEPP_PANEL = 2
OPT_EXTRA_STREAMS = 3
OPT_BORROW_TABLE = 4

_p = C.POINTER
_i32p, _i64p, _f64p, _u8p = _p(C.c_int32), _p(C.c_int64), _p(C.c_double), _p(C.c_uint8)

# name -> (restype, argtypes); every symbol include/nodal_hip.h declares
SIGNATURES = {
    "nodal_version": (C.c_char_p, []),
    "nodal_create": (C.c_int, [C.c_int, _p(C.c_void_p)]),
    "nodal_destroy": (C.c_int, [C.c_void_p]),
    "nodal_last_error": (C.c_char_p, [C.c_void_p]),
    "nodal_upload_components": (C.c_int, [C.c_void_p, C.c_int64, _u8p, _f64p, _i32p, _i32p,
                                          _i32p, _i32p, _i32p, _i32p, C.c_int32, C.c_int32]),
    "nodal_host_alloc": (C.c_int, [C.c_size_t, _p(C.c_void_p)]),
    "nodal_host_free": (C.c_int, [C.c_void_p]),
    "nodal_upload_values": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "nodal_assemble_symbolic": (C.c_int, [C.c_void_p]),
    "nodal_assemble_numeric": (C.c_int, [C.c_void_p, C.c_int32, _i64p]),
    "nodal_get_sizes": (C.c_int, [C.c_void_p, _i64p, _i64p, _i64p]),
    "nodal_export_csr": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p, _f64p]),
    "nodal_export_dense": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "nodal_solve_dense": (C.c_int, [C.c_void_p, _f64p, _i32p]),
    "nodal_solve_sparse": (C.c_int, [C.c_void_p, C.c_int32, _f64p, _i32p, _i32p, _f64p]),
    "nodal_download_x": (C.c_int, [C.c_void_p, _f64p]),
    "nodal_solve_pairs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p, _f64p, _i32p]),
    "nodal_residual": (C.c_int, [C.c_void_p, _f64p]),
    "nodal_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _i32p]),
    "nodal_run_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p, _i32p]),
    "nodal_batch_x_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "nodal_x_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "nodal_last_timings": (C.c_int, [C.c_void_p, _f64p]),
    "nodal_last_kernel_stats": (C.c_int, [C.c_void_p, _f64p, _i64p, _f64p]),
    "nodal_last_solve_info": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p]),
    "nodal_synchronize": (C.c_int, [C.c_void_p]),
    "nodal_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "nodal_debug_gemm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p, _f64p]),
}

_lib = None
_live = weakref.WeakSet()  # handles still open; closed before the HIP runtime unloads


@atexit.register
def _close_all():
    _closing[0] = True
    for h in list(_live):
        h.close()


class NodalHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libnodal_hip status {status}: {message}")
        self.status = status


def load():
    """Load libnodal_hip.so and set the prototypes.  Raises OSError with build
    instructions when the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} not found: build it with `make` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`) -- "
                "nodal_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptr(arr, ctype):
    return arr.ctypes.data_as(_p(ctype))


# ---- pinned host arrays for large component tables ---------------------------------
# A table column allocated here is page-locked: nodal_upload_components then copies it by
# DMA at link rate instead of through the runtime's staging buffers.  Blocks are recycled
# (hipHostMalloc of tens of MB costs milliseconds); without a HIP device the arrays are
# ordinary numpy memory.
PINNED_MIN_BYTES = 1 << 20
_PINNED_POOL = {}          # bytes -> [address, ...] of free blocks
_PINNED_POOL_BYTES = [0]
_PINNED_POOL_MAX = 1 << 30
_pinned_ok = [None]
_closing = [False]


def _pinned_release(addr, nbytes):
    if _lib is None or _closing[0]:
        return  # (interpreter shutdown: the runtime frees what is left)
    if _PINNED_POOL_BYTES[0] + nbytes <= _PINNED_POOL_MAX:
        _PINNED_POOL.setdefault(nbytes, []).append(addr)
        _PINNED_POOL_BYTES[0] += nbytes
    else:
        _lib.nodal_host_free(C.c_void_p(addr))


def host_empty(count, dtype):
    """1-D numpy array of `count` items, in pinned host memory when it is large and a HIP
    device is there, ordinary memory otherwise."""
    dtype = np.dtype(dtype)
    nbytes = int(count) * dtype.itemsize
    if nbytes < PINNED_MIN_BYTES or _pinned_ok[0] is False:
        return np.empty(count, dtype=dtype)
    try:
        lib = load()
    except OSError:
        _pinned_ok[0] = False
        return np.empty(count, dtype=dtype)
    size = (nbytes + 4095) & ~4095
    free = _PINNED_POOL.get(size)
    if free:
        addr = free.pop()
        _PINNED_POOL_BYTES[0] -= size
    else:
        out = C.c_void_p()
        if lib.nodal_host_alloc(size, C.byref(out)) != OK or not out.value:
            _pinned_ok[0] = False
            return np.empty(count, dtype=dtype)
        _pinned_ok[0] = True
        addr = out.value
    buf = (C.c_char * size).from_address(addr)
    # every array (and view) made from `buf` keeps it alive; the block goes back to the pool with it
    weakref.finalize(buf, _pinned_release, addr, size)
    return np.frombuffer(buf, dtype=dtype, count=count)


class Handle:
    """Owns one nodal_handle (device memory + stream) on `device`."""

    def __init__(self, device=0):
        self.lib = load()
        self._h = C.c_void_p()
        status = self.lib.nodal_create(device, C.byref(self._h))
        if status != OK:
            raise NodalHipError(
                status, f"nodal_create(device={device}) failed: no usable MI355X "
                "(HIP device) is visible; nodal_amd has no CPU fallback")
        self.n = self.nnz = self.ncontrib = 0
        # upload() keeps the table's columns alive (self._keep) until the next upload: the library may read them in place
        self.lib.nodal_set_option(self._h, OPT_BORROW_TABLE, 1)
        _live.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.nodal_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def closed(self):
        return not getattr(self, "_h", None)

    def _check(self, status, allow=()):
        if status != OK and status not in allow:
            raise NodalHipError(status, self.lib.nodal_last_error(self._h).decode())
        return status

    # -- table ------------------------------------------------------------
    def upload(self, table):
        # B == 0: resistors and current sources only -- the four columns they never read stay home
        names = ("type", "value", "a", "b") if table.B == 0 else ("type", "value", "a", "b", "c", "d", "drv", "k")
        cols = [np.ascontiguousarray(getattr(table, n)) for n in names]
        self._keep = cols
        t, v, a, b = cols[:4]
        rest = [_ptr(x, C.c_int32) for x in cols[4:]] or [None] * 4
        self.n_members = table.K + table.B
        self._check(self.lib.nodal_upload_components(
            self._h, table.ncomp, _ptr(t, C.c_uint8), _ptr(v, C.c_double),
            _ptr(a, C.c_int32), _ptr(b, C.c_int32), *rest, table.K, table.B))

    def upload_values(self, values):
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert values.ndim == 2
        self._check(self.lib.nodal_upload_values(self._h, values.shape[0], _ptr(values, C.c_double)))

    # -- assembly ---------------------------------------------------------
    def assemble_symbolic(self):
        self._check(self.lib.nodal_assemble_symbolic(self._h))
        self._refresh_sizes()

    def _refresh_sizes(self):
        n, nnz, nc = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.nodal_get_sizes(self._h, C.byref(n), C.byref(nnz), C.byref(nc)))
        self.n, self.nnz, self.ncontrib = n.value, nnz.value, nc.value

    def assemble_numeric(self, member=0):
        """Returns (status, bad_component): status is OK, E_ZERO_RESISTANCE or
        E_STAMP_COLLISION."""
        bad = C.c_int64(-1)
        status = self._check(self.lib.nodal_assemble_numeric(self._h, member, C.byref(bad)),
                             allow=(E_ZERO_RESISTANCE, E_STAMP_COLLISION))
        return status, bad.value

    def export_csr(self, values=True):
        indptr = np.empty(self.n + 1, dtype=np.int32)
        indices = np.empty(self.nnz, dtype=np.int32)
        data = np.empty(self.nnz, dtype=np.float64) if values else None
        rhs = np.empty(self.n, dtype=np.float64) if values else None
        self._check(self.lib.nodal_export_csr(
            self._h, _ptr(indptr, C.c_int32), _ptr(indices, C.c_int32),
            _ptr(data, C.c_double) if values else None,
            _ptr(rhs, C.c_double) if values else None))
        return indptr, indices, data, rhs

    def export_dense(self):
        G = np.empty((self.n, self.n), dtype=np.float64)
        rhs = np.empty(self.n, dtype=np.float64)
        self._check(self.lib.nodal_export_dense(self._h, _ptr(G, C.c_double), _ptr(rhs, C.c_double)))
        return G, rhs

    # -- solve ------------------------------------------------------------
    def solve_dense(self, download=True):
        """Returns (x or None, info).  info > 0: exact zero pivot (singular)."""
        x = host_empty(self.n, np.float64) if download else None  # (pinned when large: the copy down is plain DMA)
        info = C.c_int32(0)
        self._check(self.lib.nodal_solve_dense(
            self._h, _ptr(x, C.c_double) if download else None, C.byref(info)),
            allow=(E_SINGULAR,))
        return x, info.value

    def solve_sparse(self, method=SPARSE_AUTO, download=True):
        """Returns (x or None, info, iterations, relative residual)."""
        x = host_empty(self.n, np.float64) if download else None  # (pinned when large: the copy down is plain DMA)
        info, iters, resid = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self._check(self.lib.nodal_solve_sparse(
            self._h, method, _ptr(x, C.c_double) if download else None,
            C.byref(info), C.byref(iters), C.byref(resid)))
        return x, info.value, iters.value, resid.value

    def solve_pairs(self, ia, ib, dense):
        """Equivalent resistance for every node-index pair; returns (R array, info)."""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ib = np.ascontiguousarray(ib, dtype=np.int32)
        out = np.empty(len(ia), dtype=np.float64)
        info = C.c_int32(0)
        self._check(self.lib.nodal_solve_pairs(self._h, int(dense), len(ia), _ptr(ia, C.c_int32),
                                               _ptr(ib, C.c_int32), _ptr(out, C.c_double),
                                               C.byref(info)), allow=(E_SINGULAR,))
        return out, info.value

    def download_x(self):
        x = host_empty(self.n, np.float64)
        self._check(self.lib.nodal_download_x(self._h, _ptr(x, C.c_double)))
        return x

    def residual(self):
        r = C.c_double(0)
        self._check(self.lib.nodal_residual(self._h, C.byref(r)))
        return r.value

    def run(self, dense, member=0, reuse_symbolic=False):
        info = C.c_int32(0)
        self._check(self.lib.nodal_run(self._h, int(dense), member, int(reuse_symbolic),
                                       C.byref(info)), allow=(E_SINGULAR,))
        self._refresh_sizes()
        return info.value

    def run_batch(self, first, count, reuse_symbolic=False, download=True):
        """Members [first, first + count) of the uploaded value table as one block-diagonal
        system.  Returns ([count, n] array or None, info[count])."""
        x = np.empty((count, self.n_members), dtype=np.float64) if download else None
        info = np.zeros(count, dtype=np.int32)
        self._check(self.lib.nodal_run_batch(self._h, first, count, int(reuse_symbolic),
                                             _ptr(x, C.c_double) if download else None,
                                             _ptr(info, C.c_int32)))
        return x, info

    def batch_x_to_device(self, data_ptr, capacity_bytes):
        """Copy the last run_batch's results into device memory (a torch tensor's data_ptr())."""
        self._check(self.lib.nodal_batch_x_device(self._h, C.c_void_p(data_ptr), capacity_bytes))

    def x_to_device(self, data_ptr, capacity_bytes):
        """Copy the last single-circuit solution into device memory (a torch tensor's data_ptr())."""
        self._check(self.lib.nodal_x_device(self._h, C.c_void_p(data_ptr), capacity_bytes))

    def timings(self):
        ms = (C.c_double * 3)()
        self._check(self.lib.nodal_last_timings(self._h, ms))
        return list(ms)

    def kernel_stats(self):
        ms, launches, alg = C.c_double(0), C.c_int64(0), C.c_double(0)
        self._check(self.lib.nodal_last_kernel_stats(self._h, C.byref(ms), C.byref(launches),
                                                     C.byref(alg)))
        return ms.value, launches.value, alg.value

    def set_option(self, option, value):
        self._check(self.lib.nodal_set_option(self._h, option, int(value)))

    def debug_gemm(self, A, B, Cm):
        """C - A @ B through the LU's trailing-update kernel (testing hook)."""
        A, B, Cm = (np.asfortranarray(x, dtype=np.float64) for x in (A, B, Cm))
        out = Cm.copy(order="F")
        self._check(self.lib.nodal_debug_gemm(self._h, A.shape[0], B.shape[1], A.shape[1],
                                              _ptr(A, C.c_double), _ptr(B, C.c_double),
                                              _ptr(out, C.c_double)))
        return out

    def solve_info(self):
        """(iterations, multigrid levels, relative residual) of the last sparse solve."""
        it, lv, rr = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self._check(self.lib.nodal_last_solve_info(self._h, C.byref(it), C.byref(lv), C.byref(rr)))
        return it.value, lv.value, rr.value

    def synchronize(self):
        self._check(self.lib.nodal_synchronize(self._h))
