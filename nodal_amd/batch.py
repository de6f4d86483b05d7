"""Independent circuits across the GPUs of one node (BASELINE.json config 4).

The hot path shards only along the batch dimension: circuits (or the members of
a value sweep on one topology) are independent, so ranks never exchange matrix
data.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI; "gloo" on CPU for tests).  Collectives are used only at the edges:

  * optional broadcast of the shared component table from rank 0 (0.5 MB for
    grid(100)) when only rank 0 parsed the netlist;
  * all_gather of the per-rank solution blocks (128 x 9999 x 8 B = 10 MB per
    GPU for config 4) straight from device memory: direct all-to-all traffic
    over the xGMI mesh, no reduction, no ring.

On a GPU the members of a shard are assembled and solved as ONE block-diagonal
system that `nodal_run_batch` builds on the device from the single topology and
the `values[batch][ncomp]` table in HBM (csrc/batch.hip); what the reference
does for the same job is a Python loop of `Circuit(netlist, sparse)` +
`.solve()` (reference nodal/nodal.py:306-336).
"""

import warnings

import numpy as np


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` members for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def replicate_table(table, values):
    """Host-side statement of the block-diagonal system `nodal_run_batch` builds on the
    device: M members of a K-node, B-branch topology as one netlist with M*K nodes and
    M*B branch unknowns (independent circuits that share the ground node do not couple:
    the ground row is eliminated).  Member m owns x[m*K:(m+1)*K] and
    x[M*K + m*B : M*K + (m+1)*B].  Kept for tests and tools; the product path does this
    on the GPU."""
    from .lowering import ComponentTable
    M = values.shape[0]
    nc, K, B = table.ncomp, table.K, table.B
    big = ComponentTable(M * nc, M * K, M * B)
    big.type[:] = np.tile(table.type, M)
    big.value[:] = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    member = np.repeat(np.arange(M, dtype=np.int64), nc)

    def shift(col, stride):
        col = np.tile(col.astype(np.int64), M)
        return np.where(col >= 0, col + member * stride, -1).astype(np.int32)

    big.a[:], big.b[:] = shift(table.a, K), shift(table.b, K)
    big.c[:], big.d[:] = shift(table.c, K), shift(table.d, K)
    big.drv[:] = shift(table.drv, nc)
    big.k[:] = shift(table.k, B)
    return big


def split_solution(x, M, K, B):
    """[members, K+B] view of the block-diagonal solution vector."""
    out = np.empty((M, K + B))
    out[:, :K] = x[: M * K].reshape(M, K)
    out[:, K:] = x[M * K:].reshape(M, B)
    return out


def _raise_member_errors(info, first=0):
    """info < 0 is minus the status a member's assembly failed with: raise what
    `Circuit(netlist)` raises for that member (reference nodal/models.py:14-17 and its
    bare `assert G[i, j] == 0`).  info > 0 (singular member) is not an error on the sparse
    path: the row holds NaNs and the caller is warned, as the reference's spsolve does."""
    from . import _ffi
    from .circuit import MatrixRankWarning
    for m, inf in enumerate(np.asarray(info).tolist()):
        if inf == -_ffi.E_ZERO_RESISTANCE:
            raise ValueError(f"Model error: resistors can't have null resistance (member {first + m})")
        if inf == -_ffi.E_STAMP_COLLISION:
            raise AssertionError(f"stamp collision in member {first + m}")
    if np.any(np.asarray(info) > 0):
        warnings.warn("Matrix is exactly singular", MatrixRankWarning, stacklevel=3)


class BatchSolver:
    """One device context kept alive across calls (creating one costs ~30 ms: four HIP
    streams, a dozen events) with the topology resident in HBM.  `solve` uploads a value
    table and returns the [members, n] solutions; `solve_into` leaves them in a device
    buffer (the send buffer of the gather)."""

    def __init__(self, table, device=0):
        from . import _ffi
        self.table = table
        self.device = device
        self.h = _ffi.Handle(device)
        self.h.upload(table)
        self._fresh = True  # no block pattern yet for this topology
        self.last_info = None

    def close(self):
        if self.h is not None:
            self.h.close()
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def upload_values(self, values):
        self.h.upload_values(values)
        self.members = values.shape[0]

    def run(self, sparse=True, reuse_symbolic=False, download=True):
        """Solve the uploaded members.  Returns [members, n] (or None with download=False:
        the results stay on the device for `copy_to_device`)."""
        h = self.h
        M = self.members
        if sparse:
            x, info = h.run_batch(0, M, reuse_symbolic=reuse_symbolic and not self._fresh,
                                  download=download)
            self._fresh = False
            self.last_info = info
            _raise_member_errors(info)
            return x
        # dense members: one LU each (the dense path has no block form)
        out = np.empty((M, self.table.n))
        h.assemble_symbolic()
        for i in range(M):
            status, _bad = h.assemble_numeric(i)
            _raise_member_errors([-status], first=i)
            x, info = h.solve_dense()
            if info > 0:
                raise np.linalg.LinAlgError("Singular matrix")
            out[i] = x
        return out

    def solve(self, values, sparse=True):
        self.upload_values(np.ascontiguousarray(values, dtype=np.float64))
        return self.run(sparse)

    def copy_to_device(self, tensor):
        """Results of the last sparse `run` into a torch CUDA tensor of >= members*n doubles."""
        self.h.batch_x_to_device(tensor.data_ptr(), tensor.numel() * tensor.element_size())


def solve_members(table, values, sparse=True, device=0, solver=None, session=None):
    """Solve every row of `values` ([members, ncomp]) on one GPU with a shared
    symbolic phase.  Returns [members, n] float64.  `solver` lets tests inject
    a stand-in for the HIP handle; `session` is a BatchSolver to reuse."""
    if solver is not None:
        return solver(table, values, sparse)
    if values.shape[0] == 0:
        return np.empty((0, table.n))
    if session is not None:
        return session.solve(values, sparse)
    with BatchSolver(table, device) as s:
        return s.solve(values, sparse)


def local_device_index(rank):
    """The HIP device of this rank: LOCAL_RANK (torchrun) or the rank, modulo the devices VISIBLE to the
    process -- a launcher that gives every rank one visible device (ROCR_VISIBLE_DEVICES per rank) makes that
    device 0 for all of them."""
    import os
    import torch
    count = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", rank))
    return local % count if count > 0 else 0


class ShardedBatch:
    """This rank's shard of a value sweep, kept across steps: the one entry the benchmark
    (`bench.py` config 4), `solve_batch_distributed` and the tests share.

    `total` members of one topology are split into contiguous shards (`shard_range`); a step
    solves this rank's members as one block-diagonal system on its GPU (`nodal_run_batch`),
    copies the [members][n] results device-to-device into `block` (`nodal_batch_x_device`)
    and gathers every rank's block with `all_gather_into_tensor` -- RCCL over xGMI with
    backend "nccl", straight from device memory; gloo (CPU tests, rehearsals) goes through
    host memory.  `solver` replaces the HIP path by a stand-in (the CPU tests pass a reference
    solver: there is no GPU in the build container).  `force_collective` runs the gather even with a
    single rank (the RCCL path on a one-GPU box)."""

    def __init__(self, table, total, dist=None, device=None, solver=None, force_collective=False, session=None,
                 sparse=True):
        import torch
        self.table, self.total, self.dist, self.solver = table, total, dist, solver
        self.sparse = sparse
        grouped = dist is not None and dist.is_initialized()
        self.rank = dist.get_rank() if grouped else 0
        self.world = dist.get_world_size() if grouped else 1
        self.backend = dist.get_backend() if grouped else None
        self.lo, self.hi = shard_range(total, self.rank, self.world)
        self.width = -(-total // self.world) if total else 0  # all_gather needs equal blocks
        on_gpu = solver is None
        needs_gpu = on_gpu or self.backend == "nccl"
        if session is not None:
            dev_index = session.device
        else:
            dev_index = device if device is not None else (local_device_index(self.rank) if needs_gpu else 0)
        self.dev = torch.device("cuda", dev_index) if needs_gpu else torch.device("cpu")
        self._own_session = on_gpu and session is None
        self.session = (session if session is not None else BatchSolver(table, dev_index)) if on_gpu else None
        # short shards are padded with NaN rows (written once: a step only rewrites the members); one more row
        # travels with the block: its first word says whether this rank's shard raised (step)
        self.rows = self.width + 1
        self.block = torch.full((self.rows, max(table.n, 1)), float("nan"), dtype=torch.float64, device=self.dev)
        self.block[self.width, 0] = 0.0
        collective = grouped and (self.world > 1 or force_collective)
        self.gathered = (torch.empty((self.world * self.rows, max(table.n, 1)), dtype=torch.float64, device=self.dev)
                         if collective else None)
        self._sync_torch()
        self.values = None
        self.gather_ms = 0.0

    def _sync_torch(self):
        # `block` is written on the library's own stream and read by the collective on torch's:
        # neither may start before the other side's previous use of the buffer has finished
        if self.dev.type == "cuda":
            import torch
            torch.cuda.current_stream(self.dev).synchronize()

    def close(self):
        if self.session is not None and self._own_session:
            self.session.close()
        self.session = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def upload(self, values):
        """`values`: this rank's [hi - lo, ncomp] slice of the sweep."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert values.shape[0] == self.hi - self.lo
        self.values = values
        if self.session is not None and values.shape[0]:
            self.session.upload_values(values)

    def _solve_shard(self, reuse_symbolic):
        import torch
        if self.session is not None and self.sparse:
            self.session.run(sparse=True, reuse_symbolic=reuse_symbolic, download=False)
            self._sync_torch()
            self.session.copy_to_device(self.block)  # returns once the copy has finished
        elif self.session is not None:  # dense members: one factorisation each, this rank's shard only
            mine = self.session.run(sparse=False)
            self._sync_torch()
            self.block[: self.hi - self.lo] = torch.from_numpy(np.ascontiguousarray(mine)).to(self.dev)
        else:
            mine = self.solver(self.table, self.values, self.sparse)
            self.block[: self.hi - self.lo] = torch.from_numpy(np.ascontiguousarray(mine)).to(self.dev)

    def step(self, reuse_symbolic=False):
        import time
        import torch
        # A member that raises on ITS rank (a singular dense member: LinAlgError; a zero resistance: ValueError; a
        # stamp collision: AssertionError -- what Circuit(netlist) raises for it) must not leave the other ranks
        # waiting inside the collective: the rank keeps the exception, fills its block with NaNs, goes through the
        # gather like everybody else, and every rank raises afterwards (the owner its own exception, the others
        # the same class with the owner's rank in the message).
        failure = None
        if self.hi > self.lo:
            if self.gathered is None:
                self._solve_shard(reuse_symbolic)
            else:
                try:
                    self._solve_shard(reuse_symbolic)
                except (np.linalg.LinAlgError, ValueError, AssertionError) as e:
                    failure = e
                    self._sync_torch()
                    self.block[: self.hi - self.lo] = float("nan")
        if self.gathered is not None:
            code = 0 if failure is None else next(i for i, c in enumerate(self._FAILURES)
                                                  if c is not None and isinstance(failure, c))
            if code or self._flagged:
                self.block[self.width, 0] = float(code)
                self._flagged = bool(code)
            t0 = time.perf_counter()
            if self.backend == "nccl" or self.dev.type == "cpu":
                self.dist.all_gather_into_tensor(self.gathered, self.block)
            else:  # rehearsal: GPU results gathered over gloo through host memory
                g = torch.empty(self.gathered.shape, dtype=torch.float64)
                self.dist.all_gather_into_tensor(g, self.block.cpu())
                self.gathered.copy_(g)
            self._sync_torch()
            self.gather_ms += (time.perf_counter() - t0) * 1e3
            self._raise_together(failure)

    _FAILURES = (None, np.linalg.LinAlgError, ValueError, AssertionError)
    _flagged = False

    def _raise_together(self, failure):
        """Every rank reads the status words that travelled with the blocks and all raise alike."""
        codes = self.gathered[self.width::self.rows, 0].cpu().tolist()
        if failure is not None:
            raise failure
        for r, c in enumerate(codes):
            if c:
                cls = self._FAILURES[int(c)]
                raise cls(f"a member of rank {r}'s shard failed there with {cls.__name__}")

    def own_block(self):
        """[hi - lo, n] results of this rank's members (host copy)."""
        return self.block[: self.hi - self.lo, : self.table.n].cpu().numpy()

    def result(self):
        """[total, n]: every member, on every rank (after a step with the gather)."""
        if self.gathered is None:
            return self.own_block()
        flat = self.gathered.cpu().numpy()
        out = np.empty((self.total, self.table.n))
        for r in range(self.world):
            rlo, rhi = shard_range(self.total, r, self.world)
            out[rlo:rhi] = flat[r * self.rows: r * self.rows + (rhi - rlo), : self.table.n]
        return out


class ShardedCircuits:
    """Independent circuits over the ranks of one node (north_star: "partition independent netlists ... across
    the 8 GPUs"): every rank solves ITS OWN circuits, one after the other on its GPU -- symbolic + numeric assembly
    + solve, the equivalent of the reference's `Circuit(netlist, sparse)` + `.solve()` per circuit (reference
    nodal/nodal.py:306-336) -- and the solution of each finished circuit is shared with every rank by
    `all_gather_into_tensor` (RCCL over xGMI with backend "nccl": n x 8 bytes per rank and circuit, straight from
    device memory, no reduction).  The gather of circuit i runs on torch's stream WHILE the library's stream solves
    circuit i + 1: x is copied device-to-device into one of two send buffers (`nodal_x_device`), the collective is
    started asynchronously, and a slot is only reused after its collective has finished.

    `bench.py --gpus N` times this for config 3 at every N (the N = 1 line has no process group: no copy, no
    gather -- the plain sequence of `nodal_run` calls).  `solver(table) -> x` replaces the HIP path in the CPU
    tests (gloo); with a GPU handle and a gloo group (the one-GPU rehearsal) x travels through host memory.
    All ranks' tables must have the same number of unknowns."""

    def __init__(self, table, dist=None, device=None, solver=None, force_collective=False, dense=False):
        import torch
        self.table, self.dist, self.solver, self.dense = table, dist, solver, dense
        grouped = dist is not None and dist.is_initialized()
        self.rank = dist.get_rank() if grouped else 0
        self.world = dist.get_world_size() if grouped else 1
        self.backend = dist.get_backend() if grouped else None
        on_gpu = solver is None
        needs_gpu = on_gpu or self.backend == "nccl"
        dev_index = device if device is not None else (local_device_index(self.rank) if needs_gpu else 0)
        self.h = None
        self.upload_first_ms = None
        if on_gpu:
            from . import _ffi
            self.h = _ffi.Handle(dev_index)
            self.h.set_option(_ffi.OPT_EXTRA_STREAMS, 1)  # one handle per rank, one call at a time
            import time
            t0 = time.perf_counter()
            self.h.upload(table)
            self.h.synchronize()
            self.upload_first_ms = (time.perf_counter() - t0) * 1e3
        self.collective = grouped and (self.world > 1 or force_collective)
        # the gather runs where the group lives: device memory for RCCL, host memory for gloo
        self.dev = torch.device("cuda", dev_index) if (self.backend == "nccl") else torch.device("cpu")
        n = table.n
        self.send = self.recv = None
        if self.collective:
            self.send = [torch.empty(n, dtype=torch.float64, device=self.dev) for _ in range(2)]
            self.recv = [torch.empty(self.world * n, dtype=torch.float64, device=self.dev) for _ in range(2)]
            if self.dev.type == "cuda":
                torch.cuda.current_stream(self.dev).synchronize()
        self.work = [None, None]
        self.count = 0          # circuits this rank has solved
        self.gather_ms = 0.0    # host time spent waiting for collectives (what the overlap did not hide)
        self.last_x = None

    def close(self):
        self.drain()
        if self.h is not None:
            self.h.close()
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _finish(self, slot):
        import time
        w = self.work[slot]
        if w is not None:
            t0 = time.perf_counter()
            w.wait()
            if self.dev.type == "cuda":
                import torch
                torch.cuda.current_stream(self.dev).synchronize()
            self.gather_ms += (time.perf_counter() - t0) * 1e3
            self.work[slot] = None

    def solve_next(self, reuse_symbolic=False):
        """One circuit on this rank; its x goes into the gather (when there is a group).  Returns the solver's info."""
        import torch
        info = 0
        if self.h is not None:
            info = self.h.run(self.dense, member=0, reuse_symbolic=reuse_symbolic)
        else:
            self.last_x = np.ascontiguousarray(self.solver(self.table), dtype=np.float64)
        if self.collective:
            slot = self.count & 1
            self._finish(slot)  # (the collective that used this slot two circuits ago: long finished)
            if self.h is not None and self.dev.type == "cuda":
                self.h.x_to_device(self.send[slot].data_ptr(), self.send[slot].numel() * 8)  # returns when copied
            elif self.h is not None:  # rehearsal: a GPU solve gathered over gloo, through host memory
                self.send[slot].copy_(torch.from_numpy(self.h.download_x()))
            else:
                self.send[slot].copy_(torch.from_numpy(self.last_x).to(self.dev))
            self.work[slot] = self.dist.all_gather_into_tensor(self.recv[slot], self.send[slot], async_op=True)
        self.count += 1
        return info

    def drain(self):
        """Wait for the collectives still in flight (the end of a timed region, before reading `latest`)."""
        for slot in (0, 1):
            self._finish(slot)

    def latest(self):
        """[world, n] host array: the last circuit's solution of every rank (own solution without a group)."""
        self.drain()
        if not self.collective:
            x = self.h.download_x() if self.h is not None else self.last_x
            return np.asarray(x, dtype=np.float64)[None, :]
        return self.recv[(self.count - 1) & 1].cpu().numpy().reshape(self.world, -1)


def solve_batch_distributed(table, values, sparse=True, device=None, solver=None, dist=None,
                            session=None):
    """Every rank passes the same `table` and the full `values`; each solves its
    shard and all ranks return the gathered [members, n] array."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return solve_members(table, values, sparse, device or 0, solver, session)
    with ShardedBatch(table, values.shape[0], dist, device, solver, session=session, sparse=sparse) as shard:
        shard.upload(values[shard.lo:shard.hi])
        shard.step()
        return shard.result()


def broadcast_table(table, dist, src=0):
    """Share rank `src`'s component table with every rank (object broadcast of
    the eight columns + sizes; sub-megabyte for the batch configurations)."""
    from .lowering import ComponentTable
    payload = [None]
    if dist.get_rank() == src:
        payload = [(table.ncomp, table.K, table.B,
                    {f: getattr(table, f) for f in ("type", "value", "a", "b", "c", "d", "drv", "k")})]
    dist.broadcast_object_list(payload, src=src)
    ncomp, K, B, cols = payload[0]
    out = ComponentTable(ncomp, K, B)
    for f, arr in cols.items():
        getattr(out, f)[:] = arr
    return out
