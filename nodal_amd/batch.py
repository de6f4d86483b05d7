"""Independent circuits across the GPUs of one node (BASELINE.json config 4).

The hot path shards only along the batch dimension: circuits (or the members of
a value sweep on one topology) are independent, so ranks never exchange matrix
data.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI; "gloo" on CPU for tests).  Collectives are used only at the edges:

  * optional broadcast of the shared component table from rank 0 (0.5 MB for
    grid(100)) when only rank 0 parsed the netlist;
  * all_gather of the per-rank solution blocks (128 x 9999 x 8 B = 10 MB per
    GPU for config 4): direct all-to-all traffic over the xGMI mesh, no
    reduction, no ring.
"""

import numpy as np


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` members for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def replicate_table(table, values):
    """One block-diagonal system holding every member of a value sweep.

    Independent circuits that share the ground node do not couple (the ground row is
    eliminated), so M members of a K-node, B-branch topology are one netlist with
    M*K nodes and M*B branch unknowns.  Solving them together keeps the GPU full
    (a single 1e4-node circuit occupies a few CUs) and costs one symbolic phase and
    one multigrid setup.  Unknown layout: member m owns x[m*K:(m+1)*K] and
    x[M*K + m*B : M*K + (m+1)*B]."""
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


def solve_members(table, values, sparse=True, device=0, solver=None):
    """Solve every row of `values` ([members, ncomp]) on one GPU with a shared
    symbolic phase.  Returns [members, n] float64.  `solver` lets tests inject
    a stand-in for the HIP handle."""
    if solver is not None:
        return solver(table, values, sparse)
    from . import _ffi
    if sparse and values.shape[0] > 1:
        # all members at once, as one block-diagonal system
        h = _ffi.Handle(device)
        try:
            h.upload(replicate_table(table, values))
            info = h.run(False)
            x = h.download_x()
            return split_solution(x, values.shape[0], table.K, table.B)
        finally:
            h.close()
    h = _ffi.Handle(device)
    try:
        h.upload(table)
        h.upload_values(values)
        out = None
        for i in range(values.shape[0]):
            info = h.run(not sparse, member=i, reuse_symbolic=(i > 0))
            if out is None:
                out = np.empty((values.shape[0], h.n))
            out[i] = h.download_x() if info == 0 else np.nan
        return out if out is not None else np.empty((0, table.n))
    finally:
        h.close()


def solve_batch_distributed(table, values, sparse=True, device=None, solver=None, dist=None):
    """Every rank passes the same `table` and the full `values`; each solves its
    shard and all ranks return the gathered [members, n] array."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return solve_members(table, values, sparse, device or 0, solver)
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(values.shape[0], rank, world)
    mine = solve_members(table, values[lo:hi], sparse, device if device is not None else rank,
                         solver)
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device if device is not None else rank) if use_cuda else "cpu"
    # all_gather needs equal block sizes: pad the short shards by one row
    width = -(-values.shape[0] // world)
    block = torch.full((width, table.n), float("nan"), dtype=torch.float64, device=dev)
    block[: hi - lo] = torch.from_numpy(mine).to(dev)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    out = np.empty((values.shape[0], table.n))
    for r, b in enumerate(blocks):
        rlo, rhi = shard_range(values.shape[0], r, world)
        out[rlo:rhi] = b[: rhi - rlo].cpu().numpy()
    return out


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
