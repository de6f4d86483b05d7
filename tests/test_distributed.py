"""The batch-sharding path with world_size 2 over gloo on CPU.  The HIP solver is
replaced by the oracle here (no GPU in this container); the sharding, padding and
gather logic under test is the code the nccl/RCCL path runs."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from nodal_amd import batch
from nodal_amd import generators as gen


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 16, 1024):
        for world in (1, 2, 3, 8):
            spans = [batch.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def oracle_solver(table, values, sparse):
    from oracle import nodal_oracle as oracle
    out = np.empty((values.shape[0], table.n))
    for i, v in enumerate(values):
        t = table.truncated(table.ncomp)
        t.value[:] = v
        G, A = oracle.assemble_fast(t)
        out[i] = oracle.solve(G.tocsr(), A, True)[0]
    return out


def _worker(rank, world, port, members, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = gen.grid_table(6) if rank == 0 else None
    table = batch.broadcast_table(table, dist, src=0)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, 6)
    out = batch.solve_batch_distributed(table, vals, True, solver=oracle_solver, dist=dist)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("members", [5, 8])
def test_two_rank_gloo_batch(members):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, members, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table = gen.grid_table(6)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, 6)
    want = oracle_solver(table, vals, True)
    for rank in (0, 1):
        assert np.array_equal(results[rank], want)  # every rank holds the whole batch


def _sharded_worker(rank, world, port, members, q):
    """The same ShardedBatch entry bench.py's config 4 runs per rank (oracle stand-in: no GPU here)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = gen.grid_table(6)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, 6)
    with batch.ShardedBatch(table, members, dist, solver=oracle_solver) as shard:
        shard.upload(vals[shard.lo:shard.hi])
        shard.step()
        first = shard.result()
        shard.step()  # buffers are reused from step to step
        q.put((rank, first, shard.result(), shard.own_block(), (shard.lo, shard.hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("members", [5, 8])
def test_two_rank_sharded_batch_steps(members):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, members, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table = gen.grid_table(6)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, 6)
    want = oracle_solver(table, vals, True)
    for rank, first, second, own, (lo, hi) in got:
        assert np.array_equal(first, want) and np.array_equal(second, want)
        assert np.array_equal(own, want[lo:hi])


def _uneven_worker(rank, world, port, members, sparse, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = gen.grid_table(4)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, 4)
    out = batch.solve_batch_distributed(table, vals, sparse, solver=oracle_solver, dist=dist)
    lo, hi = batch.shard_range(members, rank, world)
    q.put((rank, (lo, hi), np.array_equal(out, oracle_solver(table, vals, sparse)), out.shape))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,members,sparse", [(3, 1024, True), (8, 1024, True), (8, 5, True), (3, 7, False)])
def test_uneven_shards_pad_and_reassemble(world, members, sparse):
    """BASELINE config 4's 1024 members over 3 ranks (342 / 341 / 341: the all_gather needs equal blocks, short
    shards are padded), over the 8 ranks of a node, fewer members than ranks (three ranks hold nothing), and the
    dense switch -- which shards like the sparse one since round 4.  gloo on CPU with the oracle as the solver."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, members, sparse, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    spans = sorted(g[1] for g in got)
    assert spans[0][0] == 0 and spans[-1][1] == members
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for _rank, _span, same, shape in got:
        assert same and shape[0] == members


def test_local_device_index_wraps_to_the_visible_devices(monkeypatch):
    """A launcher that makes one device visible per rank: LOCAL_RANK 5 must select device 0 of 1."""
    import torch
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert batch.local_device_index(5) == 0
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert batch.local_device_index(5) == 5
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert batch.local_device_index(5) == 0


def test_single_process_sharded_batch_without_group():
    """No process group: one shard holding everything, no collective."""
    table = gen.grid_table(5)
    vals = np.ones((3, table.ncomp))
    with batch.ShardedBatch(table, 3, None, solver=oracle_solver) as shard:
        assert (shard.lo, shard.hi, shard.world) == (0, 3, 1) and shard.gathered is None
        shard.upload(vals)
        shard.step()
        assert np.array_equal(shard.result(), oracle_solver(table, vals, True))


@pytest.mark.gpu
def test_rccl_single_rank_gather():
    """Backend "nccl" (= RCCL) with world_size 1 on the one GPU of the box: config 4's per-GPU
    shard through ShardedBatch in a FRESH child process (RCCL initialised before any other
    GPU call there), all_gather_into_tensor on the device tensor nodal_batch_x_device filled;
    the gathered block must equal the rank's own and member 3 the reference's golden samples."""
    import json
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, child, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["backend"] == "nccl" and out["world_size"] == 1
    assert out["tensors_on_device"] and out["gathered_equals_block"] and out["finite"]
    assert out["member3_normwise_error"] <= 1e-9


@pytest.mark.gpu
def test_rccl_single_rank_independent_circuits():
    """The N > 1 headline path of bench.py on the one GPU of the box: ShardedCircuits with backend "nccl" (= RCCL),
    world_size 1 -- nodal_run, nodal_x_device into alternating device send buffers, asynchronous all_gather_into_tensor
    behind the next solve -- five different circuits, every gathered solution against the oracle (1e-9)."""
    import json
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_circuits_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, child, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["backend"] == "nccl" and out["world_size"] == 1 and out["circuits"] == 5
    assert out["tensors_on_device"] and out["gathered_equals_own_solution"]
    assert out["worst_normwise_error"] <= 1e-9


# ---- independent circuits over the ranks (bench.py --gpus N: config 3 at every N) ----

def _single_circuit_oracle(table):
    from oracle import nodal_oracle as oracle
    G, A = oracle.assemble_fast(table)
    return oracle.solve(G.tocsr(), A, True)[0]


def _rank_table(rank, side=7):
    """The rank's own circuit: the same grid topology (the gather needs equal n), its own resistances."""
    table = gen.grid_table(side)
    table.value[:-1] = gen.cfg4_values(100 + rank, side)
    return table


def _circuits_worker(rank, world, port, circuits, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = _rank_table(rank)
    seen = []
    with batch.ShardedCircuits(table, dist, solver=_single_circuit_oracle) as sc:
        for i in range(circuits):
            # every circuit of a rank differs from the last one: a stale send / receive slot would show
            table.value[-1] = 1.0 + i
            assert sc.solve_next() == 0
            if i in (0, circuits - 1):
                seen.append(sc.latest().copy())
        q.put((rank, seen, sc.count))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("circuits", [1, 5])
def test_two_rank_gloo_independent_circuits(circuits):
    """ShardedCircuits (the N > 1 headline of bench.py): every rank solves its own circuits, x of each finished
    circuit is all-gathered through two alternating slots with the collective left in flight -- every rank must
    hold every rank's LATEST solution (reference: one Circuit(netlist).solve() per circuit, nodal/nodal.py:306-336)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_circuits_worker, args=(r, 2, port, circuits, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, count in got:
        assert count == circuits
        for which, i in zip(seen, sorted({0, circuits - 1})):
            assert which.shape[0] == 2
            for r in (0, 1):
                t = _rank_table(r)
                t.value[-1] = 1.0 + i
                assert np.array_equal(which[r], _single_circuit_oracle(t))


def test_independent_circuits_without_a_group():
    table = _rank_table(0)
    with batch.ShardedCircuits(table, None, solver=_single_circuit_oracle) as sc:
        assert sc.solve_next() == 0 and not sc.collective
        assert np.array_equal(sc.latest(), _single_circuit_oracle(table)[None, :])


# ---- a member that raises on one rank must not leave the others inside the collective ----

def dense_oracle_solver(table, values, sparse):
    from oracle import nodal_oracle as oracle
    out = np.empty((values.shape[0], table.n))
    for i, v in enumerate(values):
        t = table.truncated(table.ncomp)
        t.value[:] = v
        G, A = oracle.assemble_fast(t)
        out[i] = oracle.solve(G.toarray(), A, False)[0]  # np.linalg.solve: raises LinAlgError when singular
    return out


def _failing_member_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nodal_amd.netlist import Netlist
    from nodal_amd.lowering import lower
    # two voltage sources across the same node pair: singular whatever the values (LinAlgError, dense)
    rows = [["r1", "R", "1", "1", "g"], ["r2", "R", "1", "1", "2"], ["r3", "R", "1", "2", "g"],
            ["e1", "E", "1", "2", "g"], ["e2", "E", "1", "2", "g"]]
    table = lower(Netlist.from_rows(rows))
    members = 4
    vals = np.tile(table.value, (members, 1))

    def solver(t, v, sparse):
        if rank == 1:  # (only rank 1's members are singular: rank 0's stand-in drops the second source)
            return dense_oracle_solver(t, v, sparse)
        return np.zeros((v.shape[0], t.n))

    outcome = "no exception"
    try:
        batch.solve_batch_distributed(table, vals, sparse=False, solver=solver, dist=dist)
    except np.linalg.LinAlgError as e:
        outcome = f"LinAlgError: {e}"
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_a_singular_dense_member_raises_on_every_rank():
    """Advisor's finding (round 4): BatchSolver.run(sparse=False) raises on the rank that owns the bad member
    BEFORE the all_gather; the other ranks used to wait in the collective until its timeout.  Now the owner goes
    through the gather with a NaN block and a status word, and every rank raises LinAlgError afterwards."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_member_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1].startswith("LinAlgError") and "Singular" in got[1]
    assert got[0].startswith("LinAlgError") and "rank 1" in got[0]
