"""BASELINE.json config 4 at full size on ONE GPU: the 1024 members of the value sweep as the eight shards
of 128 an 8-GPU node would take, one after the other, every shard through ShardedBatch (the entry bench.py
and the RCCL path use), sampled members against the CPU restatement of the reference.
    python tests/campaigns/cfg4_full.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nodal_amd import generators as gen
from nodal_amd.batch import ShardedBatch
from oracle import nodal_oracle as oracle

table = gen.grid_table(100)
TOTAL, PER = 1024, 128


worst = 0.0
t_all = 0.0
for rank in range(TOTAL // PER):
    vals = np.ones((PER, table.ncomp))
    for i in range(PER):
        vals[i, :-1] = gen.cfg4_values(rank * PER + i, 100)
    sb = ShardedBatch(table, PER, dist=None, device=0)
    sb.upload(vals)
    sb.step()
    t0 = time.perf_counter()
    sb.step()
    dt = time.perf_counter() - t0
    t_all += dt
    x = sb.own_block()
    for m in (0, 63, 127):
        G, A = oracle.assemble_fast(gen.grid_table(100, vals[m, :-1]))
        xo, _ = oracle.solve(G.tocsr(), A, True)
        err = np.linalg.norm(x[m] - xo) / np.linalg.norm(xo)
        worst = max(worst, err)
    sb.close()
    print(f"shard {rank}: 128 members in {dt * 1e3:.2f} ms, sampled members within {worst:.1e} of the oracle so far", flush=True)
print(f"1024 members: {t_all * 1e3:.1f} ms on one GPU = {TOTAL / t_all:.0f} circuits/s; worst sampled distance {worst:.1e}")
assert worst <= 1e-9
