"""Child process of tests/test_distributed.py::test_rccl_single_rank_independent_circuits (a FRESH process: RCCL is
initialised before anything else touches the GPU).

One rank, backend "nccl" (= RCCL on ROCm), world_size 1: the headline path of `bench.py --gpus N` -- this rank's own
circuits through nodal_amd.batch.ShardedCircuits: nodal_run per circuit, nodal_x_device into one of two device send
buffers, all_gather_into_tensor started asynchronously and left in flight behind the next solve.  Every circuit of the
sequence differs from the one before (another source current), so a stale slot would show.  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = sys.argv[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from nodal_amd import generators as gen
    from nodal_amd.batch import ShardedCircuits
    from oracle import nodal_oracle as oracle

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    N = 150
    table = gen.grid_table(N)
    worst, on_device, matches_own = 0.0, True, True
    with ShardedCircuits(table, dist, 0, force_collective=True) as sc:
        for i in range(5):
            table.value[-1] = 1.0 + i           # the current source: x scales with it
            sc.h.upload(table)
            assert sc.solve_next() == 0
            if i in (0, 3, 4):
                got = sc.latest()
                G, A = oracle.assemble_fast(table)
                want = oracle.solve(G.tocsr(), A, True)[0]
                worst = max(worst, float(np.abs(got[0] - want).max() / np.abs(want).max()))
                matches_own = matches_own and bool(np.array_equal(got[0], sc.h.download_x()))
        on_device = bool(sc.send[0].is_cuda and sc.recv[0].is_cuda)
        count, wait_ms = sc.count, sc.gather_ms
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "circuits": count,
           "worst_normwise_error": worst, "gathered_equals_own_solution": matches_own, "tensors_on_device": on_device,
           "host_wait_for_collectives_ms": wait_ms}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
