"""Child process of tests/test_distributed.py::test_rccl_single_rank_gather (run as a FRESH
process: RCCL is initialised before anything else touches the GPU).

One rank, backend "nccl" (= RCCL on ROCm), world_size 1: config 4's per-GPU shard (128 x
grid(100) with cfg4_values) goes through nodal_amd.batch.ShardedBatch -- nodal_run_batch,
nodal_batch_x_device into a torch tensor, all_gather_into_tensor on that device tensor --
i.e. the entry bench.py runs per rank at N = 8.  Prints one JSON line with what was checked."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = sys.argv[1]
    members = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from nodal_amd import generators as gen
    from nodal_amd.batch import ShardedBatch
    from tests.conftest import load_golden

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    N = 100
    table = gen.grid_table(N)
    vals = np.ones((members, table.ncomp))
    for b in range(members):
        vals[b, :-1] = gen.cfg4_values(b, N)
    with ShardedBatch(table, members, dist, 0, force_collective=True) as shard:
        shard.upload(vals)
        shard.step()
        shard.step()  # second step: block / gathered buffers reused, gather after a gather
        own = shard.own_block()
        everything = shard.result()
        on_device = bool(shard.gathered.is_cuda and shard.block.is_cuda)
        gather_ms = shard.gather_ms / 2
    case = next(c for c in load_golden("synth.json") + load_golden("synth_large.json")
                if c["name"] == "cfg4(100,b=3)")
    idx = np.array(case["x_idx"])
    ref = np.array(case["x_sparse_samples"])
    err3 = float(np.abs(everything[3][idx] - ref).max() / case["x_sparse_absmax"]) if members > 3 else None
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "members": members,
           "gathered_equals_block": bool(np.array_equal(everything, own)),
           "finite": bool(np.isfinite(everything).all()), "member3_normwise_error": err3,
           "tensors_on_device": on_device, "gather_ms": gather_ms,
           "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
