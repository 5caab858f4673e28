"""One rank of the world-size-2 gloo tests in tests/test_abi_and_host.py.
    _gloo_worker.py RANK WORLD [nll | onesided | counts | preflight]"""
import sys

import numpy as np
import torch
import torch.distributed as dist

from normalizing_flows_amd.sharding import ShardedNLL
from oracle import flows_np as O


def handshake(mode, rank, world):
    """The failure paths bench.py relies on to exit instead of hanging inside RCCL (sharding.agree_all /
    agree_equal / RcclComm.from_torch_distributed).  Prints one line and exits 0 when the expected error came."""
    from normalizing_flows_amd import sharding
    try:
        if mode == "onesided":      # rank 1 alone reports a failure: BOTH ranks must raise, naming rank 1
            sharding.agree_all(dist, rank != 1, "set-up step", "" if rank != 1 else "librccl missing (simulated)")
        elif mode == "counts":      # the collective counts differ: both ranks must raise
            sharding.agree_equal(dist, 7 + rank, "collectives issued")
        elif mode == "preflight":   # no GPU here: the preflight fails on every rank before any ncclCommInitRank
            sharding.RcclComm.from_torch_distributed(dist)
        print(f"rank {rank} NO ERROR")
        code = 1
    except sharding.RanksDisagree as exc:
        print(f"rank {rank} RanksDisagree: {exc}")
        code = 0
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(code)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "nll"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode != "nll":
        handshake(mode, rank, world)
    spec = O.make_affine_spec(4, final_std=0.02)
    x = O.make_inputs("banana", 4097)

    def allreduce_sum(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])

    sharded = ShardedNLL(world, rank, allreduce_sum)
    nll = sharded.nll(x.shape[0], lambda b, e: O.log_prob(spec, x[b:e]).sum())
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} nll {nll!r}")


if __name__ == "__main__":
    main()
