"""One rank of tests/test_abi_and_host.py::test_sharded_nll_gloo_world2."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from normalizing_flows_amd.sharding import ShardedNLL
from oracle import flows_np as O


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
