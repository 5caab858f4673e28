"""One rank of the world-size-2 RCCL test in tests/test_multigpu_gpu.py (needs two GPUs; gloo carries the rendezvous).
    _rccl_worker.py RANK WORLD
Every rank: device RANK, an RCCL communicator, ITS rows of one global batch; the sharded NLL and one Adam step through
the gradient all-reduce.  Prints `rank R nll <v> w <checksum of the weights after the step> collectives <n>`."""
import sys

import numpy as np
import torch.distributed as dist

from normalizing_flows_amd import _lib, flow, sharding, training
from oracle import flows_np as O


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    lib = _lib.load()
    _lib.check(lib.nf_set_device(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = sharding.RcclComm.from_torch_distributed(dist)
    spec = O.make_affine_spec(4, final_std=0.02)
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
    f = flow.CouplingFlow(cfg, spec.layers)
    f.attach_comm(comm)
    x = O.make_inputs("banana", 4096)
    per = x.shape[0] // world
    mine = np.ascontiguousarray(x[rank * per:(rank + 1) * per])
    nll = f.nll(mine, rows_global=x.shape[0])
    f.train_step(mine, training.adam(1e-3), rows_global=x.shape[0])
    w = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in f.get_weights() for W, b in layer])
    n = f.collective_count()
    counts = sharding.agree_equal(dist, n, "collectives issued")  # raises on every rank when they differ
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} nll {nll!r} w {float(np.abs(w).sum())!r} collectives {n}")


if __name__ == "__main__":
    main()
