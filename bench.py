#!/usr/bin/env python3
"""Headline benchmark: RealNVP log_prob evals/sec at batch 2^20 per GPU.

A "step" is one pass of the hot path over one resident batch: the fused
coupling-chain kernel writes log_prob for every row and the per-block float64
partial sums, a second tiny kernel adds them in a fixed order, and the scalar
lands in pinned host memory.  N > 1: the K steps' local sums are all-reduced
(RCCL) in ONE collective before the clock stops -- a collective kernel beside
every step costs the chain kernel ~10 % (measured), one per batch of steps nothing.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by `python -m torch.distributed.run`; torch.distributed (gloo,
CPU) is used only for the rendezvous, the barrier and the max-over-ranks of the
timing.  All GPU work goes through libnf_hip.so (own stream, own RCCL
communicator); torch never touches the device.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measured/spec constants: /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense, spec (155 measured)
FLOP_PER_EVAL_L8 = 69632       # L * 2 * (2*64 + 64*64 + 64*2), L = 8
BYTES_PER_EVAL = 12            # 8 B in (x[2] f32) + 4 B out (log_prob f32)


def profiled_traffic(kernel_prefix="void nf::affine2d::chain_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS
    command (profiles/collect_r01.sh: separate FETCH_SIZE / WRITE_SIZE runs, reads doubled as
    MI355X_MICROARCH.md prescribes and profiles/calibrate_fetch.py confirms).  None if absent."""
    path = os.path.join(ROOT, "profiles", "r01", "final_bench_n1_pmc_summary.json")
    try:
        summary = json.load(open(path))
        for name, counters in summary.items():
            if name.startswith(kernel_prefix) and counters.get("hbm_bytes_per_launch"):
                return float(counters["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(layers, density):
    """The C restatement of the oracle (oracle/nf_oracle.c, OpenMP over the batch) timed on
    this host's cores: a reported baseline ("port"), not the target.  Bounded sample of the
    same workload (about 10 s).  The ONLY place this file touches oracle/."""
    from oracle import c_oracle as CO
    from oracle import flows_np as O
    spec = O.FlowSpec("affine", 2, layers, O.alternating_masks(2, len(layers), 1))
    rows = 1 << 20
    x = O.make_inputs(density, rows)
    threads = min(CO.max_threads(), usable_cpus())
    CO.affine_masked_log_prob(spec, x[:1 << 16], threads)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        CO.affine_masked_log_prob(spec, x, threads)
        reps += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = time.perf_counter() - t0
    return {"value": rows * reps / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": f"C/OpenMP float32 restatement (oracle/nf_oracle.c), {reps} x 2^20 rows of the same "
                      f"{len(spec.layers)}-layer flow on {density}, {threads} OpenMP threads "
                      f"(host reports {os.cpu_count()} logical CPUs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU (weak scaling)")
    ap.add_argument("--density", default="banana")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden-matmul", default="f32", choices=["f32", "bf16x3"],
                    help="nf_flow_set_option hidden_matmul for the timed steps (default: exact float32 MFMA)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra bf16x3 timing after the run")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run "
                     "--nproc-per-node N (one rank per GPU)")
        args.gpus = world

    from normalizing_flows_amd import _lib, densities, flow
    from normalizing_flows_amd.conditioner import make_conditioner

    lib = _lib.load()
    _lib.require_device()
    # NF_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank uses device 0; RCCL refuses two
    # ranks on one device, so this also exercises the gloo fallback below
    device = 0 if os.environ.get("NF_BENCH_SHARE_GPU") == "1" else local_rank
    _lib.check(lib.nf_set_device(device))

    dist = None
    comm = None
    rccl_error = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("NF_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist  # gloo: CPU-side rendezvous only
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from normalizing_flows_amd import sharding
        import torch
        try:
            comm = sharding.RcclComm.from_torch_distributed(dist)
        except Exception as exc:   # keep measuring: the per-rank partial NLLs are then summed over gloo at the end
            comm, rccl_error = None, f"{type(exc).__name__}: {exc}"
        ok = torch.tensor([1 if comm is not None else 0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0 and comm is not None:
            comm.close()
            comm, rccl_error = None, "another rank failed to initialise RCCL"

    # synthetic weights (random init of the architecture: haiku's truncated-normal hidden layers, a small
    # non-zero last Linear so that the flow is not the identity), the same on every rank
    layers = [make_conditioner((2,), [64, 64], 2, rng=np.random.default_rng(534 + l), final_std=0.02).linears
              for l in range(args.layers)]
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=args.layers, hidden=(64, 64))
    f = flow.CouplingFlow(cfg, layers)
    if comm is not None:
        f.attach_comm(comm)
    f.set_option("hidden_matmul", args.hidden_matmul)

    rows = args.rows
    rows_global = rows * world
    # every rank owns its own rows of the global batch (different seed per rank), drawn on the device by
    # the library's density generator (src/densities/*.py restated, nf_density_sample): resident in HBM
    xd = next(densities.get_density(args.density)(seed=2123 + rank, batch_size=rows, num_batches=1, device=True))
    lpd = flow.DeviceArray((rows,))

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_run():
        """W warmup steps, then exactly K steps between barrier + device synchronize; max over ranks."""
        for i in range(args.warmup):
            f.nll_enqueue(xd, rows_global, i % _lib.NF_NLL_SLOTS, lpd)
        f.synchronize()
        _lib.check(lib.nf_device_synchronize())
        barrier()
        t0 = time.perf_counter()
        f.timer_start()
        for i in range(args.steps):
            f.nll_enqueue(xd, rows_global, i % _lib.NF_NLL_SLOTS, lpd)
        ev = f.timer_stop()          # HIP events on the stream the kernels ran on
        f.synchronize()              # N > 1: the one all-reduce over the K steps' NLL sums runs here, inside the clock
        _lib.check(lib.nf_device_synchronize())
        w = time.perf_counter() - t0  # this rank's K steps, device idle again; MAX over ranks below
        barrier()
        result = f.nll_result((args.steps - 1) % _lib.NF_NLL_SLOTS)
        if dist is not None:
            import torch
            t = torch.tensor([w, ev], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w, ev = float(t[0]), float(t[1])
            if comm is None and world > 1:   # RCCL fallback: ranks hold partial means of the global batch
                r = torch.tensor([result], dtype=torch.float64)
                dist.all_reduce(r, op=dist.ReduceOp.SUM)
                result = float(r[0])
        return w, ev, result

    wall, ev_ms, nll = timed_run()
    variant = None
    if args.hidden_matmul == "f32" and not args.no_variants:
        # the same steps once more under the optional split-bf16 hidden matmul (same tolerance
        # against the oracle, tests/test_bf16x3_gpu.py); reported beside the headline, never as it
        f.set_option("hidden_matmul", "bf16x3")
        v_wall, v_ev, v_nll = timed_run()
        f.set_option("hidden_matmul", "f32")
        variant = {"hidden_matmul": "bf16x3", "value": rows_global * args.steps / v_wall, "unit": "evals/s",
                   "ms_per_step": v_wall * 1e3 / args.steps, "kernel_ms": v_ev / args.steps, "nll": v_nll,
                   "nll_minus_headline_nll": v_nll - nll}

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = rows_global * args.steps / wall
        # dominant kernel: the fused chain; its launch duration ~ event time per step
        # (the 1-block sum kernel and the 8-byte copy are < 1 % of it)
        kern_s = ev_ms * 1e-3 / args.steps
        flop_per_eval = FLOP_PER_EVAL_L8 * args.layers / 8
        achieved_tflops = rows * flop_per_eval / kern_s / 1e12
        out = {
            "metric": "log_prob evals/sec (batch=2^20 per GPU)", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.hidden_matmul == "f32" else "f32 (hidden matmul on 3-way bf16 splits, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"RealNVP {args.layers} affine coupling layers (MLP 2-64-64-2), 2-D "
                                   f"{args.density}, batch=2^{int(np.log2(rows))} per GPU, log_prob + NLL",
                       "rows_per_gpu": rows, "rows_global": rows_global, "kernel": f.kernel_name,
                       "parallelism": f"dp{world} (batch rows sharded, one RCCL all-reduce of the steps' f64 NLL sums)"
                                      if rccl_error is None else
                                      f"dp{world} (batch rows sharded; RCCL unavailable -- {rccl_error} -- so the "
                                      "per-rank partial NLLs were summed over gloo after the timed steps)"},
            "nll": nll,
            "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": profiled_traffic() if (rows == 1 << 20 and args.layers == 8) else None,
                         "traffic_unit": "bytes of HBM per launch (rocprofv3 PMC, profiles/r01); algorithmic "
                                         f"{rows * BYTES_PER_EVAL}",
                         "hbm_algorithmic_GBps": rows * BYTES_PER_EVAL / kern_s / 1e9,
                         "kernel_ms": kern_s * 1e3},
        }
        if args.hidden_matmul != "f32":
            # 6 bf16 MFMA terms per product of the hidden Linear, priced against the dense bf16 peak
            mfma_flop = rows * args.layers * 6 * 2 * 64 * 64
            out["roofline"].update({"achieved": mfma_flop / kern_s / 1e12, "peak": 2500.0,
                                    "frac": mfma_flop / kern_s / 1e12 / 2500.0,
                                    "note": "bf16 MFMA FLOP issued (6 terms) vs dense bf16 peak; the float32-"
                                            f"equivalent rate is {achieved_tflops:.1f} TFLOP/s"})
        if variant is not None:
            out["variants"] = [variant]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(layers, args.density)
        print(json.dumps(out))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
