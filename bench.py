#!/usr/bin/env python3
"""Headline benchmark: RealNVP log_prob evals/sec at batch 2^20 per GPU.

A "step" is one pass of the hot path over one resident batch: the fused
coupling-chain kernel writes log_prob for every row and the per-block float64
partial sums, a second tiny kernel adds them in a fixed order, and the scalar
lands in pinned host memory.  N > 1: the K steps' local sums are all-reduced
(RCCL) in ONE collective before the clock stops -- a collective kernel beside
every step costs the chain kernel ~10 % (measured), one per batch of steps nothing.

    python bench.py --gpus N --steps K --warmup W

Timing: W untimed warm-up steps, then BLOCKS of exactly K steps, each bracketed by a
barrier + device synchronise on both sides, wall time = max over ranks.  One block
of K = 20 steps is 11 ms -- a single draw of a kernel whose own duration scatters
by +-8 % from launch to launch -- so blocks are repeated until >= 0.5 s of them have run
(at most 200); `value` / `ms_per_step` are the MEDIAN block's, `block_ms` carries
min / median / max and the count.

N > 1: one rank per GPU, launched by `python -m torch.distributed.run` (the driver) or,
typed as `python bench.py --gpus N` without a launcher, by this script itself
(`spawn_ranks`: the parent starts N fresh rank processes before anything touches the GPU,
relays rank 0's line, ends everybody if one rank fails or 600 s pass).  torch.distributed
(gloo, CPU) is used only for the rendezvous, the barrier and the max-over-ranks of the
timing.  All GPU work goes through libnf_hip.so (own stream, own RCCL
communicator); torch never touches the device.

Prints ONE JSON line on rank 0: the contract keys for the headline (BASELINE.json
configs[1]/[3]: 8 affine layers, banana, 2^20 rows per GPU) + `roofline` +
`cpu_baseline`, and at N = 1 `other_configs`: configs[0], [2] and [4] and the flow of src/real_nvp.py timed the
same way, and `next_rows`: sampling, the training step and the density generators.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measured/spec constants: /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense, spec (155 measured)
PEAK_HBM_GBPS = 8000.0         # HBM3E spec (6,290 measured with a float4 copy)
BYTES_PER_EVAL = 12            # 8 B in (x[2] f32) + 4 B out (log_prob f32)
# algorithmic (useful) FLOP per eval, SURVEY.md 8(d)
FLOP_AFFINE_PER_LAYER = 2 * (2 * 64 + 64 * 64 + 64 * 2)           # 8,704
FLOP_RQS12_PER_LAYER = 2 * (2 * 64 + 64 * 64 + 64 * 37)           # 13,184: the 37 parameters of the transformed dim
FLOP_MNIST_PER_LAYER = 2 * (784 * 512 + 2 * 512 * 512 + 512 * 9800)  # 11,886,592: only the 392 transformed pixels


def profiled_traffic(kernel_prefix, summary_name):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this command
    (profiles/collect_r03.sh: separate FETCH_SIZE / WRITE_SIZE runs, reads doubled as
    MI355X_MICROARCH.md prescribes and profiles/calibrate_fetch.py confirms).  None if absent."""
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, summary_name)
        try:
            summary = json.load(open(path))
            for name, counters in summary.items():
                if name.startswith(kernel_prefix) and counters.get("hbm_bytes_per_launch"):
                    return float(counters["hbm_bytes_per_launch"]), f"profiles/{rnd}/{summary_name}"
        except Exception:
            pass
    return None, None


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


def cpu_baseline(layers, layers4, density):
    """The oracle timed on this host's cores: a reported baseline ("port"), not the target.  Bounded samples of
    the same workloads (about 20 s in all).  The ONLY place this file touches oracle/.
      value:  C/OpenMP restatement (oracle/nf_oracle.c) on the headline (2^20 rows, 8 layers);
      others: the same C port at BASELINE configs[0] (4096 rows, 4 layers -- the reference's own CPU-runnable
              case, src/real_nvp_toy_densities.py:139-151) and the NumPy float32 restatement (oracle/flows_np.py,
              BLAS threads as the host gives them) on the headline batch."""
    from oracle import c_oracle as CO
    from oracle import flows_np as O
    spec = O.FlowSpec("affine", 2, layers, O.alternating_masks(2, len(layers), 1))
    spec4 = O.FlowSpec("affine", 2, layers4, O.alternating_masks(2, len(layers4), 1))
    rows = 1 << 20
    x = O.make_inputs(density, rows)
    threads = min(CO.max_threads(), usable_cpus())
    CO.affine_masked_log_prob(spec, x[:1 << 16], threads)  # warm

    def timed(fn, budget):
        t0 = time.perf_counter()
        reps = 0
        while True:
            fn()
            reps += 1
            if time.perf_counter() - t0 > budget:
                break
        return reps, time.perf_counter() - t0

    reps, dt = timed(lambda: CO.affine_masked_log_prob(spec, x, threads), 10.0)
    x4k = x[:4096]
    reps4, dt4 = timed(lambda: CO.affine_masked_log_prob(spec4, x4k, threads), 3.0)
    reps_np, dt_np = timed(lambda: O.log_prob(spec, x, np.float32), 4.0)
    return {"value": rows * reps / dt, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": f"C/OpenMP float32 restatement (oracle/nf_oracle.c), {reps} x 2^20 rows of the same "
                      f"{len(spec.layers)}-layer flow on {density}, {threads} OpenMP threads "
                      f"(host reports {os.cpu_count()} logical CPUs)",
            "others": [
                {"what": "C/OpenMP port at BASELINE configs[0]: 4 layers, 4096 rows per call", "value": 4096 * reps4 / dt4,
                 "unit": "evals/s", "cores": int(threads), "ms_per_call": dt4 / reps4 * 1e3, "calls": reps4},
                {"what": "NumPy float32 restatement (oracle/flows_np.py), headline batch 2^20 x 8 layers",
                 "value": rows * reps_np / dt_np, "unit": "evals/s", "cores": usable_cpus(), "calls": reps_np}]}


def parity_check(kind, layers, x, got, rqs=None):
    """The ACHIEVED error of the measured kernel, for the record: `got` = the log_prob values the timed steps left in
    HBM for the first rows of the resident batch, against the float64 oracle on the same rows and weights (outside the
    timed region; the oracle is the checker, never the thing measured -- with cpu_baseline() the only place this file
    touches oracle/).  scaled = |got - ref| / max(1, |ref|), the tolerance of every parity test;
    max_abs_where_ref_le_8: the largest ABSOLUTE error over rows with |ref| <= 8 (north_star's literal 1e-5)."""
    from oracle import flows_np as O
    if kind == "affine":
        ref = O.log_prob(O.FlowSpec("affine", 2, layers, O.alternating_masks(2, len(layers), 1)), x)
    elif kind == "rqs":
        ref = O.log_prob(O.FlowSpec("rqs", 2, layers, O.alternating_masks(2, len(layers), 1), "normal", rqs[0], rqs[1],
                                    rqs[2]), x)
    else:  # the flow of src/real_nvp.py: params[layer] = [[W, b], [W, b]], flips F, T, F, T ...
        ref = O.a_log_prob_chain([[(W, b) for W, b in net] for net in layers], [bool(l & 1) for l in range(len(layers))], x)
    got = np.asarray(got, np.float64)
    err = np.abs(got - ref)
    scaled = err / np.maximum(1.0, np.abs(ref))
    small = np.abs(ref) <= 8.0
    return {"rows": int(len(ref)), "against": "oracle/flows_np.py float64 (parity unpinned: DESIGN.md 1c)",
            "max_scaled": float(scaled.max()), "rms_scaled": float(np.sqrt((scaled ** 2).mean())),
            "outside_1e-5": int((scaled > 1e-5).sum()), "rows_abs_ref_le_8": int(small.sum()),
            "max_abs_where_ref_le_8": float(err[small].max()) if small.any() else None,
            "tolerance": "1e-5 * max(1, |ref|)"}


def spawn_ranks(n, argv, worker=None, timeout_s=None):
    """`python bench.py --gpus N` typed as it stands (no launcher, WORLD_SIZE unset, N > 1): this process becomes the
    PARENT of N fresh rank processes and never touches the GPU itself -- nothing of libnf_hip, HIP or torch is loaded
    before or after the children start, so no process that has initialised a device is ever re-executed.  Every child
    is this same script (or `worker`, NF_BENCH_WORKER: the stub of tests/test_abi_and_host.py) with RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1 and a free MASTER_PORT in its environment -- what `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N` would have set -- in its own process group.  Rank 0's stdout
    (the one JSON line) is relayed when ALL ranks have exited with 0; the first rank to fail, or NF_BENCH_SPAWN_TIMEOUT
    (600 s) passing, ends every other rank (SIGTERM to the group, SIGKILL after 5 s) and the parent exits non-zero
    (the failed rank's code, 124 for the limit) without printing a result line.  Returns the exit code."""
    import signal
    import socket
    import subprocess
    import tempfile
    worker = worker or os.environ.get("NF_BENCH_WORKER") or os.path.abspath(__file__)
    if timeout_s is None:
        timeout_s = float(os.environ.get("NF_BENCH_SPAWN_TIMEOUT", "600"))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []

    def end_all():
        for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            live = [p for p in procs if p.poll() is None]
            if not live:
                return
            for p in live:
                try:
                    os.killpg(p.pid, sig)     # the child's own group (start_new_session): it and whatever it started
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + wait
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)

    code = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
            procs.append(subprocess.Popen([sys.executable, worker] + list(argv), env=env, start_new_session=True,
                                          stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))
        deadline = time.monotonic() + timeout_s
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                r, c = failed[0]
                print(f"bench.py: rank {r} of {n} exited with {c}; ending the other ranks", file=sys.stderr)
                code = c if 0 < c < 256 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                running = [r for r, c in enumerate(codes) if c is None]
                print(f"bench.py: ranks {running} of {n} still running after {timeout_s:.0f} s; ending them",
                      file=sys.stderr)
                code = 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        code = 130
    finally:
        end_all()
    if code == 0:
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU (weak scaling)")
    ap.add_argument("--density", default="banana")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step block until this much of it has run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden-matmul", default="f32", choices=["f32", "bf16x3"],
                    help="nf_flow_set_option hidden_matmul for the timed steps (default: exact float32 MFMA)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra bf16x3 timing after the run")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs[0], [2], [4] (N = 1 only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # typed without a launcher: become the parent of N rank processes (before anything here touches the GPU)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # under a launcher its world size is the truth
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
        import datetime
        import torch.distributed as dist  # gloo: CPU-side rendezvous only
        # (a rank that died before the rendezvous must not leave the others waiting for gloo's default 30 minutes)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        from normalizing_flows_amd import sharding
        try:
            # every step that can fail on one rank alone (device, librccl, the id, the init itself) is agreed on
            # over gloo inside; a failure raises RanksDisagree on EVERY rank -- nobody is left inside ncclCommInitRank
            comm = sharding.RcclComm.from_torch_distributed(dist)
        except sharding.RanksDisagree as exc:
            # keep measuring: the per-rank partial NLLs are then summed over gloo at the end (the JSON line says so)
            comm, rccl_error = None, str(exc)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(*vals):
        if dist is None:
            return vals
        import torch
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def measure(f, step, steps, warmup, min_seconds):
        """`step(i)` enqueues one pass on handle f.  Returns (median block wall s, its event ms, all walls)."""
        for i in range(warmup):
            step(i)
        f.synchronize()
        _lib.check(lib.nf_device_synchronize())

        def block():
            barrier()
            t0 = time.perf_counter()
            f.timer_start()
            for i in range(steps):
                step(i)
            ev = f.timer_stop()          # HIP events on the stream the kernels ran on
            f.synchronize()              # N > 1: the one all-reduce over the K steps' NLL sums runs here, inside the clock
            _lib.check(lib.nf_device_synchronize())
            w = time.perf_counter() - t0  # this rank's K steps, device idle again; MAX over ranks below
            barrier()
            return max_over_ranks(w, ev)

        blocks = [block()]
        # the same count on every rank: the first block's wall is already the max over ranks
        n = max(1, min(200, int(math.ceil(min_seconds / max(blocks[0][0], 1e-6)))))
        blocks += [block() for _ in range(n - 1)]
        walls = sorted(b[0] for b in blocks)
        med = sorted(blocks)[len(blocks) // 2]
        return med[0], med[1], walls

    def block_stats(walls, steps):
        return {"n": len(walls), "min": walls[0] * 1e3 / steps, "median": walls[len(walls) // 2] * 1e3 / steps,
                "max": walls[-1] * 1e3 / steps, "unit": "ms per step, per block of K steps"}

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

    def headline_step(i):
        f.nll_enqueue(xd, rows_global, i % _lib.NF_NLL_SLOTS, lpd)

    def headline_run():
        wall, ev_ms, walls = measure(f, headline_step, args.steps, args.warmup, args.min_seconds)
        result = f.nll_result((args.steps - 1) % _lib.NF_NLL_SLOTS)
        if dist is not None and comm is None and world > 1:   # RCCL fallback: ranks hold partial means
            import torch
            r = torch.tensor([result], dtype=torch.float64)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            result = float(r[0])
        return wall, ev_ms, walls, result

    wall, ev_ms, walls, nll = headline_run()
    # what the timed steps left in HBM for the first 4096 rows (checked against the oracle after the measurements)
    parity_rows = (xd.rows(0, 4096).to_numpy(), lpd.rows(0, 4096).to_numpy()) if rank == 0 and world == 1 and rows >= 4096 else None
    variant = None
    if args.hidden_matmul == "f32" and not args.no_variants:
        # the same steps once more under the optional split-bf16 hidden matmul (same tolerance
        # against the oracle, tests/test_bf16x3_gpu.py); reported beside the headline, never as it
        f.set_option("hidden_matmul", "bf16x3")
        v_wall, v_ev, v_walls, v_nll = headline_run()
        f.set_option("hidden_matmul", "f32")
        variant = {"hidden_matmul": "bf16x3", "value": rows_global * args.steps / v_wall, "unit": "evals/s",
                   "ms_per_step": v_wall * 1e3 / args.steps, "kernel_ms": v_ev / args.steps, "nll": v_nll,
                   "nll_minus_headline_nll": v_nll - nll, "block_ms": block_stats(v_walls, args.steps)}

    # every rank must have issued the same number of collectives by now (one batched all-reduce per block and
    # result): a rank that skipped one would deadlock the NEXT collective -- say so instead
    coll_error = None
    if dist is not None and world > 1:
        from normalizing_flows_amd import sharding
        try:
            sharding.agree_equal(dist, f.collective_count(), "collectives issued by the headline handle")
        except sharding.RanksDisagree as exc:
            coll_error = str(exc)

    other, nxt, side_errors = [], [], []
    if world == 1 and not args.no_other_configs and args.hidden_matmul == "f32":
        # the contract line must not depend on the extra measurements: a failure there is reported, not raised
        for name, fn in (("other_configs", other_configs), ("next_rows", next_rows)):
            try:
                res = fn(args, lib, _lib, flow, densities, make_conditioner, measure, block_stats)
            except Exception as exc:  # noqa: BLE001
                side_errors.append(f"{name}: {type(exc).__name__}: {exc}")
                res = []
            if name == "other_configs":
                other = res
            else:
                nxt = res

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = rows_global * args.steps / wall
        # dominant kernel: the fused chain; its launch duration ~ event time per step
        # (the 1-block sum kernel and the 8-byte copy are < 1 % of it)
        kern_s = ev_ms * 1e-3 / args.steps
        flop_per_eval = FLOP_AFFINE_PER_LAYER * args.layers
        achieved_tflops = rows * flop_per_eval / kern_s / 1e12
        traffic, traffic_src = profiled_traffic("void nf::affine2d::chain_kernel", "final_bench_n1_pmc_summary.json")
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
            "block_ms": block_stats(walls, args.steps),
            "nll": nll,
            "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic if (rows == 1 << 20 and args.layers == 8) else None,
                         "traffic_unit": f"bytes of HBM per launch (rocprofv3 PMC, {traffic_src}); algorithmic "
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
        if world == 1 and not args.no_cpu_baseline and parity_rows is not None:
            try:
                out["parity"] = parity_check("affine", layers, parity_rows[0], parity_rows[1])
            except Exception as exc:  # noqa: BLE001
                out.setdefault("side_errors", []).append(f"parity: {type(exc).__name__}: {exc}")
        if variant is not None:
            out["variants"] = [variant]
        if other:
            out["other_configs"] = other
        if nxt:
            out["next_rows"] = nxt
        if side_errors:
            out["side_errors"] = side_errors
        if coll_error is not None:
            out["error"] = coll_error
        if not args.no_cpu_baseline and world == 1:
            layers4 = [make_conditioner((2,), [64, 64], 2, rng=np.random.default_rng(534 + l), final_std=0.02).linears
                       for l in range(4)]
            try:
                out["cpu_baseline"] = cpu_baseline(layers, layers4, args.density)
            except Exception as exc:  # noqa: BLE001 -- e.g. the oracle's C library missing on the box
                out.setdefault("side_errors", []).append(f"cpu_baseline: {type(exc).__name__}: {exc}")
        print(json.dumps(out))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if coll_error is not None:
        sys.exit(f"bench.py: {coll_error}")


def other_configs(args, lib, _lib, flow, densities, make_conditioner, measure, block_stats):
    """BASELINE.json configs[0], [2] and [4] on this GPU, timed like the headline (blocks of steps, median),
    each with its useful-FLOP roofline.  Synthetic weights of the architecture, inputs resident in HBM."""
    res = []

    def run(name, cfg, layers, xd, rows, steps, flop_per_eval, note, dominant=None, hbm_bound=False, parity=None):
        f = flow.CouplingFlow(cfg, layers)
        lpd = flow.DeviceArray((rows,))
        step = lambda i: f.nll_enqueue(xd, rows, i % _lib.NF_NLL_SLOTS, lpd)  # noqa: E731
        wall, ev_ms, walls = measure(f, step, steps, max(2, min(args.warmup, steps)), args.min_seconds)
        nll = f.nll_result((steps - 1) % _lib.NF_NLL_SLOTS)
        kern_s = ev_ms * 1e-3 / steps
        tf = rows * flop_per_eval / kern_s / 1e12
        res.append({"config": name, "kernel": f.kernel_name, "rows": rows, "steps": steps,
                    "value": rows * steps / wall, "unit": "evals/s", "ms_per_step": wall * 1e3 / steps,
                    "kernel_ms": kern_s * 1e3, "block_ms": block_stats(walls, steps), "nll": nll,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / PEAK_F32_MFMA_TFLOPS, "flop_per_eval_useful": flop_per_eval,
                                 "note": note}})
        if hbm_bound:   # no matrix work: priced against HBM (12 algorithmic bytes per eval)
            gbps = rows * BYTES_PER_EVAL / kern_s / 1e9
            res[-1]["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                   "frac": gbps / PEAK_HBM_GBPS, "bytes_per_eval": BYTES_PER_EVAL, "note": note}
        if dominant is not None:   # HBM bytes per launch of the dominant kernel, from the committed PMC passes
            traffic, src = profiled_traffic(*dominant)
            res[-1]["roofline"].update({"traffic": traffic, "traffic_unit": f"bytes of HBM per launch of {dominant[0]} "
                                                                            f"(rocprofv3 PMC, {src})"})
        if parity is not None and not args.no_cpu_baseline:
            try:
                n = min(rows, 4096)
                res[-1]["parity"] = parity_check(parity[0], layers, xd.rows(0, n).to_numpy(), lpd.rows(0, n).to_numpy(),
                                                 parity[1] if len(parity) > 1 else None)
            except Exception as exc:  # noqa: BLE001
                res[-1]["parity_error"] = f"{type(exc).__name__}: {exc}"
        f.close()

    # configs[0]: 4 affine layers, banana, 4096 rows (the reference's CPU-runnable case; latency-bound on a GPU)
    layers = [make_conditioner((2,), [64, 64], 2, rng=np.random.default_rng(534 + l), final_std=0.02).linears
              for l in range(4)]
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
    xd = next(densities.get_density("banana")(seed=2123, batch_size=4096, num_batches=1, device=True))
    run("configs[0]: RealNVP 4 affine layers, banana, batch=4096", cfg, layers, xd, 4096, max(args.steps, 200),
        4 * FLOP_AFFINE_PER_LAYER, "16 of the chip's 1024 SIMDs hold a tile: latency-bound, not a roofline case",
        parity=("affine",))
    # configs[2]: spline coupling ("MAF" in the reference), 8 layers, K = 12 on [-8, 8], gaussian_blobs, 2^20 rows
    layers = [make_conditioner((2,), [64, 64], 37, per_dim=True, rng=np.random.default_rng(534 + l),
                               final_std=0.05).linears for l in range(8)]
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), rqs_bins=12,
                          rqs_range_min=-8.0, rqs_range_max=8.0)
    xd = next(densities.get_density("gaussian_blobs")(seed=2123, batch_size=1 << 20, num_batches=1, device=True))
    run("configs[2]: spline coupling (maf_toy_densities.py) 8 layers K=12, gaussian_blobs, batch=2^20", cfg, layers,
        xd, 1 << 20, args.steps, 8 * FLOP_RQS12_PER_LAYER,
        "useful FLOP = the 37 spline parameters of the transformed dim; the kernel computes 26 of them on the MFMA "
        "and the 2 slope logits of the sample's bin on the VALU",
        dominant=("void nf::rqs2d::chain_kernel", "config_c_pmc_summary.json"), parity=("rqs", (12, -8.0, 8.0)))
    del xd
    # the flow of src/real_nvp.py (north_star's other drop-in script; not a BASELINE.json config): split halves + flip,
    # MLP 1-4-2 with leaky relu, 8 layers, 2^24 rows -- 12 FMA per sample and layer, no matrix work
    rng = np.random.default_rng(534)
    layers = [[((rng.standard_normal((1, 4)) * 0.1).astype(np.float32), np.zeros(4, np.float32)),
               ((rng.standard_normal((4, 2)) * 0.1).astype(np.float32), np.zeros(2, np.float32))] for _ in range(8)]
    cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,),
                          activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
    xd = next(densities.get_density("energy2")(seed=2123, batch_size=1 << 24, num_batches=1, device=True))
    run("src/real_nvp.py flow: split-halves coupling, MLP 1-4-2, 8 layers, energy2, batch=2^24", cfg, layers, xd, 1 << 24,
        args.steps, 8 * 2 * (4 + 8),
        "vector-issue bound: 17 instructions per sample and layer (activation as alpha a + beta |a|: three fma per hidden "
        "unit), counters in profiles/r03/config_a_pmc_summary.json; HBM moves 12 B per eval", dominant=("void nf::split2d::chain_kernel", "config_a_pmc_summary.json"),
        hbm_bound=True, parity=("split",))
    del xd
    # configs[4]: MNIST shape, 6 layers, 784-D, MLP 512x3, K = 8 on [0, 1], Uniform base, 2^16 rows
    layers = [make_conditioner((784,), [512, 512, 512], 25, per_dim=True, rng=np.random.default_rng(534 + l),
                               final_std=0.05).linears for l in range(6)]
    cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=6, hidden=(512, 512, 512), rqs_bins=8,
                          rqs_range_min=0.0, rqs_range_max=1.0, base=_lib.BASE_UNIFORM01)
    rows = 1 << 16
    x = (np.random.default_rng(2123).integers(0, 256, (rows, 784)) / 256.0).astype(np.float32)  # prepare_data's k / 256
    xd = flow.DeviceArray.from_numpy(x)
    run("configs[4]: spline coupling (maf_mnist.py) 6 layers, 784-D, MLP 512x3, K=8, batch=2^16", cfg, layers, xd, rows,
        max(2, min(args.steps, 5)), 6 * FLOP_MNIST_PER_LAYER,
        "whole call (4 GEMMs + spline epilogue per layer); useful FLOP = transformed pixels' columns only",
        dominant=("void nf::generic::gemm_rqs_kernel", "config_d_pmc_summary.json"))
    return res


def next_rows(args, lib, _lib, flow, densities, make_conditioner, measure, block_stats):
    """The callers either side of the path, measured like the configs above: sampling (f1), the
    training step (f2) and the on-device density generators (f3).  N = 1 only; never the contract `value`."""
    import ctypes as C
    from normalizing_flows_amd import training
    res = []
    steps = args.steps
    warm = max(2, min(args.warmup, steps))

    def line(row, what, kernel, rows, wall, ev_ms, walls, unit, roofline):
        res.append({"row": row, "what": what, "kernel": kernel, "rows": rows, "steps": steps,
                    "value": rows * steps / wall, "unit": unit, "ms_per_step": wall * 1e3 / steps,
                    "kernel_ms": ev_ms / steps, "block_ms": block_stats(walls, steps), "roofline": roofline})

    def mfma(rows, flop, ev_ms, note):
        tf = rows * flop / (ev_ms * 1e-3 / steps) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_F32_MFMA_TFLOPS, "flop_per_sample_useful": flop, "note": note}

    def hbm(rows, nbytes, ev_ms, note):
        gbps = rows * nbytes / (ev_ms * 1e-3 / steps) / 1e9
        return {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                "bytes_per_sample": nbytes, "note": note}

    rows = 1 << 20
    aff_layers = [make_conditioner((2,), [64, 64], 2, rng=np.random.default_rng(534 + l), final_std=0.02).linears
                  for l in range(8)]
    aff_cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64))
    rqs_layers = [make_conditioner((2,), [64, 64], 37, per_dim=True, rng=np.random.default_rng(534 + l),
                                   final_std=0.05).linears for l in range(8)]
    rqs_cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=8, hidden=(64, 64), rqs_bins=12,
                              rqs_range_min=-8.0, rqs_range_max=8.0)

    # f1: Transformed.sample -- Philox base draw + the generative walk (Chain.forward of the scripts' Inverse(Chain))
    # in one call, samples left in HBM (src/real_nvp_toy_densities.py:126-136,197-203 draws 10^6)
    for name, cfg, layers, flop in (("affine", aff_cfg, aff_layers, 8 * FLOP_AFFINE_PER_LAYER),
                                    ("spline K=12", rqs_cfg, rqs_layers, 8 * FLOP_RQS12_PER_LAYER)):
        f = flow.CouplingFlow(cfg, layers)
        out = flow.DeviceArray((rows, 2))
        step = lambda i: _lib.check(lib.nf_sample(f._h, C.c_uint64(1000 + i), rows, out.ptr, _lib.MEM_DEVICE))  # noqa: E731
        wall, ev_ms, walls = measure(f, step, steps, warm, args.min_seconds)
        line("f1", f"sample(2^20), 8 {name} layers: base draw + generative walk, device array out", f.kernel_name,
             rows, wall, ev_ms, walls, "samples/s",
             mfma(rows, flop, ev_ms, "same conditioner FLOP as log_prob; the walk is the chain's other direction"))
        f.close()
        out.free()

    # f2: the training step (`update`: value_and_grad + optimiser, src/real_nvp_toy_densities.py:164-173) on
    # resident batches.  Useful FLOP = 3 x forward (backward = data gradient + weight gradient); the fused affine
    # backward stores no activations and recomputes the hidden Linear (a 4th product, not counted).
    xd = next(densities.get_density("banana")(seed=2123, batch_size=rows, num_batches=1, device=True))
    x128 = flow.DeviceArray.from_numpy(xd.to_numpy()[:128])
    for name, cfg, layers, flop, opt in (("affine (headline flow), Adam", aff_cfg, aff_layers, 8 * FLOP_AFFINE_PER_LAYER,
                                          training.adam(1e-4)),
                                         ("spline K=12 (fused backward, rqs2d_bwd), Adam", rqs_cfg, rqs_layers,
                                          8 * FLOP_RQS12_PER_LAYER, training.adam(1e-4))):
        for n, x in ((rows, xd), (128, x128)):
            f = flow.CouplingFlow(cfg, layers)
            step = lambda i: f.train_step(x, opt, want_nll=False)  # noqa: E731
            wall, ev_ms, walls = measure(f, step, steps, warm, args.min_seconds if n == rows else 0.2)
            line("f2", f"training step (forward + backward + optimiser), 8 layers {name}, batch {n}", f.kernel_name, n,
                 wall, ev_ms, walls, "samples/s",
                 mfma(n, 3 * flop, ev_ms, "useful FLOP = 3 x the forward's; batch 128 (the reference's) is launch-bound"))
            f.close()
    rng = np.random.default_rng(534)
    a_layers = [[((rng.standard_normal((1, 4)) * 0.1).astype(np.float32), np.zeros(4, np.float32)),
                 ((rng.standard_normal((4, 2)) * 0.1).astype(np.float32), np.zeros(2, np.float32))] for _ in range(8)]
    a_cfg = flow.FlowConfig(kind=_lib.KIND_AFFINE_SPLIT, event_dim=2, num_layers=8, hidden=(4,),
                            activation=_lib.ACT_LEAKY_RELU, first_mask_parity=0)
    f = flow.CouplingFlow(a_cfg, a_layers)
    opt = training.sgd(1e-3)
    xd.free()
    x128.free()
    rows24 = 1 << 24   # (75 us per step at 2^20 rows is launch and tail: 1.4e10 samples/s)
    xa = next(densities.get_density("energy2")(seed=2123, batch_size=rows24, num_batches=1, device=True))
    step = lambda i: f.train_step(xa, opt, want_nll=False)  # noqa: E731
    wall, ev_ms, walls = measure(f, step, steps, warm, args.min_seconds)
    line("f2", "training step (forward + backward + SGD), src/real_nvp.py flow (MLP 1-4-2), 8 layers, batch 2^24",
         f.kernel_name, rows24, wall, ev_ms, walls, "samples/s",
         hbm(rows24, 8 + 8 + 16 + 8, ev_ms, "forward reads x and writes z (16 B), the one backward launch reads z (8 B): "
             "no per-layer state through HBM; vector-issue bound"))
    f.close()
    xa.free()

    # b/d rows at the reference's own batch: log_prob of the 8-layer MNIST-shaped flow on 128 resident rows
    # (src/maf_mnist.py:141-146 evaluates at batch 128).  Weight-bound: 184 MB of live weights per call
    mn_layers = [make_conditioner((784,), [512, 512, 512], 25, per_dim=True, rng=np.random.default_rng(534 + l),
                                  final_std=0.01).linears for l in range(8)]
    mn_cfg = flow.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=8, hidden=(512, 512, 512), rqs_bins=8,
                             rqs_range_min=0.0, rqs_range_max=1.0, base=_lib.BASE_UNIFORM01)
    f = flow.CouplingFlow(mn_cfg, mn_layers)
    xm = flow.DeviceArray.from_numpy((np.random.default_rng(1).integers(0, 256, (128, 784)) / 256.0).astype(np.float32))
    lpm = flow.DeviceArray((128,))
    step = lambda i: f.log_prob(xm, lpm)  # noqa: E731
    wall, ev_ms, walls = measure(f, step, steps, warm, 0.2)
    live = 8 * (392 * 512 + 512 + 2 * (512 * 512 + 512) + 512 * 9800 + 9800)
    line("d1-d2", "log_prob, MNIST shape of src/maf_mnist.py: 8 layers, 784-D, batch 128 (the reference's evaluation batch)",
         f.kernel_name, 128, wall, ev_ms, walls, "evals/s",
         hbm(1, live * 4, ev_ms, "per call: the live weights streamed once; 4 dependent launches per layer (three one-launch "
             "Linears of gemm_small_m16_kernel, the last Linear + spline in linear_rqs_small_m_kernel): launch boundaries "
             "and what one CU can pull, DESIGN.md 3.8"))
    f.close()
    lpm.free()

    # f2, image shape: the reference's own MNIST training configuration (src/maf_mnist.py:34-89,159-168: 8 layers, 784-D,
    # MLP 512 x 3, K = 8, batch 128) -- weight-bound: every step reads the 46 M live weights twice, writes their gradient
    # and runs Adam over them (10 floats of traffic per live weight)
    f = flow.CouplingFlow(mn_cfg, mn_layers)
    del mn_layers
    opt = training.adam(1e-4)
    step = lambda i: f.train_step(xm, opt, want_nll=False)  # noqa: E731
    wall, ev_ms, walls = measure(f, step, steps, warm, 0.2)
    live = 8 * (392 * 512 + 512 + 2 * (512 * 512 + 512) + 512 * 9800 + 9800)
    line("f2", "training step (forward + backward + Adam), MNIST shape of src/maf_mnist.py: 8 layers, 784-D, batch 128",
         f.kernel_name, 128, wall, ev_ms, walls, "samples/s",
         hbm(1, live * 4 * 10, ev_ms, "per step: the live weights read by the forward and the backward walk, their gradient "
             "written, Adam's 7 floats per weight; launch-bound (about 150 dependent launches per step)"))
    f.close()
    xm.free()

    # f3: the toy-density generators on the device (src/densities/*.py restated; Philox4x32-10), 8 B written per row;
    # 2^24 rows (a 2^20-row call is 18 us of launch and tail: profiles/scratch/density_rate.py)
    rows = 1 << 24
    for dens in ("banana", "gaussian_blobs"):
        out = flow.DeviceArray((rows, 2))
        call = lambda i: _lib.check(lib.nf_density_sample(densities.DENSITIES[dens], C.c_uint64(77 + i), rows, out.ptr,  # noqa: E731
                                                          _lib.MEM_DEVICE))
        for i in range(warm):
            call(i)
        _lib.check(lib.nf_device_synchronize())
        walls = []
        for b in range(10):
            t0 = time.perf_counter()
            for i in range(steps):
                call(i)
            _lib.check(lib.nf_device_synchronize())
            walls.append(time.perf_counter() - t0)
        walls.sort()
        wall = walls[len(walls) // 2]
        line("f3", f"density generator {dens}, 2^24 rows into HBM", "density_sample_kernel", rows, wall, wall * 1e3,
             walls, "rows/s", hbm(rows, 8, wall * 1e3, "wall-clock (the generator runs on the default stream); bound by "
                                  "vector issue (Philox4x32-10 + Box-Muller per row), 8 B written per row"))
        out.free()
    return res


if __name__ == "__main__":
    main()
