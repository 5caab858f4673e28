"""Data-parallel evaluation: rows of the sample matrix are split over ranks, the
only exchange is the sum of ONE float64.

The reference is single-process (`-jnp.mean(log_prob.apply(params, batch))`,
src/real_nvp_toy_densities.py:143,150); this is the multi-GPU form of that mean:

    NLL = -(sum over ranks of sum_local(log_prob)) / rows_global

On GPUs the all-reduce runs inside libnf_hip (RCCL over xGMI, on the flow's
stream).  `ShardedNLL` takes any `allreduce_sum(float) -> float`, which is how the
world_size-2 gloo tests drive the same host logic on CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

from . import _lib
from ._lib import check


def shard_rows(rows_global: int, world: int, rank: int, align: int = 2) -> Tuple[int, int]:
    """Contiguous block [begin, end) of rank `rank`.  Blocks differ by at most
    `align` rows and begin on a multiple of `align` (keeps 16-byte row alignment
    for [rows, 2] float32 matrices); the last rank takes the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    if rows_global < 0:
        raise ValueError("rows_global must be >= 0")
    units = rows_global // align
    base, extra = divmod(units, world)
    begin_u = rank * base + min(rank, extra)
    end_u = begin_u + base + (1 if rank < extra else 0)
    begin, end = begin_u * align, end_u * align
    if rank == world - 1:
        end = rows_global
    return begin, end


class RanksDisagree(RuntimeError):
    """Raised on EVERY rank when at least one rank reported a failure (or a different value)."""


def agree_all(dist, ok: bool, what: str, detail: str = "") -> None:
    """All ranks call this at the same point with their own verdict; if any rank says `ok=False` every rank
    raises RanksDisagree (naming the failing ranks and their messages) -- nobody walks on into a collective
    the failed rank will never join.  `dist` is a torch.distributed-like module on a CPU (gloo) group."""
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, (bool(ok), str(detail)))
    bad = [(r, msg) for r, (good, msg) in enumerate(box) if not good]
    if bad:
        raise RanksDisagree(f"{what}: failed on rank(s) " + "; ".join(f"{r} ({msg or 'no detail'})" for r, msg in bad))


def agree_equal(dist, value, what: str) -> None:
    """All ranks call this with a value that must be the same everywhere (e.g. CouplingFlow.collective_count());
    every rank raises RanksDisagree when it is not."""
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, value)
    if any(v != box[0] for v in box):
        raise RanksDisagree(f"{what}: ranks disagree: " + ", ".join(f"rank {r}: {v}" for r, v in enumerate(box)))


class RcclComm:
    """One RCCL communicator (one rank per GPU)."""

    def __init__(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != _lib.NF_COMM_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self.world, self.rank = world, rank
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, _lib.NF_COMM_ID_BYTES)
        check(_lib.load().nf_comm_init_rank(C.byref(self._h), world, rank, buf))

    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.NF_COMM_ID_BYTES)
        check(_lib.load().nf_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def preflight() -> None:
        """nf_comm_preflight: a usable device is current and librccl loads.  Raises NFError otherwise."""
        check(_lib.load().nf_comm_preflight())

    @classmethod
    def from_torch_distributed(cls, dist) -> "RcclComm":
        """Rank 0 makes the id; the (CPU) process group carries it to the others.

        ncclCommInitRank blocks until every rank has entered it, so everything that can fail on ONE rank is
        done and agreed on (over the CPU group) BEFORE any rank enters it: the preflight (device + librccl)
        and the making of the id.  After the init the outcome is agreed on again, so that a rank whose init
        returned an error does not leave the others to find out at their first all-reduce.  On any failure
        every rank raises RanksDisagree."""
        rank, world = dist.get_rank(), dist.get_world_size()
        try:
            cls.preflight()
            ok, detail = True, ""
        except Exception as exc:  # reported to every rank below
            ok, detail = False, f"{type(exc).__name__}: {exc}"
        agree_all(dist, ok, "RCCL preflight (device + librccl)", detail)
        uid, ok, detail = None, True, ""
        if rank == 0:
            try:
                uid = cls.new_unique_id()
            except Exception as exc:
                ok, detail = False, f"{type(exc).__name__}: {exc}"
        agree_all(dist, ok, "ncclGetUniqueId", detail)
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        comm = None
        try:
            comm = cls(world, rank, box[0])
            ok, detail = True, ""
        except Exception as exc:
            ok, detail = False, f"{type(exc).__name__}: {exc}"
        try:
            agree_all(dist, ok, "ncclCommInitRank", detail)
        except RanksDisagree:
            if comm is not None:
                comm.close()
            raise
        return comm

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().nf_comm_destroy(self._h)
            self._h = C.c_void_p()


class ShardedNLL:
    """Host logic of the sharded mean, independent of who computes the local sum.

    local_sum_fn(begin, end) -> float64 sum of log_prob over this rank's rows;
    allreduce_sum(v) -> sum of v over all ranks.
    """

    def __init__(self, world: int, rank: int, allreduce_sum: Callable[[float], float]):
        self.world, self.rank, self.allreduce_sum = world, rank, allreduce_sum

    def nll(self, rows_global: int, local_sum_fn: Callable[[int, int], float]) -> float:
        if rows_global < 1:
            raise ValueError("NLL of an empty batch is undefined")
        begin, end = shard_rows(rows_global, self.world, self.rank)
        local = float(local_sum_fn(begin, end)) if end > begin else 0.0
        return -self.allreduce_sum(local) / rows_global
