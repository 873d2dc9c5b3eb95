"""Data-parallel request sharding: one process per GPU, full replicas, no collective on
the accept path.  Mirrors the reference's `distributed.py:18-83` helpers
and its `range(rank, n, world)` prompt split (benchmark.py:445); the only collectives
are the end-of-run gather of per-request results and the bench's timing scalars.  On
PyTorch-ROCm the "nccl" backend is RCCL; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from typing import Any, Optional

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> None:
    """env:// rendezvous iff RANK is set (distributed.py:18-22); otherwise single process."""
    if "RANK" not in os.environ or dist.is_initialized():
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend, init_method="env://")


def destroy() -> None:
    if dist.is_initialized():
        try:
            dist.barrier()
        except Exception:
            pass
        dist.destroy_process_group()


def size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank() -> int:
    return int(os.environ.get("RANK", "0"))


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def is_main() -> bool:
    return rank() == 0


def shard_indices(n: int, r: Optional[int] = None, w: Optional[int] = None) -> range:
    """Requests of this rank: r, r+W, r+2W, ... (benchmark.py:445)."""
    return range(rank() if r is None else r, n, size() if w is None else w)


def gather(obj: Any, dst: int = 0):
    """gather_object to `dst` (distributed.py:66-75); identity without a process group."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def merge_sharded(per_rank: list, n: int) -> list:
    """Undo the round-robin split: per_rank[r] lists the results of requests r, r+W, ..."""
    w = len(per_rank)
    out = [None] * n
    for r, items in enumerate(per_rank):
        for j, item in enumerate(items):
            out[r + j * w] = item
    return out


def reduce_timing(seconds: float, units: float, device=None, group=None):
    """(max seconds over ranks, total units): the bench's whole-job throughput inputs.  group: the process group to
    reduce over (bench.py: an RCCL group created only after the timed regions; default: the default group)."""
    if not dist.is_initialized():
        return seconds, units
    if dist.get_backend(group) == "gloo":
        device = None   # host tensors (CPU tests, one-GPU rehearsal, bench.py's default group)
    t = torch.tensor([seconds, units], dtype=torch.float64, device=device)
    mx, sm = t.clone(), t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=group)
    return float(mx[0]), float(sm[1])
