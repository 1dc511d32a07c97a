"""Multi-GPU harness: reference images are independent units.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  Image i goes to rank i % world_size, weights are replicated, and the only exchange is
an all-gather of per-image metric rows at the end.  Row counts may differ per rank (images
without valid ground truth are skipped, test.py:223-225), so counts are gathered first and the
payload is padded to the maximum -- a few dozen bytes per image, latency-bound over xGMI.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the process group
    when WORLD_SIZE > 1 -- or, with `force`, also for a single rank (a world-size-1 "nccl" group
    loads RCCL and sends the metric rows through its all-gather on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            # (the rank's device named up front: RCCL binds its communicator to it instead of guessing from the global
            # rank at the first collective -- "can cause a hang if rank to GPU mapping is heterogeneous")
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_indices(num_items: int, rank: int, world: int) -> List[int]:
    """Round-robin partition of image indices."""
    return list(range(rank, num_items, world))


def gather_metric_rows(rows: torch.Tensor, indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather (n_local, C) metric rows and their global image indices from every rank.

    Returns (all_rows, all_indices) sorted by image index, identical on every rank.  Works with
    zero local rows.  With no process group it is the identity; with one -- of any size, a single
    rank included -- the rows travel through the collective.
    """
    rows = rows.reshape(-1, rows.shape[-1]) if rows.numel() else rows.reshape(0, rows.shape[-1])
    if not (dist.is_available() and dist.is_initialized()):
        order = torch.argsort(indices)
        return rows[order], indices[order]
    world = dist.get_world_size()
    dev = rows.device
    count = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    C = rows.shape[1]
    payload = torch.zeros((cap, C + 1), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        payload[: rows.shape[0], :C] = rows.to(torch.float64)
        payload[: rows.shape[0], C] = indices.to(torch.float64)
    gathered = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    parts = [g[:n] for g, n in zip(gathered, counts) if n]
    if not parts:
        return rows.new_zeros((0, C)), indices.new_zeros((0,))
    allp = torch.cat(parts, 0)
    order = torch.argsort(allp[:, C])
    allp = allp[order]
    return allp[:, :C].to(rows.dtype), allp[:, C].to(indices.dtype)


def average_rows(rows: torch.Tensor) -> torch.Tensor:
    """Unweighted mean over rows = compute_avg_metrics (test.py:155-162)."""
    return rows.double().mean(0) if rows.shape[0] else rows.new_zeros(rows.shape[1]).double()
