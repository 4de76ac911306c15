"""Multi-GPU sharding of the hot path: one process per GPU, no data-path collective.

The path shards over independent units: whole captures (BASELINE
config 4) or channels of one capture (config 5).  Every rank runs the single-GPU pipeline on
its own units; the only exchange is the gather of the finished 48 kHz audio (a few MB per
channel) and a max-reduce of the peak level.  ``backend="nccl"`` is RCCL over xGMI on ROCm;
``gloo`` is used by the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np


def shard_units(n_units: int, rank: int, world: int) -> list[int]:
    """Contiguous, balanced assignment of unit indices to ranks (first ``n % world`` ranks get one more)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world).
    No-op (0, 1) when WORLD_SIZE is unset or 1."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def gather_audio(local_audio: list, unit_ids: list[int], n_units: int, dst: int = 0):
    """Gather per-unit 1-D float32 audio tensors (any lengths) to ``dst``.

    Returns ``{unit_id: np.ndarray}`` on ``dst`` and ``None`` elsewhere.  Two collectives: an
    all_gather of the (unit id, length) table, then one gather of the rank's audio padded to the
    longest rank payload -- the audio is tiny next to the capture, so no bucketing is needed.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return {u: (a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)) for u, a in zip(unit_ids, local_audio)}
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = local_audio[0].device if local_audio else torch.device("cuda" if dist.get_backend() == "nccl" else "cpu")
    per_rank = -(-n_units // world)
    meta = torch.full((per_rank, 2), -1, dtype=torch.int64, device=dev)
    for i, (u, a) in enumerate(zip(unit_ids, local_audio)):
        meta[i, 0], meta[i, 1] = u, a.numel()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    totals = [int(m[:, 1].clamp(min=0).sum().item()) for m in metas]
    width = max(max(totals), 1)
    payload = torch.zeros(width, dtype=torch.float32, device=dev)
    pos = 0
    for a in local_audio:
        payload[pos : pos + a.numel()] = a.reshape(-1).to(torch.float32)
        pos += a.numel()
    bufs = [torch.empty(width, dtype=torch.float32, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None
    out = {}
    for r in range(world):
        host = bufs[r].cpu().numpy()
        pos = 0
        for u, n in metas[r].cpu().numpy():
            if u >= 0:
                out[int(u)] = host[pos : pos + int(n)].copy()
                pos += int(n)
    return out


def max_over_ranks(value: float) -> float:
    """Max-reduce of a host scalar (audio peak, elapsed time)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return float(value)
    dev = torch.device("cuda") if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
