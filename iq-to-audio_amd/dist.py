"""Multi-GPU sharding of the hot path: one process per GPU, no data-path collective.

The path shards over independent units along two axes:

* **captures** (BASELINE config 4): every rank runs the single-GPU path on its own captures;
* **channels of one capture** (config 5: 40 channels -> 5 per GPU): the capture is needed whole on every rank, so it is
  uploaded once on the source rank and replicated with ONE ``broadcast`` over RCCL/xGMI (24 GB at C5: ~0.2 s against
  120 s of signal), then every rank extracts its share of the channels in one pass (``processing.ChannelBank``).

Either way the only exchange after that is the gather of the finished 48 kHz audio (a few MB per unit) to rank 0 and a
max-reduce of scalars (peak level, elapsed time).  ``backend="nccl"`` is RCCL on ROCm; the CPU tests drive the very same
functions over ``gloo`` with CPU tensors.  ``bench.py --gpus N`` uses :class:`AudioGather`, :func:`fence` and
:func:`max_over_ranks` from here; ``batch.demodulate_sharded`` uses :func:`run_sharded`.
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


def check_launch_env(expected_world: int | None = None, device_count: int | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment, validated BEFORE anything touches a GPU: the job must
    have the number of ranks it was asked for and every rank a device of its own (two ranks on one device make RCCL
    fail inside ``init_process_group`` with "Duplicate GPU detected").  Raises ``SystemExit`` with the fix."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if expected_world is not None and world != expected_world:
        raise SystemExit(f"WORLD_SIZE={world} but {expected_world} ranks were asked for: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {expected_world} --master-addr 127.0.0.1 ...`")
    if not 0 <= rank < world:
        raise SystemExit(f"RANK={rank} outside WORLD_SIZE={world}")
    if device_count is not None and world > 1 and local >= device_count:
        raise SystemExit(f"LOCAL_RANK={local} but this node shows {device_count} GPU(s): one rank per GPU")
    return rank, world, local


def init_from_env(backend: str | None = None, *, high_priority: bool = False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world).
    No-op (0, 1) when WORLD_SIZE is unset or 1.  ``high_priority``: RCCL's kernels on a high-priority stream (the audio
    gather shares the GPU with a channelizer that holds every CU for half a millisecond at a time)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not os.environ.get("IQA_FORCE_DIST"):
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            opts = None
            if high_priority:
                try:
                    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                except Exception:  # pragma: no cover - older builds
                    opts = None
            dist.init_process_group(backend, device_id=torch.device("cuda", local), pg_options=opts)
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def _group_device():
    import torch
    import torch.distributed as dist

    return torch.device("cuda") if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")


def gather_audio(local_audio: list, unit_ids: list[int], n_units: int, dst: int = 0):
    """Gather per-unit 1-D audio tensors (any lengths, one dtype) to ``dst``.

    Returns ``{unit_id: np.ndarray}`` on ``dst`` and ``None`` elsewhere.  Two collectives: an all_gather of the
    (unit id, length) table, then one gather of the rank's audio -- as bytes, RCCL has no int16 -- padded to the longest
    rank payload; the audio is tiny next to the capture, so no bucketing is needed.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return {u: (a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)) for u, a in zip(unit_ids, local_audio)}
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = local_audio[0].device if local_audio else _group_device()
    dtype = local_audio[0].dtype if local_audio else torch.float32
    code = {torch.float32: 0, torch.int16: 1, torch.float64: 2}.get(dtype)
    if code is None:
        raise ValueError(f"gather_audio takes float32 / int16 / float64 audio, not {dtype}")
    per_rank = -(-n_units // world)
    meta = torch.full((per_rank + 1, 2), -1, dtype=torch.int64, device=dev)
    meta[per_rank, 0] = code if local_audio else -1
    for i, (u, a) in enumerate(zip(unit_ids, local_audio)):
        meta[i, 0], meta[i, 1] = u, a.numel() * a.element_size()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    totals = [int(m[:per_rank, 1].clamp(min=0).sum().item()) for m in metas]
    width = max(max(totals), 1)
    payload = torch.zeros(width, dtype=torch.uint8, device=dev)
    pos = 0
    for a in local_audio:
        nb = a.numel() * a.element_size()
        payload[pos : pos + nb] = a.reshape(-1).contiguous().view(torch.uint8)
        pos += nb
    bufs = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None
    np_dtype = {0: np.float32, 1: np.int16, 2: np.float64}
    out = {}
    for r in range(world):
        host = bufs[r].cpu().numpy()
        table = metas[r].cpu().numpy()
        kind = int(table[per_rank, 0])
        pos = 0
        for u, nb in table[:per_rank]:
            if u >= 0:
                out[int(u)] = host[pos : pos + int(nb)].view(np_dtype[kind]).copy()
                pos += int(nb)
    return out


def max_over_ranks(value: float) -> float:
    """Max-reduce of a host scalar (audio peak, elapsed time)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_group_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_capture(capture, numel: int, dtype, *, src: int = 0, device=None):
    """The whole capture on every rank: ``capture`` (1-D tensor of interleaved values) on ``src``, ``None`` elsewhere;
    one ``broadcast`` (RCCL over xGMI for device tensors).  Without a process group the capture is returned as is."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        if capture is None:
            raise ValueError("the source rank must supply the capture")
        return capture
    dev = device if device is not None else _group_device()
    if dist.get_rank() == src:
        if capture is None or capture.numel() != numel:
            raise ValueError("the source rank must supply the whole capture")
        buf = capture.to(dev)
    else:
        buf = torch.empty(numel, dtype=dtype, device=dev)
    dist.broadcast(buf.view(torch.uint8), src=src)  # as bytes: neither RCCL nor gloo has an int16 type
    return buf


class ShardedJob:
    """Independent ``units`` across the ranks of a job, as an object that can be stepped repeatedly (``bench.py --gpus N
    --axis channels`` times its steps; :func:`run_sharded` is one step).

    ``units``: descriptors of the independent pieces of work -- whole captures (BASELINE config 4) or channels of one
    capture (config 5); every rank takes a contiguous share (:func:`shard_units`).
    ``shared``: ``None`` (capture axis: a unit brings its own capture) or ``dict(tensor=..., numel=..., dtype=...,
    device=...)`` -- the capture all units read, present on ``dst`` and replicated to every rank with ONE broadcast here,
    at construction; ``broadcast_s`` is its wall time (max over ranks, after ``sync()`` -- the caller's device
    synchronise -- and a barrier), reported apart from the steady-state step.
    ``step(stage)``: ``stage(my_units, shared) -> [(audio 1-D tensor, peak float), ...]`` -- the single-GPU hot path for
    ALL of a rank's units at once (so that the channels of a capture can share one pass over it), one result per unit,
    in order -- then the gather of every unit's audio on ``dst``.  No collective runs between the broadcast and that
    gather.  ref: the reference's sequential loop over ``--ft`` targets, cli.py:683-710."""

    def __init__(self, units: list, *, shared=None, dst: int = 0, sync=None):
        import time

        import torch.distributed as dist

        self.units, self.dst = list(units), dst
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.common, self.broadcast_s = None, 0.0
        if shared is not None:
            t0 = time.perf_counter()
            self.common = broadcast_capture(shared.get("tensor"), int(shared["numel"]), shared["dtype"], src=dst,
                                            device=shared.get("device"))
            if sync is not None:
                sync()
            if dist.is_initialized():
                dist.barrier()
            self.broadcast_s = max_over_ranks(time.perf_counter() - t0)
        self.mine = shard_units(len(self.units), self.rank, self.world)

    def my_units(self) -> list:
        return [self.units[u] for u in self.mine]

    def step(self, stage):
        """One pass: returns ``({unit index: np.ndarray}, peak)`` on ``dst`` and ``(None, peak)`` elsewhere; ``peak`` is the
        max over all units of all ranks."""
        results = list(stage(self.my_units(), self.common)) if self.mine else []
        if len(results) != len(self.mine):
            raise RuntimeError(f"stage returned {len(results)} results for {len(self.mine)} units")
        audio = [a for a, _ in results]
        peak = max([float(p) for _, p in results], default=0.0)
        gathered = gather_audio(audio, self.mine, len(self.units), dst=self.dst)
        return gathered, max_over_ranks(peak)


def run_sharded(units: list, stage, *, shared=None, dst: int = 0):
    """Process independent ``units`` across the ranks of the job and collect their audio on ``dst``: one
    :class:`ShardedJob` step (see there for ``units``, ``stage`` and ``shared``).
    Returns ``({unit index: np.ndarray}, peak)`` on ``dst`` and ``(None, peak)`` elsewhere."""
    return ShardedJob(units, shared=shared, dst=dst).step(stage)


class AudioGather:
    """The per-capture gather of a batch job: every rank's finished audio of step i goes to ``dst`` while step i + 1
    computes.  At most one gather is in flight (the receive buffers are reused); with ``stream`` (a torch CUDA stream)
    the collective is queued there behind ``after`` events, otherwise (gloo / CPU tensors) it is issued directly.
    ``bench.py --gpus N`` drives this class; ``tests/test_dist_gloo.py`` drives it over gloo."""

    def __init__(self, nbytes: int, *, dst: int = 0, stream=None, device=None):
        import torch
        import torch.distributed as dist

        self.dst, self.stream = dst, stream
        self.active = dist.is_initialized()
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if self.active else (0, 1)
        dev = device if device is not None else _group_device()
        self.recv = ([torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.world)]
                     if (self.active and self.rank == dst) else None)
        self._pending = []
        self.count = 0

    def queue(self, audio, after=None) -> None:
        """Hand one step's audio (any dtype, sent as bytes) to the gather.  ``after``: event(s) behind the audio's last
        producer (stream mode)."""
        import torch
        import torch.distributed as dist

        if not self.active:
            return
        payload = audio.reshape(-1).view(torch.uint8)
        if self.stream is None:
            self._wait()
            self._pending.append(dist.gather(payload, self.recv, dst=self.dst, async_op=True))
        else:
            for ev in ([after] if after is not None and not isinstance(after, (list, tuple)) else (after or [])):
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):  # the gather overlaps the next steps' kernels
                self._wait()
                audio.record_stream(self.stream)
                self._pending.append(dist.gather(payload, self.recv, dst=self.dst, async_op=True))
        self.count += 1

    def _wait(self) -> None:
        while self._pending:
            self._pending.pop().wait()

    def drain(self) -> None:
        """All queued gathers complete (host side; stream mode: enqueued behind them)."""
        import torch

        if self.stream is None:
            self._wait()
        else:
            with torch.cuda.stream(self.stream):
                self._wait()

    def latest(self):
        """The receive buffers (``dst`` only): what the most recent gather delivered, one uint8 tensor per rank."""
        return self.recv


def fence(gather: AudioGather | None = None, *, sync=None) -> None:
    """End of a timed region: outstanding gathers complete, every rank arrives (barrier), ``sync()`` (the device
    synchronise of the caller) runs last."""
    import torch.distributed as dist

    if gather is not None:
        gather.drain()
    if dist.is_initialized():
        dist.barrier()
    if sync is not None:
        sync()
