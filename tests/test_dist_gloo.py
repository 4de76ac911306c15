"""world_size-2 (and 3) gloo tests of the N>1 path on CPU: unit sharding, the audio gather
(the only exchange step of the path) and the max-reduce.  No GPU needed."""
from __future__ import annotations

import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iq_to_audio_amd import dist as D


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _unit_audio(u: int) -> torch.Tensor:
    n = 1000 + 37 * u  # ragged lengths, as different captures give
    return torch.from_numpy((np.arange(n, dtype=np.float32) * 1e-3 + u).astype(np.float32))


def _worker(rank: int, world: int, port: int, n_units: int, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        mine = D.shard_units(n_units, rank, world)
        audio = [_unit_audio(u) for u in mine]
        got = D.gather_audio(audio, mine, n_units, dst=0)
        peak = D.max_over_ranks(float(rank) + 0.5)
        if rank == 0:
            ok = sorted(got) == list(range(n_units)) and all(np.array_equal(got[u], _unit_audio(u).numpy()) for u in got)
            q.put((ok, peak))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 2), (2, 5), (3, 8)])
def test_gather_audio_gloo(world, n_units):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, peak = q.get(timeout=10)
    assert ok
    assert peak == world - 0.5


def test_shard_units_partition():
    for n, w in ((8, 8), (40, 8), (5, 2), (1, 4), (0, 3), (7, 3)):
        parts = [D.shard_units(n, r, w) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        D.shard_units(4, 2, 2)


def test_gather_without_process_group_is_identity():
    got = D.gather_audio([_unit_audio(3)], [3], 1)
    assert list(got) == [3] and np.array_equal(got[3], _unit_audio(3).numpy())
    assert D.max_over_ranks(1.25) == 1.25
