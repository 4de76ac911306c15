"""world_size-2 (and 3) gloo tests of the N>1 path on CPU.  They drive the SAME functions the GPU job executes --
``dist.run_sharded`` on both axes (captures; channels of one broadcast capture), ``dist.AudioGather`` + ``dist.fence`` +
``dist.max_over_ranks`` (what ``bench.py --gpus N`` calls per step) and ``dist.check_launch_env`` -- with CPU tensors and
a stub in place of the single-GPU stage.  No GPU needed."""
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


def _unit_audio(u: int, dtype=torch.float32) -> torch.Tensor:
    n = 1000 + 37 * u  # ragged lengths, as different captures give
    x = np.arange(n, dtype=np.float64) * 1e-3 + u
    return torch.from_numpy(x.astype(np.float32)) if dtype == torch.float32 else torch.from_numpy((x * 100).astype(np.int16))


def _spawn(worker, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return q.get(timeout=10)


def _gather_worker(rank, world, port, q, n_units, use_int16):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        dtype = torch.int16 if use_int16 else torch.float32
        mine = D.shard_units(n_units, rank, world)
        got = D.gather_audio([_unit_audio(u, dtype) for u in mine], mine, n_units, dst=0)
        peak = D.max_over_ranks(float(rank) + 0.5)
        if rank == 0:
            ok = sorted(got) == list(range(n_units)) and all(
                got[u].dtype == _unit_audio(u, dtype).numpy().dtype and np.array_equal(got[u], _unit_audio(u, dtype).numpy()) for u in got)
            q.put((ok, peak))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units,use_int16", [(2, 2, False), (2, 5, True), (3, 8, False), (3, 2, True)])
def test_gather_audio_gloo(world, n_units, use_int16):
    ok, peak = _spawn(_gather_worker, world, n_units, use_int16)
    assert ok
    assert peak == world - 0.5


# ---- run_sharded: the function batch.demodulate_sharded calls, with a stub stage -----------------------------------


def _stub_stage(mine, shared):
    """Stands in for the single-GPU hot path: 'audio' of a unit = a function of the unit and of the shared capture."""
    out = []
    for unit in mine:
        if shared is None:  # capture axis: the unit brings its own capture (here: a seed)
            cap = torch.arange(64, dtype=torch.int16) * int(unit["seed"])
        else:  # channel axis: every unit reads the one capture every rank received
            cap = shared
        audio = (cap[: 40 + unit["k"]].to(torch.int16) + unit["k"]).contiguous()
        out.append((audio, float(unit["k"]) + 0.25))
    return out


def _sharded_worker(rank, world, port, q, axis, n_units):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        units = [dict(k=k, seed=3 + k) for k in range(n_units)]
        capture = torch.arange(200, dtype=torch.int16) * 7
        if axis == "channels":
            shared = dict(tensor=capture if rank == 0 else None, numel=200, dtype=torch.int16, device=torch.device("cpu"))
            got, peak = D.run_sharded(units, _stub_stage, shared=shared)
        else:
            got, peak = D.run_sharded(units, _stub_stage)
        if rank == 0:
            want = _stub_stage(units, capture if axis == "channels" else None)
            ok = sorted(got) == list(range(n_units)) and all(np.array_equal(got[i], want[i][0].numpy()) for i in range(n_units))
            q.put((ok, peak))
        else:
            assert got is None and peak == n_units - 1 + 0.25
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("axis", ["channels", "captures"])
@pytest.mark.parametrize("world,n_units", [(2, 5), (3, 8), (2, 1)])
def test_run_sharded_both_axes_gloo(axis, world, n_units):
    """BASELINE config 5's axis (one capture broadcast from rank 0, the channels sharded) and config 4's (independent
    captures): every unit's audio arrives on rank 0 exactly as the single-process path produces it."""
    ok, peak = _spawn(_sharded_worker, world, axis, n_units)
    assert ok and peak == n_units - 1 + 0.25


def _job_worker(rank, world, port, q, n_units, steps):
    """bench.py --axis channels: ONE broadcast at construction, then repeated steps (stage + gather), a fence at the end."""
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        units = [dict(k=k, seed=3 + k) for k in range(n_units)]
        capture = torch.arange(200, dtype=torch.int16) * 7
        calls = []
        job = D.ShardedJob(units, shared=dict(tensor=capture if rank == 0 else None, numel=200, dtype=torch.int16, device=torch.device("cpu")),
                           sync=lambda: calls.append("sync"))
        assert calls == ["sync"] and job.broadcast_s >= 0.0
        assert torch.equal(job.common, capture)  # every rank holds the whole capture after the one broadcast
        assert [u["k"] for u in job.my_units()] == D.shard_units(n_units, rank, world)
        outs = []
        for i in range(steps):
            def stage(mine, shared, i=i):
                return [(a + i, p) for a, p in _stub_stage(mine, shared)]
            outs.append(job.step(stage))
        D.fence(None)
        if rank == 0:
            want = _stub_stage(units, capture)
            ok = all(sorted(got) == list(range(n_units)) and all(np.array_equal(got[u], (want[u][0] + i).numpy()) for u in got)
                     for i, (got, _) in enumerate(outs))
            q.put((ok, outs[-1][1]))
        else:
            assert all(got is None for got, _ in outs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 5), (3, 40)])
def test_sharded_job_steps_gloo(world, n_units):
    """dist.ShardedJob -- what ``bench.py --gpus N --axis channels`` constructs and steps: the capture is broadcast once,
    every step gathers every unit's audio on rank 0 (40 units over 3 ranks: config 5's channel count on an odd world)."""
    ok, peak = _spawn(_job_worker, world, n_units, 3)
    assert ok and peak == n_units - 1 + 0.25


def test_run_sharded_without_process_group():
    units = [dict(k=k, seed=3 + k) for k in range(3)]
    got, peak = D.run_sharded(units, _stub_stage)
    assert sorted(got) == [0, 1, 2] and peak == 2.25
    cap = torch.arange(200, dtype=torch.int16)
    got, _ = D.run_sharded(units, _stub_stage, shared=dict(tensor=cap, numel=200, dtype=torch.int16))
    assert np.array_equal(got[2], (cap[:42] + 2).numpy())
    with pytest.raises(RuntimeError):
        D.run_sharded(units, lambda mine, shared: [])  # a stage must answer for every unit


# ---- the per-step gather of bench.py --------------------------------------------------------------------------------


def _bench_step_worker(rank, world, port, q, steps):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n48 = 300
        gather = D.AudioGather(2 * n48, dst=0, stream=None, device=torch.device("cpu"))
        seen = []
        for i in range(steps):  # bench.py's loop: hand the PREVIOUS step's audio to the gather, then compute the next
            pcm = (torch.arange(n48, dtype=torch.int16) + 1000 * rank + i)
            gather.queue(pcm)
            if rank == 0:
                gather.drain()
                seen.append([buf.view(torch.int16).clone() for buf in gather.latest()])
        D.fence(gather)
        elapsed = D.max_over_ranks(0.001 * (rank + 1))
        if rank == 0:
            ok = all(torch.equal(seen[i][r], torch.arange(n48, dtype=torch.int16) + 1000 * r + i)
                     for i in range(steps) for r in range(world))
            q.put((ok and gather.count == steps, elapsed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_audio_gather_and_fence_gloo(world):
    ok, elapsed = _spawn(_bench_step_worker, world, 4)
    assert ok and abs(elapsed - 0.001 * world) < 1e-12


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
    g = D.AudioGather(16)  # no process group: a no-op
    g.queue(torch.zeros(8, dtype=torch.int16))
    D.fence(g)
    assert g.count == 0 and g.latest() is None


def test_launch_env_is_checked_before_any_gpu_call(monkeypatch):
    """What killed round 1's only 2-rank attempt: two ranks on one device.  The environment is validated up front."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert D.check_launch_env(2, device_count=8) == (1, 2, 1)
    with pytest.raises(SystemExit, match="one rank per GPU"):
        D.check_launch_env(2, device_count=1)
    with pytest.raises(SystemExit, match="ranks were asked for"):
        D.check_launch_env(4, device_count=8)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert D.check_launch_env(1, device_count=1) == (0, 1, 0)
    with pytest.raises(SystemExit):
        D.check_launch_env(2, device_count=1)
