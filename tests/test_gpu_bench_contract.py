"""The driver's contract for bench.py, on small workloads: ONE JSON line on stdout with the agreed keys, for the capture
axis (the default command the driver runs at N = 1, 2, 4, 8) and for the channel axis (BASELINE config 5's partitioning),
including the single-rank rehearsal of the N > 1 code path (process group, RCCL gather, barrier, max over ranks)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _run(args, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]  # exactly one line: the JSON
    return json.loads(lines[0])


@pytest.mark.parametrize("force_dist", [False, True])
def test_capture_axis_line(force_dist):
    d = _run(["--gpus", "1", "--steps", "4", "--warmup", "2", "--settle", "0", "--seconds", "3", "--unique-seconds", "1", "--no-extras",
              "--cpu-seconds", "0.2"], {"IQA_FORCE_DIST": "1"} if force_dist else None)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "MS/s" and d["value"] > 1000.0 and abs(d["value"] - 30e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and r["kernel"].startswith("k_channelize_mfma_s16")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    if force_dist:
        assert d["config"]["gathers"] == 4 + 2  # every capture's PCM16 went through the RCCL gather
    else:
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "MS/s" and cb["value"] > 1.0
        assert d["parity"]["rms_err_vs_oracle_fs_channel"] < 1e-4


@pytest.mark.parametrize("force_dist", [False, True])
def test_channel_axis_line(force_dist):
    d = _run(["--gpus", "1", "--axis", "channels", "--seconds", "1.5", "--channels", "6", "--steps", "2", "--warmup", "1"],
             {"IQA_FORCE_DIST": "1"} if force_dist else None)
    assert KEYS <= set(d) and d["scaling"] == "strong" and d["n_gpus"] == 1 and d["steps"] == 2
    c = d["config"]
    assert c["axis"] == "channels" and c["channels"] == 6 and c["channels_per_gpu"] == 6 and c["audio_units_gathered"] == 6
    assert c["frames"] == 75_000_000 and c["broadcast_s"] >= 0.0
    assert d["value"] > 100.0 and abs(d["value"] - c["frames"] / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
