#!/usr/bin/env python3
"""Host time of ResidentBankRunner.submit / collect at config 3 (is the host ahead of the GPU?).  python profiles/c3_host_submit.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from iq_to_audio_amd import dsp_plan as P  # noqa: E402
from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner  # noqa: E402
from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16  # noqa: E402

fs, secs, uniq = 20e6, 60.0, 2.0
n = int(round(fs * secs))
d, _ = P.choose_decimation(fs, 96_000.0)
host = synthetic_multi_iq_s16(fs, uniq, [(t["freq_offset"], 0.14, t["demod_mode"]) for t in bench.C3_TARGETS]).reshape(-1)
slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32_769)[1], 8192)
buf = bench.padded_resident(host, n, slack)
raw = buf[: 2 * n]
runner = ResidentBankRunner(bench.C3_TARGETS, sample_rate=fs, n_frames=n)
for _ in range(4):
    runner.collect(runner.submit(raw, enclosing=buf, lead_frames=0))
torch.cuda.synchronize()
t_sub, tickets = [], []
t0 = time.perf_counter()
for _ in range(12):
    h0 = time.perf_counter()
    tickets.append(runner.submit(raw, enclosing=buf, lead_frames=0))
    t_sub.append((time.perf_counter() - h0) * 1e3)
t_queued = (time.perf_counter() - t0) * 1e3
for t in tickets:
    runner.collect(t)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) * 1e3
print("host ms per submit:", np.round(t_sub, 2), f"; all 12 queued after {t_queued:.1f} ms, finished after {total:.1f} ms ({total / 12:.2f} per capture)")
