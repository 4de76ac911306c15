#!/usr/bin/env python3
"""Config 1 replayed as hipGraphs: where the host's ~85 us per replay go.  Times, per capture: graph.replay() alone,
event record, collect() of a finished capture, the whole submit_captured(); and the GPU side alone (replays queued
back to back without collects in between, by events).  python profiles/c1_host_anatomy.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import iq_to_audio_amd as A  # noqa: E402
from iq_to_audio_amd import _dev as D  # noqa: E402
from iq_to_audio_amd import dsp_plan as P  # noqa: E402
from iq_to_audio_amd.batch import ResidentCaptureRunner  # noqa: E402
from iq_to_audio_amd.benchmark import synthetic_iq_s16  # noqa: E402

fs, secs, f_off = 2.5e6, 5.0, 25e3
n = int(fs * secs)
d, fs_ch = P.choose_decimation(fs, 96_000.0)
taps = A.design_channel_filter(fs, 12_500.0, d)
host = synthetic_iq_s16(fs, secs, f_off).reshape(-1)
_, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
buf = torch.zeros(2 * (n + slack), dtype=torch.int16, device=D.device())
buf[: 2 * n] = torch.from_numpy(host).to(D.device())
raw = buf[: 2 * n]
torch.cuda.synchronize()
runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=P.tune_chunk_size(fs, 1_048_576),
                               n_frames=n, slots=8)
for _ in range(64):
    runner.collect(runner.submit_captured(raw, enclosing=buf, lead_frames=0))
torch.cuda.synchronize()
graphs = [e["graph"] for e in runner._graphs.values()]
print("graphs captured:", len(graphs))
K = 2000
# (1) graph.replay() alone, round-robin over the slots' graphs (GPU may lag behind: launches only)
t0 = time.perf_counter()
for i in range(K):
    graphs[i % len(graphs)].replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph.replay() alone: {1e6 * (t1 - t0) / K:.1f} us per call on the host; with the GPU drained at the end {1e6 * (t2 - t0) / K:.1f} us per capture")
# (2) GPU side alone: events around K replays
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(K):
    graphs[i % len(graphs)].replay()
e1.record()
torch.cuda.synchronize()
print(f"GPU time per replayed capture (events around {K} replays): {1e3 * e0.elapsed_time(e1) / K:.1f} us")
# (3) an event record
ev = torch.cuda.Event()
t0 = time.perf_counter()
for i in range(K):
    ev.record()
t1 = time.perf_counter()
print(f"event record: {1e6 * (t1 - t0) / K:.1f} us")
torch.cuda.synchronize()
# (4) the runner's own path
ts = []
t0 = time.perf_counter()
for i in range(K):
    ts.append(runner.submit_captured(raw, enclosing=buf, lead_frames=0))
t1 = time.perf_counter()
for t in ts[-8:]:
    runner.collect(t)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"submit_captured (collect of the capture 8 back inside): {1e6 * (t1 - t0) / K:.1f} us per call; per capture incl. drain {1e6 * (t2 - t0) / K:.1f} us")
# (5) collect of an already finished capture
tk = [runner.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(8)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in tk:
    runner.collect(t)
t1 = time.perf_counter()
print(f"collect() of a finished capture: {1e6 * (t1 - t0) / 8:.1f} us")
