#!/usr/bin/env python3
"""Config 1 replayed as hipGraphs on 1, 2, 3, 4 streams (ResidentCaptureRunner(graph_streams=...)): ms per capture.
python profiles/c1_graph_streams.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
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
ref = None
for rep in range(2):
    for gs in (1, 2, 3, 4):
        runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=P.tune_chunk_size(fs, 1_048_576),
                                       n_frames=n, slots=8, graph_streams=gs)
        for _ in range(100):
            r = runner.collect(runner.submit_captured(raw, enclosing=buf, lead_frames=0))
        torch.cuda.synchronize()
        K = 2000
        t0 = time.perf_counter()
        ts = [runner.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(K)]
        res = [runner.collect(t) for t in ts[-8:]]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        pcm = res[-1]["pcm_host"].numpy().copy()
        if ref is None:
            ref = pcm
        print(f"graph_streams={gs}: {dt * 1e6:6.1f} us per capture = {n / dt / 1e9:6.1f} GS/s; PCM16 identical to one stream: {np.array_equal(pcm, ref)}; redone {runner.redone}", flush=True)
        del runner
