#!/usr/bin/env python3
"""Two-stream replay of config 1 in a FRESH process: per batch of 200 captures, for two runners made one after the other.
(bench.sub_bench_c1's first call in a process measured 0.17-0.20 ms with two streams, every later call 0.053.)
python profiles/c1_streams_first_use.py [graph_streams] [prime]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import iq_to_audio_amd as A  # noqa: E402
from iq_to_audio_amd import _dev as D  # noqa: E402
from iq_to_audio_amd import dsp_plan as P  # noqa: E402
from iq_to_audio_amd.batch import ResidentCaptureRunner  # noqa: E402
from iq_to_audio_amd.benchmark import synthetic_iq_s16  # noqa: E402

gs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prime = "prime" in sys.argv[2:]
like_bench = "bench" in sys.argv[2:]  # what bench.sub_bench_c1 does in front of its two-stream runner
direct_only = "direct" in sys.argv[2:]
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
if prime:  # touch a handful of streams first
    ss = [torch.cuda.Stream() for _ in range(8)]
    for s in ss:
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
if like_bench or direct_only:
    r1 = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=P.tune_chunk_size(fs, 1_048_576),
                               n_frames=n, slots=8)
    for t in [r1.submit(raw, enclosing=buf, lead_frames=0, resident=True) for _ in range(300)]:
        r1.collect(t)
    if like_bench:
        for t in [r1.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(300)]:
            r1.collect(t)
    torch.cuda.synchronize()
for which in range(3):
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=P.tune_chunk_size(fs, 1_048_576),
                                   n_frames=n, slots=8, graph_streams=gs)
    for _ in range(100):
        runner.collect(runner.submit_captured(raw, enclosing=buf, lead_frames=0))
    torch.cuda.synchronize()
    out = []
    for batch in range(12):
        t0 = time.perf_counter()
        ts = [runner.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(200)]
        for t in ts:
            runner.collect(t)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"runner {which} (graph_streams={gs}, prime={prime}): us per capture by batch of 200:", " ".join(f"{v:.0f}" for v in out), flush=True)
    del runner
