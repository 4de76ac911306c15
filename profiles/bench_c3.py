"""Scratch: BASELINE config 3 shape (60 s @ 20 MS/s, five --ft targets nfm/am/usb/lsb/nfm with their own bandwidths, AGC
on) on one resident capture: one ResidentCaptureRunner per target, all five queued per pass over the capture.  Not a
bench.py line (bench.py measures config 2); the numbers go to DESIGN.md section 6."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, dsp_plan as P
from iq_to_audio_amd.batch import ResidentCaptureRunner
from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

fs, secs, uniq = 20e6, 60.0, 2.0
targets = [(25e3, "nfm", 12500.0), (-150e3, "am", 10000.0), (400e3, "usb", 2800.0), (-1.1e6, "lsb", 2800.0), (2.3e6, "nfm", 12500.0)]
n_total = int(fs * secs)
host = synthetic_multi_iq_s16(fs, uniq, [(o, 0.14, m) for o, m, _ in targets]).reshape(-1)
d, fs_ch = P.choose_decimation(fs, 96000.0); chunk = P.tune_chunk_size(fs, 1048576)
slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32769)[1], 8192)
buf = torch.zeros(2 * (n_total + slack), dtype=torch.int16, device="cuda")
buf[: 2 * n_total] = torch.from_numpy(host).cuda().repeat(int(secs / uniq))[: 2 * n_total]
raw = buf[: 2 * n_total]
runners = []
for off, mode, bw in targets:
    taps = A.design_channel_filter(fs, bw, d)
    runners.append((mode, len(taps), ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                                          n_frames=n_total, demod_mode=mode, agc_enabled=True)))
def one_pass():
    ts = [r.submit(raw, enclosing=buf, lead_frames=0, resident=True) for _, _, r in runners]
    return [r.collect(t) for (_, _, r), t in zip(runners, ts)]
for _ in range(int(__import__("os").environ.get("WARM", "10"))): res = one_pass()
torch.cuda.synchronize()
K = int(__import__("os").environ.get("K", "60"))
t0 = time.perf_counter()
for _ in range(K): res = one_pass()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"C3 shape: {len(targets)} targets over one 60 s @ 20 MS/s capture: {dt*1e3:.2f} ms per capture = {n_total/dt/1e6:.0f} MS/s of capture "
      f"({len(targets)*n_total/dt/1e9:.2f} G channel-samples/s); signs {[r['sign'] for r in res]}; taps {[n for _, n, _ in runners]}")
for (mode, ntaps, r), out in zip(runners, res):
    print(f"  {mode:4s} {ntaps:6d} taps  kernel {out['kernel']}  peak {out['demod'].peak:.4f}")
