"""Scratch: the same step as bench.py but with the capture starting in PINNED HOST memory every time (H2D of the
2.4 GB of int16 frames inside the timed region, double-buffered against the previous capture's kernels).  Reported in
DESIGN.md section 6; never bench.py's `value`."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, dsp_plan as P
from iq_to_audio_amd.benchmark import synthetic_iq_s16
from iq_to_audio_amd.batch import ResidentCaptureRunner

fs, f_off, n_total = 10e6, 25e3, 600_000_000
d, fs_ch = P.choose_decimation(fs, 96000.0); chunk = P.tune_chunk_size(fs, 1048576)
taps = A.design_channel_filter(fs, 12500.0, d)
host = torch.from_numpy(np.tile(synthetic_iq_s16(fs, 1.0, f_off).reshape(-1), 60)[: 2 * n_total]).pin_memory()
lead, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
bufs = [torch.zeros(2 * (n_total + slack), dtype=torch.int16, device="cuda") for _ in range(2)]
runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk, n_frames=n_total)
up = torch.cuda.Stream()
def step(i):
    buf = bufs[i % 2]
    with torch.cuda.stream(up):          # upload of capture i beside the kernels of capture i-1
        buf[: 2 * n_total].copy_(host, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    torch.cuda.current_stream().wait_event(ev)
    return runner.submit(buf[: 2 * n_total], enclosing=buf, lead_frames=0)
ts = [step(i) for i in range(3)]
for t in ts: runner.collect(t)
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
ts = [step(i) for i in range(K)]
for t in ts: runner.collect(t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"host-resident capture -> 48 kHz PCM16 on the host: {dt*1e3:.1f} ms per 60 s capture = {n_total/dt/1e6:.0f} MS/s "
      f"({2*2*n_total/dt/1e9:.1f} GB/s over PCIe)")
