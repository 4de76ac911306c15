"""Scratch: A/B of ring-kernel variants on the C2 launch, interleaved launch by launch (the chip's clock drifts with
load and temperature, so back-to-back blocks of one variant are not comparable).  FLAGS_A / FLAGS_B: extra bits for
iqa_mfma_params.reserved (256 = linear LDS image instead of the swizzled one)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, d, f_off, n_total = 10e6, 104, 25e3, 600_000_000
raw = torch.from_numpy(synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, 12500.0, d)
z = D.empty(-(-n_total // d), "complex64")
flags = [int(os.environ.get("FLAGS_A", "0")), int(os.environ.get("FLAGS_B", "256"))]
ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
ch.plan_ahead()
base = ch._kernel.mfma_params[0].reserved
spacer = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
times = {0: [], 1: []}
for it in range(int(os.environ.get("REPS", "60"))):
    for v in (0, 1):
        ch._kernel.mfma_params[0].reserved = base | flags[v]
        ch.consumed = 0; ch._hist = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ch.process(raw, out_dev=z, events=(e0, e1), last_block=True)
        spacer.zero_()  # ~0.2 ms of light work between launches, as in a bench step
        torch.cuda.synchronize()
        if it >= 5:
            times[v].append(e0.elapsed_time(e1))
for v in (0, 1):
    t = np.array(times[v])
    print(f"flags +{flags[v]:3d}: median {np.median(t):.4f} ms  mean {t.mean():.4f}  min {t.min():.4f}  (n={t.size})")
