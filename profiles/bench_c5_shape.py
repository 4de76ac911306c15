"""Scratch: the channelizer on BASELINE config 5's shape (50 MS/s, D = 521, 32001 taps, 33 k steps): per-lane kernel
(three k-step-range passes, each reading the whole capture) against the row-staged ring kernel (three passes of 11,
each fetching a third of every row).  12 s of signal = 600 M frames = 2.4 GB resident."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, f_off, n_total = 50e6, 25e3, 600_000_000
d = 521
raw = torch.from_numpy(synthetic_iq_s16(fs, 0.2, f_off).reshape(-1)).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, 12500.0, d)
z = D.empty(-(-n_total // d), "complex64")
outs = {}
for variant in ("plain", "ring", "plain", "ring"):
    PR._ChannelKernel.mfma_variant = variant
    PR._KERNEL_CACHE.clear()
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
    ts = []
    for it in range(5):
        ch.consumed = 0; ch._hist = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ch.process(raw, out_dev=z, events=(e0, e1), last_block=True); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    outs[variant] = z.clone()
    print(f"{variant:6s} {ch._kernel.last_kernel:28s} passes {len(ch._kernel.mfma.passes)}: ms {[round(t, 3) for t in ts]}  -> {n_total / np.median(ts[1:]) / 1e6:.0f} GS/s")
print("max |ring - plain| =", float((outs["ring"] - outs["plain"]).abs().max()))
