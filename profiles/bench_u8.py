"""Scratch: the channelizer on a uint8 capture of the C2 shape (60 s @ 10 MS/s, 600 M frames = 1.2 GB): row-staged
ring kernel (matrix cores) against the float32 VALU kernel."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, f_off, n_total, d = 10e6, 25e3, 600_000_000, 104
s16 = synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)
raw = torch.from_numpy(((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, 12500.0, d)
z = D.empty(-(-n_total // d), "complex64")
for use in (True, False, True):
    PR._ChannelKernel.use_mfma = use
    PR._KERNEL_CACHE.clear()
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="u8")
    ts = []
    for it in range(4):
        ch.consumed = 0; ch._hist = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ch.process(raw, out_dev=z, events=(e0, e1), last_block=True); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{ch._kernel.last_kernel:28s}: ms {[round(t, 3) for t in ts]} -> {n_total / np.median(ts[1:]) / 1e6:.0f} GS/s")
