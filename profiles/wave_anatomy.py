"""Where the waves of the ring kernel (13 k steps, D = 208) spend their cycles: diagnostic builds (debug bit 2) count, per
wave, the cycles waiting in front of / at the round barrier and the cycles between barriers.  Usage:
python profiles/wave_anatomy.py   (prints per wave role, averaged over the workgroups; with and without the DMA stream)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import os
import iq_to_audio_amd as A
from iq_to_audio_amd import _native as NATIVE
if os.environ.get("IQA_LIB"):  # an experiment build of the library
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, d, f_off = 20e6, 208, 25e3
n_total = int(fs * 60)
host = synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)
raw = torch.from_numpy(host).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, 12500.0, d)
z = D.empty(-(-n_total // d), "complex64")
roles = {w: f"rt{w & 3} cp{w >> 2}" for w in range(8)}  # (13 k steps: loader waves feed the ring and emit; builds without them --
# IQA_RING_LOADERS_MAX_KS=8 -- have waves 0, 1, 6, 7 issue the DMAs and wave 2 emit)
for dbg, name in ((2, "everything"), (18, "no DMA stream")):
    PR._KERNEL_CACHE.clear()
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
    ch.plan_ahead()
    stamps = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
    prm = ch._kernel.mfma_params[0]
    prm.reserved |= dbg
    prm.debug_stamps = stamps.data_ptr()
    for _ in range(30):  # into the sustained regime
        ch.consumed = 0; ch._hist = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ch.process(raw, out_dev=z, events=(e0, e1), last_block=True)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(256, 8, 4).astype(np.float64)
    rounds = s[:, :, 2].mean()
    print(f"{name}: kernel {e0.elapsed_time(e1):.3f} ms, {rounds:.0f} rounds per workgroup; cycles per round and wave "
          f"(matrix work of a tile: 39 MFMAs x 32 = 1248 pipe cycles, two waves share a SIMD's pipe):")
    for w in range(8):
        wait, work = s[:, w, 0].mean() / rounds, s[:, w, 1].mean() / rounds
        print(f"   wave {w} {roles[w]:24s}: waiting at the barrier {wait:7.0f}   between barriers {work:7.0f}   sum {wait + work:7.0f}")
