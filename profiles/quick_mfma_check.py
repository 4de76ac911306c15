"""Scratch check used while bringing up the int8-MFMA channelizer: MFMA vs VALU kernel vs oracle, and timing."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, processing as PR
from oracle import cpu_ref as O

def rms(a): return float(np.sqrt(np.mean(np.abs(a.astype(np.complex128))**2)))

for fs, d, bw, nfr in ((2.5e6, 26, 12500., 3_000_000), (10e6, 104, 12500., 12_000_000)):
    f_off = 25e3
    raw = O.synth_capture_s16(fs, nfr / fs, f_off).reshape(-1)
    taps = A.design_channel_filter(fs, bw, d)
    x = D.to_device(raw, "int16")
    outs = {}
    for use in (False, True):
        PR._ChannelKernel.use_mfma = use
        ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
        z = ch.process(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ch2 = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
            z2 = ch2.process(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        outs[use] = z.cpu().numpy()
        same = bool(torch.equal(z, z2))
        print(f"fs={fs/1e6}M D={d} L={len(taps)} mfma={use} has_mfma={ch._kernel.mfma is not None}: {dt*1e3:.3f} ms -> {nfr/dt/1e9:.1f} GS/s; reproducible={same}")
    n_cpu = min(nfr, 2_000_000)
    want = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw[:2*n_cpu], 's16'), O.NcoState(f_off, fs), 1), O.OverlapSaveState(taps, 65536)), O.DecimState(d))
    k = want.size
    print("  valu vs oracle rms", rms(outs[False][:k] - want), "max", np.abs(outs[False][:k] - want).max())
    print("  mfma vs oracle rms", rms(outs[True][:k] - want), "max", np.abs(outs[True][:k] - want).max())
    print("  mfma vs valu  rms", rms(outs[True] - outs[False]), "max", np.abs(outs[True] - outs[False]).max(), "n", outs[True].size)
