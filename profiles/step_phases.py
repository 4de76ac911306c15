"""Scratch: wall time of each phase of a bench step, with a device sync after every phase."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, dsp_plan as P
from iq_to_audio_amd.processing import ChannelDemod, MixSignProbe, Resampler48k
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, f_off, bw = 10e6, 25e3, 12500.0
n_total = 600_000_000
d, fs_ch = P.choose_decimation(fs, 96000.0); chunk = P.tune_chunk_size(fs, 1048576)
taps = A.design_channel_filter(fs, bw, d)
host = synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)
raw = torch.from_numpy(host).to("cuda").repeat(60)[: 2 * n_total].contiguous()
n_dec = -(-n_total // d); starts = P.chunk_output_starts(chunk, d, 0, n_total)
rs = Resampler48k(fs_ch); n48 = rs.plan.n_out(n_dec)
pcm_host = torch.empty(n48, dtype=torch.int16).pin_memory()
z = D.empty(n_dec, "complex64"); audio = D.empty(n_dec, "float32")
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return time.perf_counter()
for it in range(8):
    t = time.perf_counter()
    probe = MixSignProbe(raw[: 2 * chunk], fs, f_off, taps, d, fmt="s16"); t = lap("probe launch+run", t)
    chan = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="s16"); chan.plan_ahead(); t = lap("plan+upload", t)
    dem = ChannelDemod("nfm", fs_ch, deemph_us=300.0, agc_enabled=True); dem.prepare(n_dec, starts); t = lap("demod prepare", t)
    sign = probe.result(); t = lap("probe result", t)
    chan.process(raw, out_dev=z); t = lap("channelize", t)
    dem.process(z, starts, audio); t = lap("demod", t)
    y48 = rs.process(audio); t = lap("resample", t)
    pcm = rs.to_pcm16(y48); pcm_host.copy_(pcm, non_blocking=True); t = lap("pcm+d2h", t)
for k, v in T.items():
    print(f"{k:22s} median {np.median(v[2:]):7.3f} ms   all {[round(x,3) for x in v]}")
