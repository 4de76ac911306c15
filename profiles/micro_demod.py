"""Scratch: the fused demodulator (iqa_demodulate: reduce / carry / apply / finish) alone on the bench shape, with and
without the per-chunk statistics, and the resampler alone."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import os
import iq_to_audio_amd as A
from iq_to_audio_amd import _native as NATIVE
if os.environ.get("IQA_LIB"):  # an experiment build of the library
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
from iq_to_audio_amd import _dev as D, dsp_plan as P
from iq_to_audio_amd.processing import ChannelDemod, Resampler48k

n, d, fs = 5_769_231, 104, 10e6
fs_ch = fs / d
z = torch.randn(n, dtype=torch.complex64, device="cuda") * 0.5
audio = D.empty(n, "float32")
starts = P.chunk_output_starts(4194304, d, 0, 600_000_000)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mode in ("nfm", "am", "usb"):
    for label, st in (("144 chunks", starts), ("1 chunk", np.array([0], dtype=np.int64))):
        dem = ChannelDemod(mode, fs_ch, deemph_us=300.0, agc_enabled=True)
        def run():
            dem.reset(); dem.prepare(n, st); dem.process(z, st, audio)
        print(f"demod {mode:4s} {label:10s}: {timeit(run):7.1f} us")
rs = Resampler48k(fs_ch)
print(f"resample: {timeit(lambda: rs.process(audio)):7.1f} us")

# which part of the sink costs time: the same NFM call with the peak and/or the per-chunk sums switched off
from ctypes import byref, c_int64, c_void_p
from iq_to_audio_amd import _native as N
dem = ChannelDemod("nfm", fs_ch, deemph_us=300.0, agc_enabled=True)
dem.prepare(n, starts)
_, _, starts_dev, sumsq, work, scratch = dem._prepared
for pk, ss in ((1, 1), (0, 1), (1, 0), (0, 0)):
    def run():
        N.call("iqa_demodulate", byref(dem.params), N.ptr(z), c_int64(n), N.ptr(dem.state_dev), N.ptr(starts_dev) if ss else c_void_p(0),
               c_int64(len(starts) if ss else 0), N.ptr(dem.peak_dev) if pk else c_void_p(0), N.ptr(sumsq) if ss else c_void_p(0),
               N.ptr(audio), N.ptr(scratch), N.ptr(work), N.stream_ptr())
    print(f"nfm peak={pk} sums={ss}: {timeit(run):7.1f} us")
