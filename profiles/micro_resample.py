"""Scratch: the 48 kHz resampler alone on the bench shape (config 2: 5 769 231 samples at 96 153.8 Hz -> PCM16), against
the oracle on a short stretch, for a few launch sizes (IQA_RS_TARGET_WAVES is read once per process: run once per value)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _native as NATIVE
if os.environ.get("IQA_LIB"):  # an experiment build of the library
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
from iq_to_audio_amd import _dev as D
from iq_to_audio_amd.processing import Resampler48k

n, fs_ch = 5_769_231, 10e6 / 104
rs = Resampler48k(fs_ch)
g = torch.Generator(device="cuda").manual_seed(5)
audio = (torch.rand(n, device="cuda", generator=g) - 0.5) * 1.6
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for want in ("pcm16", "f32", "both"):
    print(f"lib={os.path.basename(os.environ.get('IQA_LIB', 'default'))} want={want}: {timeit(lambda: rs.process(audio, want=want)):7.1f} us", flush=True)
if os.environ.get("IQA_LIB"):
    sys.exit(0)
# parity on a stretch that includes both stream edges
from oracle import cpu_ref as O
for m in (200_000, 77):
    x = audio[:m].contiguous()
    y, pcm = rs.process(x, want="both")
    ref = O.resample_48k(x.cpu().numpy(), fs_ch)
    if ref is not None:
        print("n_in", m, "max |gpu - oracle|", float(np.abs(y.cpu().numpy() - ref).max()), "pcm equal to rounding of y:",
              bool((pcm.cpu().numpy() == np.clip(np.rint(y.cpu().numpy().astype(np.float64) * 32768), -32768, 32767).astype(np.int16)).all()))
