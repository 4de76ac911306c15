"""Scratch micro-benchmark: which part of the writer/sink costs time (peak atomic, sumsq atomics, stores)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from ctypes import c_int64, c_void_p
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, _native as N, dsp_plan as P

n = 5_769_231
a = torch.randn(n, device="cuda") * 0.1
out = torch.empty_like(a)
peak = torch.zeros(1, device="cuda")
starts = torch.from_numpy(P.chunk_output_starts(4194304, 104, 0, 600_000_000)).cuda()
sumsq = torch.zeros(starts.numel() * 8, dtype=torch.float64, device="cuda")

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def call(pk, ss, o):
    N.call("iqa_writer_clip", N.ptr(a), c_int64(n), N.ptr(peak) if pk else c_void_p(0), N.ptr(starts) if ss else c_void_p(0),
           c_int64(starts.numel() if ss else 0), N.ptr(sumsq) if ss else c_void_p(0), N.ptr(out) if o else c_void_p(0), N.stream_ptr())

for pk, ss, o in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (0, 0, 1), (1, 1, 0), (0, 0, 0)):
    print(f"peak={pk} sumsq={ss} out={o}: {timeit(lambda: call(pk, ss, o)):.1f} us")
print("torch clamp:", timeit(lambda: torch.clamp(a, -0.99, 0.99, out=out)), "us")
