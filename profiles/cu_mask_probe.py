#!/usr/bin/env python3
"""Does hipExtStreamCreateWithCUMask work here, and does a kernel on a CU-masked stream run beside a pass of pair workgroups?
python profiles/cu_mask_probe.py"""
import ctypes
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        pass
print("hip runtime:", hip)
torch.zeros(1, device="cuda")
a = torch.randn(64 << 20, device="cuda")
for _ in range(3):
    b = a * 1.0001
torch.cuda.synchronize()


def masked_stream(cus_per_32: int):
    n_words = 8  # 256 CUs
    mask = (ctypes.c_uint32 * n_words)()
    for x in range(8):
        mask[x] = (1 << cus_per_32) - 1 if cus_per_32 < 32 else 0xFFFFFFFF
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(n_words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


streams = [(torch.cuda.current_stream(), "default stream")] + [(masked_stream(k), f"{8 * k}-CU stream") for k in (1, 2, 4, 8, 16, 32)]
for rep in range(2):
    for stream, label in streams:
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b = a * 1.0001
            e1.record()
        torch.cuda.synchronize()
        print(f"{label}: 10 x (256 MB read + 256 MB written) in {e0.elapsed_time(e1):.3f} ms", flush=True)
