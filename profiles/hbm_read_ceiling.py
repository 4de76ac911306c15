"""Scratch: read-only streaming rate of this GPU on the bench capture (context for the roofline fraction)."""
import torch
n = 600_000_000
x = torch.randint(-30000, 30000, (2 * n,), dtype=torch.int16, device="cuda")
for name, fn in (("int32 sum (torch reduce)", lambda: x.view(torch.int32).sum()),
                 ("int64 max (torch reduce)", lambda: x.view(torch.int64).max()),
                 ("clone (read+write)", lambda: x.clone())):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = x.numel() * 2 / 1e9 * (2 if "clone" in name else 1)
    print(f"{name}: {ms:.3f} ms -> {gb / ms:.2f} TB/s")
