#!/usr/bin/env python3
"""bench.sub_bench_c4_unit alone (for rocprofv3 --kernel-trace --stats).  python3 profiles/c4_unit_run.py [steps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

r = bench.sub_bench_c4_unit(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 40, warm=20)
print({k: r[k] for k in ("ms_per_step", "roofline")})
