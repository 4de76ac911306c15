#!/usr/bin/env python3
"""Config 3 through ResidentBankRunner by tail mode (bench.sub_bench_c3): chains on one side stream / on three, the next
pass waiting for them or not, edges + combines on the side stream or the caller's, everything on one stream.
python profiles/c3_tail_modes.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402

from iq_to_audio_amd import _native as NATIVE  # noqa: E402

if os.environ.get("IQA_LIB"):  # an experiment build of the library
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
import bench  # noqa: E402
from iq_to_audio_amd.batch import ResidentBankRunner as R  # noqa: E402

modes = [dict(), dict(edges_on_side=True), dict()] if os.environ.get("IQA_LIB") else [dict(), dict(probes_on_side=True), dict(), dict(probes_on_side=True)]
defaults = dict(overlap_tails=True, edges_on_side=False, tail_streams=1, pass_waits_for_tails=False, probes_on_side=False)
for m in modes:
    for k, v in {**defaults, **m}.items():
        setattr(R, k, v)
    r = bench.sub_bench_c3(steps=12, warm=4)
    fast = r.get("all_targets_fast", {})
    print(f"{m or 'default'}: product {r['ms_per_step']:.3f} ms per capture (pass {r['roofline']['kernel_ms']:.3f}); "
          f"all fast {fast.get('ms_per_step')} (pass {fast.get('kernel_ms')})", flush=True)
