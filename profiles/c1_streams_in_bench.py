#!/usr/bin/env python3
"""bench.sub_bench_c1 called several times in one process: does the two-stream replay figure depend on what ran before?
(It did: the first call in a process measured 0.17-0.20 ms per capture, later ones 0.053 -- ONE submit of 55 ms inside the
timed loop.  This script times Python's garbage collections beside it.)   python profiles/c1_streams_in_bench.py"""
import gc
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

import bench  # noqa: E402

_t = {}
_log = []


def _cb(phase, info):
    if phase == "start":
        _t["t"] = time.perf_counter()
    else:
        _log.append((info["generation"], (time.perf_counter() - _t["t"]) * 1e3, info["collected"]))


gc.callbacks.append(_cb)
for rep in range(3):
    _log.clear()
    r = bench.sub_bench_c1()
    slow = [(g, round(ms, 1), c) for g, ms, c in _log if ms > 1.0]
    print(f"call {rep}: four streams {r['ms_per_step']:.4f} ms (host per replay {r['host_us_per_replay_four_streams']}), one stream "
          f"{r['ms_per_step_one_stream']:.4f} ms (host {r['host_us_per_replay']['mean']}), direct {r['ms_per_step_direct_launches']:.4f}; "
          f"garbage collections: {len(_log)}, those over 1 ms (generation, ms, collected): {slow}", flush=True)
