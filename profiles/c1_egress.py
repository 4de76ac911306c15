"""Scratch: BASELINE config 1's captured step for several numbers of workgroups of the PCM16 copy into pinned memory."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from iq_to_audio_amd import batch as B
orig = B.ResidentCaptureRunner.__init__
for wg in [int(b) for b in sys.argv[1:]] or [8, 16, 32, 64]:
    def init(self, *a, _wg=wg, **k):
        orig(self, *a, **k)
        self.egress_workgroups = _wg
    B.ResidentCaptureRunner.__init__ = init
    r = bench.sub_bench_c1(steps=200, warm=50)
    print(wg, "copy workgroups:", r["ms_per_step"], "ms per capture (graph),", r["ms_per_step_direct_launches"], "direct", flush=True)
