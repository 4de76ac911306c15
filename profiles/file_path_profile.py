#!/usr/bin/env python3
"""Where the wall time of a file -> 48 kHz WAV run goes (host side): cProfile around ProcessingPipeline.run on the
reference's --benchmark capture (5 s @ 2.5 MS/s) and on 10 s @ 10 MS/s, file in tmpfs.  python profiles/file_path_profile.py"""
import cProfile
import io
import pstats
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import iq_to_audio_amd as A  # noqa: E402
from iq_to_audio_amd.benchmark import _generate_synthetic_iq  # noqa: E402

root = "/dev/shm" if Path("/dev/shm").is_dir() else None
for fs, secs in ((2.5e6, 5.0), (10e6, 10.0)):
    with tempfile.TemporaryDirectory(prefix="iq_prof_", dir=root) as tmp:
        wav = Path(tmp) / "cap_fc-400000000Hz.wav"
        _generate_synthetic_iq(wav, fs, secs, 25e3)
        cfg = dict(in_path=wav, target_freq=400e6 + 25e3, center_freq=400e6, output_path=Path(tmp) / "out.wav")
        for rep in range(3):
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            A.ProcessingPipeline(A.ProcessingConfig(**cfg)).run()
            pr.disable()
            dt = time.perf_counter() - t0
            print(f"== {secs:g} s @ {fs / 1e6:g} MS/s, run {rep}: {dt * 1e3:.1f} ms")
            if rep in (0, 2):
                s = io.StringIO()
                pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
                print("\n".join(line[:150] for line in s.getvalue().splitlines()[4:44]))
