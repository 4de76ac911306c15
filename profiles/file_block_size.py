#!/usr/bin/env python3
"""File -> 48 kHz WAV wall time against the device block size (ProcessingPipeline.block_frames_target): smaller blocks
overlap the file -> pinned copy of block k+1, the H2D of block k and the kernels of block k-1.  10 s @ 10 MS/s and
30 s @ 20 MS/s (2.4 GB) in tmpfs; best of 3 warm runs.  python profiles/file_block_size.py"""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import iq_to_audio_amd as A  # noqa: E402
from iq_to_audio_amd import iqio  # noqa: E402
from iq_to_audio_amd.benchmark import synthetic_iq_s16  # noqa: E402
import numpy as np  # noqa: E402

root = "/dev/shm" if Path("/dev/shm").is_dir() else None
for fs, secs in ((10e6, 10.0), (20e6, 30.0)):
    with tempfile.TemporaryDirectory(prefix="iq_blk_", dir=root) as tmp:
        wav = Path(tmp) / "cap_fc-400000000Hz.wav"
        uniq = synthetic_iq_s16(fs, 2.0, 25e3)
        iqio.write_wav_iq(wav, np.tile(uniq, (int(secs / 2.0), 1)), int(fs), "s16")
        for blk in (64, 32, 16, 8, 4):
            best = 1e9
            for rep in range(4):
                pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=wav, target_freq=400e6 + 25e3, center_freq=400e6, output_path=Path(tmp) / "o.wav"))
                pipe.block_frames_target = blk * 1024 * 1024
                t0 = time.perf_counter()
                pipe.run()
                dt = time.perf_counter() - t0
                if rep:
                    best = min(best, dt)
            n = fs * secs
            print(f"{secs:g} s @ {fs / 1e6:g} MS/s ({4 * n / 1e9:.1f} GB), block {blk:3d} Mi frames: {best * 1e3:7.1f} ms = {n / best / 1e9:5.2f} GS/s = {4 * n / best / 1e9:5.1f} GB/s", flush=True)
