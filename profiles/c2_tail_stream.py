"""Scratch: BASELINE config 2's sustained step (bench.py's main loop, no extras) with the demodulator + resampler of a
capture on their own stream beside the next capture's channelizer (IQA_TAIL_STREAM=1) and for experiment builds of the
library (IQA_LIB=: ring depth).  python profiles/c2_tail_stream.py [steps]"""
import json, os, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
steps = sys.argv[1] if len(sys.argv) > 1 else "300"
for lib in (None, "iq-to-audio_amd/csrc/_abl/libiqa_m4.so", "iq-to-audio_amd/csrc/_abl/libiqa_m3.so"):
    for tail in ("0", "1"):
        env = dict(os.environ, IQA_TAIL_STREAM=tail)
        if lib:
            if not (root / lib).exists():
                continue
            env["IQA_LIB"] = str(root / lib)
        out = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", steps, "--warmup", "60", "--no-cpu-baseline", "--no-extras"],
                             env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, tail, "FAILED", out.stderr[-400:]); continue
        d = json.loads(line[-1])
        print(f"lib={lib or 'default':45s} tail_stream={tail}: {d['ms_per_step']:.4f} ms per capture, channelizer {d['roofline']['kernel_ms']:.4f} ms, "
              f"parity {d.get('parity', {}).get('rms_err_vs_oracle_fs_channel')}", flush=True)
