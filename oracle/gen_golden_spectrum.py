#!/usr/bin/env python3
"""oracle/gen_golden_spectrum.py -- fixtures for the spectrum / waterfall row.

Runs the reference's own ``compute_psd`` and ``streaming_waterfall`` (``/root/reference/src/iq_to_audio/spectrum.py``,
NumPy + SciPy only) on small seeded inputs and stores inputs' recipes and outputs in ``tests/golden/spectrum.npz``.
Build container only; the reference never travels.  Arrays and scalars only.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
REF_SRC = Path("/root/reference/src")


def stream(seed: int, n: int) -> np.ndarray:
    """Two tones + noise, complex64 (re-creatable from the seed)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = 0.6 * np.exp(2j * np.pi * 0.11 * t) + 0.05 * np.exp(-2j * np.pi * 0.31 * t)
    x = x + rng.normal(scale=0.01, size=n) + 1j * rng.normal(scale=0.01, size=n)
    return x.astype(np.complex64)


def main() -> None:
    if not REF_SRC.exists():
        raise SystemExit("reference not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, str(REF_SRC))
    from iq_to_audio import spectrum as ref

    store = {}
    # compute_psd: longer than, equal to and shorter than nfft (the short one is zero-padded)
    for k, (seed, n, nfft, fs) in enumerate([(1, 5000, 1024, 2.5e6), (2, 1024, 1024, 1e6), (3, 700, 1024, 48e3), (4, 999, 999, 1e6)]):
        freqs, psd = ref.compute_psd(stream(seed, n), fs, nfft)
        store[f"psd{k}_case"] = np.array([seed, n, nfft, fs], dtype=np.float64)
        store[f"psd{k}_freqs"] = freqs
        store[f"psd{k}_db"] = psd
    # streaming_waterfall: ragged blocks (some shorter than nfft, one None, one empty), default and explicit hop,
    # enough frames to trigger two pairwise reductions with max_slices = 10 (odd and even lengths)
    x = stream(9, 40_000)
    cuts = [0, 300, 1500, 1500, 9000, 9100, 22_222, 40_000]
    for k, (nfft, hop, max_slices) in enumerate([(512, None, 400), (512, 100, 10), (256, 64, 7)]):
        chunks = [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        chunks.insert(2, None)
        freqs, avg, wf, frames = ref.streaming_waterfall(chunks, 2.0e6, nfft=nfft, hop=hop, max_slices=max_slices)
        store[f"wf{k}_case"] = np.array([nfft, -1 if hop is None else hop, max_slices], dtype=np.int64)
        store[f"wf{k}_freqs"] = freqs
        store[f"wf{k}_avg"] = avg
        store[f"wf{k}_times"] = wf.times
        store[f"wf{k}_matrix"] = wf.matrix
        store[f"wf{k}_frames"] = np.int64(frames)
    store["wf_cuts"] = np.array(cuts, dtype=np.int64)
    np.savez_compressed(GOLD / "spectrum.npz", **store)
    print(f"spectrum.npz: {(GOLD / 'spectrum.npz').stat().st_size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
