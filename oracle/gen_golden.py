"""oracle/gen_golden.py -- generate tests/golden/*.npz FROM THE REAL REFERENCE.

Runs only in the build container, where /root/reference exists.  It imports the
reference's own stage classes (``iq_to_audio.processing`` / ``iq_to_audio.decoders``
from /root/reference/src; an empty ``soundfile`` module is registered first because
the reference imports it at module scope but none of the DSP stages touch it --
SURVEY.md section 8(c)), feeds them seeded inputs and stores inputs' *recipes* plus
the reference's outputs as small fixtures.  Nothing of the reference's source is
stored: fixtures are arrays and scalars only.

The chain driven here is the hand-composed one of SURVEY.md section 8(c):
  int16 -> /32768 -> complex64 -> ComplexOscillator.mix -> OverlapSaveFIR.process
        -> Decimator.process -> decoder.process -> clip(+-0.99)
because ProcessingPipeline.run itself needs the ffmpeg binary (absent).

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import hashlib
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
REF_SRC = Path("/root/reference/src")


def _import_reference():
    if not REF_SRC.exists():
        raise SystemExit("reference not present; fixtures can only be regenerated in the build container")
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    sys.path.insert(0, str(REF_SRC))
    import iq_to_audio.decoders as dec  # noqa: E402
    import iq_to_audio.processing as proc  # noqa: E402

    return proc, dec


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _synth_s16(fs, seconds, f_off, seed=42, amplitude=0.7, noise_std=0.02):
    """Same recipe as oracle.cpu_ref.synth_capture_s16 (kept separate on purpose:
    the fixture input must not depend on the code under test)."""
    n = int(round(fs * seconds))
    t = np.arange(n, dtype=np.float64) / fs
    tone = np.exp(1j * 2.0 * np.pi * f_off * t)
    noise = np.random.default_rng(seed).normal(scale=noise_std, size=(n, 2))
    iq = np.column_stack((amplitude * tone.real + noise[:, 0], amplitude * tone.imag + noise[:, 1]))
    iq = np.clip(iq.astype(np.float32), -0.999, 0.999)
    return np.rint(iq.astype(np.float64) * 32767.0).astype(np.int16)


def _modulated_s16(fs, seconds, seed):
    """A richer small capture: NFM-modulated carrier at +f1, AM carrier at -f2, noise."""
    n = int(round(fs * seconds))
    t = np.arange(n, dtype=np.float64) / fs
    rng = np.random.default_rng(seed)
    msg = np.sin(2 * np.pi * 1000.0 * t) + 0.5 * np.sin(2 * np.pi * 2300.0 * t)
    nfm = 0.35 * np.exp(1j * (2 * np.pi * 0.125 * fs * t + 3.0 * np.cumsum(msg) * 2 * np.pi * 1000.0 / fs))
    am = 0.25 * (1.0 + 0.8 * np.sin(2 * np.pi * 700.0 * t)) * np.exp(-1j * 2 * np.pi * 0.2 * fs * t)
    x = nfm + am + rng.normal(scale=0.01, size=n) + 1j * rng.normal(scale=0.01, size=n)
    iq = np.clip(np.column_stack((x.real, x.imag)).astype(np.float32), -0.999, 0.999)
    return np.rint(iq.astype(np.float64) * 32767.0).astype(np.int16)


def _ref_chain(proc, dec, raw_s16, *, fs, f_off, bw, mode, chunk, block, fs_ch_target=96_000.0,
               deemph_us=300.0, agc=True, sign=None, iq_order="iq"):
    """Drive the reference's stage objects exactly as ProcessingPipeline.run does."""
    d = max(1, int(round(fs / fs_ch_target)))
    if fs / d > fs_ch_target * 1.5:
        d = max(int(np.floor(fs / fs_ch_target)), 1)
    fs_ch = fs / d
    taps = proc.design_channel_filter(fs, bw, d)
    osc = proc.ComplexOscillator(f_off, fs)
    fir = proc.OverlapSaveFIR(taps, block)
    dcm = proc.Decimator(d)
    decoder = dec.create_decoder(mode, deemph_us=deemph_us, agc_enabled=agc)
    decoder.setup(fs_ch)
    flat = raw_s16.reshape(-1).astype(np.float32) / np.float32(32768.0)
    i, q = (flat[0::2], flat[1::2]) if iq_order.startswith("iq") else (flat[1::2], flat[0::2])
    if iq_order.endswith("_inv"):
        q = -q
    iq = (i + 1j * q).astype(np.complex64)
    peak = 0.0
    zs, auds, dbs = [], [], []
    chosen = sign
    for s in range(0, iq.size, chunk):
        blk = iq[s : s + chunk]
        if chosen is None:
            chosen = proc.choose_mix_sign(blk, fs, f_off, taps, d)
        z = dcm.process(fir.process(osc.mix(blk, chosen)))
        zs.append(np.array(z, copy=True))
        a, st = decoder.process(z)
        dbs.append(st.rms_dbfs)
        if a.size:
            peak = max(peak, float(np.max(np.abs(a))))
        auds.append(np.clip(a, -0.99, 0.99).astype(np.float32))
    return dict(z=np.concatenate(zs), audio=np.concatenate(auds), peak=peak, sign=chosen, d=d,
                fs_ch=fs_ch, ntaps=len(taps), rms_dbfs=np.array(dbs))


def main() -> None:
    proc, dec = _import_reference()
    GOLD.mkdir(parents=True, exist_ok=True)

    # ---- (i) scalars: tune_chunk_size / decimation / filter taps ------------
    tune_cases = [(fs, req) for fs in (0.0, 48e3, 200e3, 1e6, 1.9e6, 2e6, 2.5e6, 4.9e6, 5e6, 10e6, 20e6, 50e6)
                  for req in (1, 32768, 131072, 1048576, 3_000_000, 8_000_000)]
    tune_out = [proc.tune_chunk_size(fs, req) for fs, req in tune_cases]
    tap_cases = [(2.5e6, 12500.0, 26), (1e6, 12500.0, 10), (10e6, 12500.0, 104), (20e6, 2800.0, 208),
                 (50e6, 12500.0, 521), (200e3, 12500.0, 2), (250e3, 200000.0, 3), (20e6, 10000.0, 208)]
    taps_store = {}
    for k, (fs, bw, d) in enumerate(tap_cases):
        h = proc.design_channel_filter(fs, bw, d)
        taps_store[f"taps{k}_n"] = np.int64(h.size)
        taps_store[f"taps{k}_sum"] = np.float64(h.sum())
        taps_store[f"taps{k}_sha"] = np.array(_sha(h))
        # full vector for the short ones, every 8th tap for the 32k ones (size)
        taps_store[f"taps{k}"] = h if h.size <= 8192 else h[::8].copy()
    np.savez_compressed(GOLD / "plan_and_taps.npz", tune_cases=np.array(tune_cases), tune_out=np.array(tune_out),
                        tap_cases=np.array(tap_cases), **taps_store)

    # ---- (ii) stage-by-stage on small captures ------------------------------
    # 0.2 s @ 200 kS/s modulated capture, explicit (untuned) chunk sizes incl. one
    # that is not a multiple of D and one smaller than L-1.
    fs, secs = 200e3, 0.2
    raw = _modulated_s16(fs, secs, seed=7)
    store = {"fs": fs, "seconds": secs, "seed": 7}
    f_nfm, f_am = 0.125 * fs, -0.2 * fs
    cases = []
    for mode, f_off, bw in (("nfm", f_nfm, 12500.0), ("am", f_am, 10000.0), ("usb", f_am, 2800.0),
                            ("lsb", f_am, 2800.0)):
        for chunk in (40000, 10001, 4096, 700):
            for agc in ((True, False) if mode in ("usb", "lsb") and chunk in (40000, 4096) else (True,)):
                key = f"{mode}_c{chunk}_agc{int(agc)}"
                r = _ref_chain(proc, dec, raw, fs=fs, f_off=f_off, bw=bw, mode=mode, chunk=chunk, block=8192, agc=agc)
                cases.append(key)
                store[key + "_audio"] = r["audio"]
                store[key + "_meta"] = np.array([f_off, bw, chunk, 8192, int(agc), r["sign"], r["d"], r["ntaps"], r["peak"]])
                store[key + "_rms"] = r["rms_dbfs"]
                if chunk == 10001:
                    store[key + "_z"] = r["z"]
    # forced sign -1 and iq_order variants (nfm)
    for order in ("iq", "qi", "iq_inv", "qi_inv"):
        key = f"nfm_order_{order}"
        r = _ref_chain(proc, dec, raw, fs=fs, f_off=f_nfm, bw=12500.0, mode="nfm", chunk=16384, block=8192,
                       sign=-1 if order != "iq" else None, iq_order=order)
        cases.append(key)
        store[key + "_audio"] = r["audio"]
        store[key + "_meta"] = np.array([f_nfm, 12500.0, 16384, 8192, 1, r["sign"], r["d"], r["ntaps"], r["peak"]])
    store["cases"] = np.array(cases)
    np.savez_compressed(GOLD / "small_200k.npz", **store)

    # ---- (ii-b) per-stage unit vectors --------------------------------------
    rng = np.random.default_rng(11)
    x = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64)
    osc = proc.ComplexOscillator(12345.678, 1e6)
    m1 = osc.mix(x[:3000], 1)
    m2 = osc.mix(x[3000:], 1)
    osc_n = proc.ComplexOscillator(-4321.0, 1e6)
    m3 = osc_n.mix(x, -1)
    h = proc.design_channel_filter(1e6, 12500.0, 10)
    fir = proc.OverlapSaveFIR(h, 2048)
    f1 = fir.process(x[:1000])
    f2 = fir.process(x[1000:1100])  # shorter than L-1: history shift-in branch
    f3 = fir.process(x[1100:])
    qd = dec.nfm.QuadratureDemod()
    q1 = qd.process(x[:2500])
    q2 = qd.process(x[2500:])
    de = dec.nfm.DeemphasisFilter(300.0, 96153.84615384616)
    d1 = de.process(q1)
    d2 = de.process(q2)
    dcb = dec.common.DCBlocker()
    xr = x.real.astype(np.float32)
    b1 = dcb.process(xr[:1234])
    b2 = dcb.process(xr[1234:])
    ssb = dec.ssb.SSBDecoder("usb", True)
    g1 = ssb._apply_agc((xr[:2000] * np.float32(0.01)))
    tiny = np.array([0.5, 1e-7, -2e-6, 0.0, 3e-3, -0.25, 9.9e-7, 1.1e-6], dtype=np.float32)
    g2 = ssb._apply_agc(tiny)
    sign_tone = proc.choose_mix_sign(
        np.exp(1j * 2 * np.pi * 12500.0 * np.arange(100000) / 1e6).astype(np.complex64), 1e6, 12500.0, h, 10)
    np.savez_compressed(
        GOLD / "stage_vectors.npz", seed=11, mix_a=np.concatenate([m1, m2]), mix_phase_end=osc.phase, mix_b=m3,
        fir_out=np.concatenate([f1, f2, f3]), fir_state=fir.state, quad=np.concatenate([q1, q2]),
        deemph=np.concatenate([d1, d2]), deemph_state=de.state, deemph_alpha=de.alpha,
        dc=np.concatenate([b1, b2]), agc=g1, agc_tiny_in=tiny, agc_tiny=g2, sign_tone=sign_tone)

    # ---- (iii) a 0.25 s slice of the C1 benchmark capture, 3 chunks ---------
    fs, f_off = 2.5e6, 25e3
    raw = _synth_s16(fs, 0.25, f_off)
    st = {"fs": fs, "seconds": 0.25, "f_off": f_off, "chunk": 262144, "block": 65536,
          "raw_head": raw[:4].copy(), "raw_sha": np.array(_sha(raw))}
    for mode in ("nfm", "am", "usb", "lsb"):
        r = _ref_chain(proc, dec, raw, fs=fs, f_off=f_off, bw=12500.0, mode=mode, chunk=262144, block=65536)
        st[mode + "_audio"] = r["audio"]
        st[mode + "_meta"] = np.array([r["sign"], r["d"], r["ntaps"], r["peak"]])
        if mode == "nfm":
            st["z"] = r["z"]
    np.savez_compressed(GOLD / "c1_quarter_second.npz", **st)

    # ---- (iv) full-length C1 (5 s @ 2.5 MS/s): scalars + thinned audio ------
    raw = _synth_s16(fs, 5.0, f_off)
    chunk = proc.tune_chunk_size(fs, 1_048_576)
    full = {"fs": fs, "seconds": 5.0, "f_off": f_off, "chunk": chunk, "raw_head": raw[:3].copy(),
            "raw_minmax": np.array([raw.min(), raw.max()]), "raw_sha": np.array(_sha(raw))}
    for mode in ("nfm", "am", "usb", "lsb"):
        r = _ref_chain(proc, dec, raw, fs=fs, f_off=f_off, bw=12500.0, mode=mode, chunk=chunk, block=65536)
        a = r["audio"]
        full[mode + "_n"] = np.int64(a.size)
        full[mode + "_rms"] = np.float64(np.sqrt(np.mean(a.astype(np.float64) ** 2)))
        full[mode + "_peak"] = np.float64(r["peak"])
        full[mode + "_sha"] = np.array(_sha(a))
        full[mode + "_thin"] = a[::97].copy()
        full[mode + "_head"] = a[:4096].copy()
        full[mode + "_sign"] = np.int64(r["sign"])
        if mode == "nfm":
            full["z_thin"] = r["z"][::97].copy()
            full["z_sha"] = np.array(_sha(r["z"]))
            full["z_2000"] = r["z"][2000]
            full["z_meanabs"] = np.float64(np.mean(np.abs(r["z"])))
    np.savez_compressed(GOLD / "c1_full_scalars.npz", **full)

    # ---- (v) pass-through encoders (--demod none): the reference's own _encode_iq_raw (processing.py:527-539) --------
    rng = np.random.default_rng(23)
    z = (rng.normal(scale=0.4, size=6000) + 1j * rng.normal(scale=0.4, size=6000)).astype(np.complex64)
    edges = np.array([1.0, -1.0, 0.999969, 0.99997, 0.9999695, 1.5, -1.5, 0.0, -0.0, 1e-9, -1e-9, 0.5 / 32767.0, -0.5 / 32767.0,
                      1.0 / 127.5, 0.5 / 127.5, -0.5 / 127.5, 1.5 / 127.5, 2.5 / 127.5, 0.25, -0.25, 32766.5 / 32767.0], dtype=np.float32)
    z[: edges.size] = edges + 1j * edges[::-1]
    # values that sit exactly on the uint8 rounding boundaries ((x + 1) * 127.5 = k + 0.5: np.round is half-to-even)
    ks = np.arange(0, 255, dtype=np.float64)
    ties = ((ks + 0.5) / 127.5 - 1.0).astype(np.float32)
    z[100 : 100 + ties.size] = ties + 1j * ties[::-1]
    enc = {"z": z}
    for codec, dt in (("pcm_s16le", "<i2"), ("pcm_u8", np.uint8), ("pcm_f32le", "<f4")):
        enc[codec] = np.frombuffer(proc._encode_iq_raw(z, codec), dtype=dt).copy()
    np.savez_compressed(GOLD / "encode_raw.npz", **enc)

    import scipy

    (GOLD / "README.md").write_text(
        "Golden fixtures generated by `oracle/gen_golden.py` from the reference's own stage classes\n"
        f"(reference @ /root/reference, numpy {np.__version__}, scipy {scipy.__version__}).\n"
        "Arrays and scalars only; inputs are re-creatable from the seeds/recipes stored beside them.\n"
    )
    for p in sorted(GOLD.glob("*.npz")):
        print(f"{p.name}: {p.stat().st_size/1024:.0f} KiB")


if __name__ == "__main__":
    main()
