"""GPU parity tests: every stage of the HIP path (through the C ABI) against the oracle
and against the committed reference fixtures.  Run with ``-m gpu`` on an MI355X.

Bars (BASELINE.json north_star): sample counts / indexing bit-exact; float audio within
1e-4 RMS of the CPU path.  The asserted tolerances below are far tighter than that and are
stated per test.
"""
from __future__ import annotations

import numpy as np
import pytest

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import iq_to_audio_amd as pkg

    pkg.native.lib()  # fail loudly if the HIP library is missing
    pkg.native.require_gpu()
    return pkg


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)) ** 2)))


def _modulated(fs, seconds, seed):
    n = int(round(fs * seconds))
    t = np.arange(n, dtype=np.float64) / fs
    rng = np.random.default_rng(seed)
    msg = np.sin(2 * np.pi * 1000.0 * t) + 0.5 * np.sin(2 * np.pi * 2300.0 * t)
    nfm = 0.35 * np.exp(1j * (2 * np.pi * 0.125 * fs * t + 3.0 * np.cumsum(msg) * 2 * np.pi * 1000.0 / fs))
    am = 0.25 * (1.0 + 0.8 * np.sin(2 * np.pi * 700.0 * t)) * np.exp(-1j * 2 * np.pi * 0.2 * fs * t)
    x = nfm + am + rng.normal(scale=0.01, size=n) + 1j * rng.normal(scale=0.01, size=n)
    iq = np.clip(np.column_stack((x.real, x.imag)).astype(np.float32), -0.999, 0.999)
    return np.rint(iq.astype(np.float64) * 32767.0).astype(np.int16)


# ---- stand-alone stages (the reference's pluggable stage API) ---------------------------------


def test_oscillator_matches_oracle_across_calls(A):
    rng = np.random.default_rng(11)
    x = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64)
    osc = A.ComplexOscillator(12345.678, 1e6)
    st = O.NcoState(12345.678, 1e6)
    for lo, hi in ((0, 3000), (3000, 5000)):
        got = osc.mix(x[lo:hi], 1)
        want = O.nco_mix(x[lo:hi], st, 1)
        assert isinstance(got, np.ndarray) and got.dtype == np.complex64 and got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    assert abs(osc.phase - st.phase) < 1e-12
    assert osc.mix(x[:0], 1).size == 0  # empty in -> returned unchanged


def test_oscillator_golden(A, golden):
    g = golden("stage_vectors.npz")
    rng = np.random.default_rng(int(g["seed"]))
    x = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64)
    osc = A.ComplexOscillator(-4321.0, 1e6)
    np.testing.assert_allclose(osc.mix(x, -1), g["mix_b"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("order", ["iq", "qi", "iq_inv", "qi_inv"])
@pytest.mark.parametrize("fmt", ["s16", "u8"])
def test_oscillator_ingest_formats(A, fmt, order):
    rng = np.random.default_rng(5)
    raw = rng.integers(-32768, 32767, size=2000).astype(np.int16) if fmt == "s16" else rng.integers(0, 255, size=2000).astype(np.uint8)
    got = A.ComplexOscillator(1000.0, 48000.0).mix(raw, 1, fmt=fmt, iq_order=order)
    want = O.nco_mix(O.ingest_to_complex64(raw, fmt, order), O.NcoState(1000.0, 48000.0), 1)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


def test_fir_stage_matches_overlap_save(A, golden):
    g = golden("stage_vectors.npz")
    rng = np.random.default_rng(int(g["seed"]))
    x = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64)
    h = A.design_channel_filter(1e6, 12500.0, 10)
    fir = A.OverlapSaveFIR(h, 2048)
    assert (fir.filter_len, fir.overlap, fir.fft_size) == (1025, 1024, 4096)
    got = np.concatenate([fir.process(x[:1000]), fir.process(x[1000:1100]), fir.process(x[1100:])])
    assert got.shape == (5000,) and got.dtype == np.complex64
    # fixture = the reference's complex128-FFT overlap-save rounded to complex64
    np.testing.assert_allclose(got, g["fir_out"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(fir.state, g["fir_state"])  # history = last L-1 inputs, exact
    with pytest.raises(ValueError):
        A.OverlapSaveFIR(h, 0)


def test_decimator_continuity_exact(A):
    # reference tests/test_processing.py:22-28
    d = A.Decimator(3)
    first = d.process(np.arange(9, dtype=np.complex64))
    second = d.process(np.arange(9, 18, dtype=np.complex64))
    np.testing.assert_array_equal(np.concatenate([first, second]), np.arange(0, 18, 3, dtype=np.complex64))
    d = A.Decimator(26)
    st = O.DecimState(26)
    x = np.arange(100000, dtype=np.float32).astype(np.complex64)
    for lo, hi in ((0, 7), (7, 40001), (40001, 100000)):
        np.testing.assert_array_equal(d.process(x[lo:hi]), O.decimate(x[lo:hi], st))
        assert d.offset == st.offset


def test_choose_mix_sign_positive_offset(A):
    # reference tests/test_processing.py:31-40
    fs, f = 1_000_000.0, 12_500.0
    taps = A.design_channel_filter(fs, 12_500.0, 10)
    n = np.arange(0, int(fs * 0.1))
    warm = np.exp(1j * 2.0 * np.pi * f * n / fs).astype(np.complex64)
    assert A.choose_mix_sign(warm, fs, f, taps, 10) == 1
    assert A.choose_mix_sign(np.conj(warm), fs, f, taps, 10) == -1
    assert A.choose_mix_sign(warm[:0], fs, f, taps, 10) == 1


def test_mix_sign_probes_share_a_launch(A):
    """int16 / uint8 captures: both signs' probes are the two lanes of ONE matrix-core launch and one batched reduction
    (MixSignProbe._probe_pair); the powers it reads are those of the one-sign-at-a-time path, and the decision is the
    oracle's for a carrier on either side."""
    from ctypes import c_int32, c_int64

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import _native as N
    from iq_to_audio_amd.processing import MixSignProbe

    fs, d, f_off = 2.5e6, 26, 25e3
    taps = A.design_channel_filter(fs, 12_500.0, d)
    for side in (1, -1):
        raw = D.to_device(O.synth_capture_s16(fs, 0.4, side * f_off).reshape(-1), "int16")
        pair = MixSignProbe(raw, fs, f_off, taps, d, fmt="s16")
        assert pair._valid == [True, True]
        sign = pair.result()
        powers = []  # the same probes, one sign per launch
        for s_ in (1, -1):
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=s_, decimation=d, fmt="s16")
            n_in = raw.numel() // 2
            snippet = min(n_in, max(int(fs * 0.05), len(taps) * 4, 131_072))
            n_z = -(-snippet // d)
            discard = min(len(taps), n_z // 4)
            z = D.empty(n_z - discard, "complex64")
            assert ch._kernel.run_interior_only(raw, n_in, discard, n_z - discard, z)
            zz = z.cpu().numpy()
            powers.append(float(np.mean(np.abs(zz).astype(np.float32).astype(np.float64) ** 2)))
        assert sign == side == (1 if powers[0] >= powers[1] else -1)
        assert abs(pair.power - max(powers)) <= 1e-6 * max(powers)
    # several targets at once (processing.probe_targets: both signs of every target as channels of ONE bank over the
    # snippet -- tap-row groups, lane pairs and combine launches included) against one MixSignProbe per target
    import torch

    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16
    from iq_to_audio_amd.processing import probe_targets

    fs3, d3 = 20e6, 208
    carriers = [(25e3, 0.2, "nfm"), (-150e3, 0.2, "am"), (400e3, 0.2, "usb"), (-1.1e6, 0.2, "nfm")]
    cap = D.to_device(synthetic_multi_iq_s16(fs3, 0.12, carriers, seed=5).reshape(-1), "int16")
    specs = [(25e3, A.design_channel_filter(fs3, 12_500.0, d3)), (-150e3, A.design_channel_filter(fs3, 10_000.0, d3)),
             (400e3, A.design_channel_filter(fs3, 2_800.0, d3)), (1.1e6, A.design_channel_filter(fs3, 12_500.0, d3))]  # last one: mirrored
    host_slots = torch.empty(2 * len(specs), dtype=torch.float64).pin_memory()
    grouped = probe_targets(cap, fs3, specs, d3, fmt="s16", iq_order="iq", host=host_slots)
    assert grouped is not None and len(grouped) == len(specs)
    for (f_off3, taps3), pr in zip(specs, grouped):
        one = MixSignProbe(cap, fs3, f_off3, taps3, d3, fmt="s16")
        assert pr.result() == one.result()
        assert abs(pr.power - one.power) <= 1e-6 * one.power
    assert [pr.result() for pr in grouped] == [1, 1, 1, -1]
    # the batched reduction against numpy, short (one workgroup per part, written) and long (accumulated) parts
    rng = np.random.default_rng(11)
    for n_each, parts, skip in ((1000, 3, 7), (70_000, 2, 0)):
        z = (rng.normal(size=n_each * parts) + 1j * rng.normal(size=n_each * parts)).astype(np.complex64)
        out = D.empty(parts, "float64")
        N.call("iqa_mean_power_batch", N.ptr(D.to_device(z, "complex64")), c_int64(n_each), c_int32(parts), c_int64(skip), N.ptr(out),
               N.stream_ptr())
        want = [np.mean(np.abs(z[p * n_each + skip : (p + 1) * n_each]).astype(np.float64) ** 2) for p in range(parts)]
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6)


# ---- decoders ------------------------------------------------------------------------------------


def test_quadrature_and_deemphasis(A, golden):
    from iq_to_audio_amd.decoders.nfm import DeemphasisFilter, QuadratureDemod

    g = golden("stage_vectors.npz")
    rng = np.random.default_rng(int(g["seed"]))
    x = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64)
    qd = QuadratureDemod()
    quad = np.concatenate([qd.process(x[:2500]), qd.process(x[2500:])])
    # atan2f vs numpy arctan2: a few float32 ulps at |angle| <= pi
    np.testing.assert_allclose(quad, g["quad"], rtol=0, atol=1e-6)
    assert qd.prev == x[-1]
    de = DeemphasisFilter(300.0, 96153.84615384616)
    assert de.alpha == float(g["deemph_alpha"])
    dd = np.concatenate([de.process(g["quad"][:2500]), de.process(g["quad"][2500:])])
    np.testing.assert_allclose(dd, g["deemph"], rtol=0, atol=3e-7)  # float64 scan vs float64 lfilter
    assert abs(de.state - float(g["deemph_state"])) < 1e-12
    # reference tests/test_processing.py:43-55
    fs_a = 96_000.0
    k = np.arange(0, int(fs_a * 0.01))
    tone = np.exp(1j * np.cumsum(2.0 * np.pi * 1000.0 / fs_a * np.ones_like(k))).astype(np.complex64)
    audio = QuadratureDemod().process(tone)
    shaped = DeemphasisFilter(300.0, fs_a).process(audio)
    assert audio.size == tone.size and shaped.size == audio.size and np.isfinite(shaped).all()


def test_dc_blocker_and_agc(A, golden):
    from iq_to_audio_amd.decoders.common import DCBlocker
    from iq_to_audio_amd.decoders.ssb import SSBDecoder

    g = golden("stage_vectors.npz")
    rng = np.random.default_rng(int(g["seed"]))
    xr = (rng.normal(size=5000) + 1j * rng.normal(size=5000)).astype(np.complex64).real.astype(np.float32)
    dc = DCBlocker()
    got = np.concatenate([dc.process(xr[:1234]), dc.process(xr[1234:])])
    # float64 scan vs the reference's float32 sequential loop: |y| ~ 3, 1/(1-r) = 200 steps of memory
    np.testing.assert_allclose(got, g["dc"], rtol=0, atol=2e-5)
    ssb = SSBDecoder("usb", True)
    ssb.setup(96000.0)
    import iq_to_audio_amd._dev as D

    agc = ssb.agc(D.to_device(xr[:2000] * np.float32(0.01), "float32")).cpu().numpy()
    np.testing.assert_allclose(agc, g["agc"], rtol=2e-5, atol=1e-6)
    tiny = ssb.agc(D.to_device(g["agc_tiny_in"], "float32")).cpu().numpy()
    np.testing.assert_allclose(tiny, g["agc_tiny"], rtol=2e-5, atol=1e-9)
    with pytest.raises(ValueError):
        DCBlocker(1.5)


def test_decoder_factory_and_errors(A):
    from iq_to_audio_amd.decoders import AMDecoder, NarrowbandFMDecoder, SSBDecoder

    assert isinstance(A.create_decoder("FM", deemph_us=300.0, agc_enabled=True), NarrowbandFMDecoder)
    assert isinstance(A.create_decoder("am", deemph_us=300.0, agc_enabled=True), AMDecoder)
    assert isinstance(A.create_decoder("ssb", deemph_us=300.0, agc_enabled=False), SSBDecoder)
    with pytest.raises(ValueError):
        A.create_decoder("wfm", deemph_us=300.0, agc_enabled=True)
    dec = A.create_decoder("nfm", deemph_us=300.0, agc_enabled=True)
    with pytest.raises(RuntimeError):
        dec.process(np.ones(4, dtype=np.complex64))  # setup() not called (reference nfm.py:83-84)


@pytest.mark.parametrize("mode", ["nfm", "am", "usb", "lsb"])
def test_decoders_chunked_vs_oracle(A, mode):
    rng = np.random.default_rng(21)
    z = (0.3 * (rng.normal(size=30000) + 1j * rng.normal(size=30000))).astype(np.complex64)
    z *= np.exp(1j * 0.05 * np.arange(z.size)).astype(np.complex64)
    fs_ch = 96153.84615384616
    dec = A.create_decoder(mode, deemph_us=300.0, agc_enabled=True)
    dec.setup(fs_ch)
    st, st64 = O.DemodState(mode, fs_ch), O.DcState()
    for lo, hi in ((0, 9000), (9000, 9001), (9001, 30000)):
        got, stats = dec.process(z[lo:hi])
        want, db = O.demodulate(z[lo:hi], st)
        assert got.shape == want.shape and got.dtype == np.float32
        got_c, want_c = np.clip(got, -0.99, 0.99), np.clip(want, -0.99, 0.99)
        if mode in ("usb", "lsb"):
            # AGC on, IDENTICAL input.  Logic: the two recurrences restated in float64 (same restart per call, same
            # threshold, DC state carried) must agree with the GPU on every sample.  Rounding: the reference's float32
            # loops differ from that by their own rounding amplified by 1/|s| -- on this white-noise input (|s| crosses
            # zero at random every other sample) 1.5e-4 .. 3.7e-4 RMS between the oracle's own float32 and float64
            # statements, so that is all the float32 comparison can hold.
            y64 = O.ssb_demod_f64(z[lo:hi], st64, lsb=(mode == "lsb"))
            assert np.abs(got.astype(np.float64) - y64).max() <= 1e-5 * max(1.0, float(np.abs(y64).max())), (mode, lo)
            db64 = 20.0 * np.log10(np.sqrt(np.mean(y64.astype(np.float64) ** 2) + 1e-18) + 1e-12)
            assert abs(stats.rms_dbfs - db64) < 1e-3
            assert rms(got_c - want_c) < 2e-5 + 3.0 * rms(np.clip(y64, -0.99, 0.99) - want_c), (mode, lo)
            assert abs(stats.rms_dbfs - db) < 0.05
            continue
        assert rms(got_c - want_c) < 2e-5, (mode, rms(got_c - want_c))
        assert abs(stats.rms_dbfs - db) < 1e-3
    assert set(dec.intermediates()) >= {"audio"}


# ---- fused channelizer --------------------------------------------------------------------------


@pytest.mark.parametrize("order", ["iq", "qi", "iq_inv", "qi_inv"])
@pytest.mark.parametrize("fmt", ["s16", "u8", "f32"])
def test_channelizer_formats_orders_streaming(A, fmt, order):
    """Fused kernel == mix -> overlap-save -> decimate of the oracle, for every ingest format and
    iq_order, fed in ragged blocks (history path, blocks shorter than L-1, D not dividing blocks)."""
    fs, f_off, d = 1e6, 31250.0, 10
    rng = np.random.default_rng(9)
    n = 30000
    if fmt == "s16":
        raw = rng.integers(-20000, 20000, size=2 * n).astype(np.int16)
    elif fmt == "u8":
        raw = rng.integers(0, 255, size=2 * n).astype(np.uint8)
    else:
        raw = rng.normal(scale=0.3, size=2 * n).astype(np.float32)
    taps = A.design_channel_filter(fs, 12500.0, d)
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=-1, decimation=d, fmt=fmt, iq_order=order)
    nco, fir, dst = O.NcoState(f_off, fs), O.OverlapSaveState(taps, 4096), O.DecimState(d)
    edges = [0, 7, 500, 1501, 1502, 12345, 30000]
    for lo, hi in zip(edges[:-1], edges[1:]):
        got = ch.process(raw[2 * lo : 2 * hi])
        x = O.ingest_to_complex64(raw[2 * lo : 2 * hi], fmt, order)
        want = O.decimate(O.overlap_save(O.nco_mix(x, nco, -1), fir), dst)
        assert got.shape == want.shape, (lo, hi)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)


def test_channelizer_c1_slice_against_reference_fixture(A, golden):
    """0.25 s of the --benchmark capture: decimated stream vs the REFERENCE's own output."""
    g = golden("c1_quarter_second.npz")
    fs, f_off = float(g["fs"]), float(g["f_off"])
    raw = O.synth_capture_s16(fs, float(g["seconds"]), f_off)
    taps = A.design_channel_filter(fs, 12500.0, 26)
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=26)
    z = ch.process(raw)
    assert z.shape == g["z"].shape
    assert rms(z - g["z"]) < 1e-6
    np.testing.assert_allclose(z, g["z"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("fs,bw,d,f_off", [(10e6, 12500.0, 104, 1.2e6), (20e6, 2800.0, 208, -3.3e6),
                                           (50e6, 12500.0, 521, 7.7e6)])
def test_channelizer_long_filters(A, fs, bw, d, f_off):
    """BASELINE configs 2/3/5 filter shapes (6401 / 32769 / 32001 taps) on a short noise capture."""
    rng = np.random.default_rng(2)
    n = 400_000
    raw = rng.integers(-12000, 12000, size=2 * n).astype(np.int16)
    t = np.arange(n)
    tone = 8000 * np.exp(2j * np.pi * (f_off + 900.0) / fs * t)
    raw[0::2] += np.rint(tone.real).astype(np.int16)
    raw[1::2] += np.rint(tone.imag).astype(np.int16)
    taps = A.design_channel_filter(fs, bw, d)
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
    got = np.concatenate([ch.process(raw[: 2 * 150_001]), ch.process(raw[2 * 150_001 :])])
    x = O.ingest_to_complex64(raw, "s16")
    want = O.decimate(O.overlap_save(O.nco_mix(x, O.NcoState(f_off, fs), 1), O.OverlapSaveState(taps, 65536)), O.DecimState(d))
    assert got.shape == want.shape
    assert rms(got - want) < 2e-6
    assert rms(want) > 0.05  # the tone is in the pass band: a non-trivial comparison


# ---- whole chain ---------------------------------------------------------------------------------


def _gpu_chain(A, raw, *, fs, f_off, bw, mode, chunk, agc=True, order="iq", sign=None, fmt="s16", block_chunks=3,
               want_z=False):
    """Pipeline body on in-memory frames: Channelizer + ChannelDemod over blocks of whole chunks."""
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.processing import ChannelDemod

    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, bw, d)
    flat = np.asarray(raw).reshape(-1)
    n = flat.size // 2
    if sign is None:
        sign = A.choose_mix_sign(flat[: 2 * min(chunk, n)], fs, f_off, taps, d, fmt=fmt, iq_order=order)
    from iq_to_audio_amd.processing import base_precision

    # (the precision the pipeline picks per demodulator: "full" for SSB with the AGC on, "fast" otherwise)
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=sign, decimation=d, fmt=fmt, iq_order=order,
                       precision=base_precision(mode, agc))
    dem = ChannelDemod(mode, fs_ch, deemph_us=300.0, agc_enabled=agc)
    audio = D.empty(-(-n // d), "float32")
    pos, zs = 0, []
    step = chunk * block_chunks
    for lo in range(0, n, step):
        hi = min(lo + step, n)
        z = ch.process(D.to_device(flat[2 * lo : 2 * hi], "int16" if fmt == "s16" else "uint8"))
        starts = P.chunk_output_starts(chunk, d, lo, hi - lo)
        dem.process(z, starts, audio[pos : pos + z.numel()])
        if want_z:
            zs.append(z.cpu().numpy())
        pos += z.numel()
    if want_z:
        return audio[:pos].cpu().numpy(), sign, dem, np.concatenate(zs)
    return audio[:pos].cpu().numpy(), sign, dem


def test_small_capture_against_reference_fixtures(A, golden):
    """All modes x chunkings x iq_orders of tests/golden/small_200k.npz (REFERENCE outputs)."""
    g = golden("small_200k.npz")
    fs = float(g["fs"])
    from iq_to_audio_amd import dsp_plan as P

    raw = _modulated(fs, float(g["seconds"]), int(g["seed"]))
    worst = worst_agc = 0.0
    for key in [str(c) for c in g["cases"]]:
        f_off, bw, chunk, block, agc, sign, d, ntaps, peak = g[key + "_meta"]
        if key.startswith("nfm_order_"):
            order = key[len("nfm_order_"):]
            mode, override = "nfm", (-1 if order != "iq" else None)
        else:
            order, mode, override = "iq", key.split("_")[0], None
        got, got_sign, dem, z_got = _gpu_chain(A, raw, fs=fs, f_off=float(f_off), bw=float(bw), mode=mode, chunk=int(chunk),
                                               agc=bool(agc), order=order, sign=override, want_z=True)
        want = g[key + "_audio"]
        assert got.shape == want.shape, key  # sample count exact
        assert got_sign == int(sign), key
        err = rms(got - want)
        if mode in ("usb", "lsb") and bool(agc):
            # ill-conditioned in the reference itself: held link by link (test_gpu_configs.ssb_agc_evidence) against
            # the oracle's z and audio of the same case
            from test_gpu_configs import chunk_lens_for, ssb_agc_evidence

            ref = O.run_chain(raw, sample_rate=fs, freq_offset=float(f_off), bandwidth=float(bw), demod_mode=mode,
                              chunk_size=int(chunk), filter_block=int(block), tune_chunk=False)
            lens = chunk_lens_for(raw.shape[0], int(chunk), int(d), ref.decimated.size)
            ev = ssb_agc_evidence(f"fixture {key}", z_got, got, ref.decimated, ref.audio, lens, mode, ref.fs_channel, z_tol=2e-5)
            worst_agc = max(worst_agc, ev["err"])
            continue
        worst = max(worst, err)
        assert err < 1e-4, (key, err)  # the north_star bar
        assert err < 2e-5, (key, err)  # what we actually hold
        assert abs(dem.peak - float(peak)) <= 2e-4 * max(1.0, float(peak)), key
    print("worst rms error over fixture cases:", worst, "(SSB+AGC cases:", worst_agc, ")")


@pytest.mark.parametrize("mode", ["nfm", "am", "usb", "lsb"])
def test_c1_full_length_against_reference_scalars(A, golden, mode):
    """BASELINE config 1 at FULL size (5 s @ 2.5 MS/s, chunk 1 048 576) vs the reference's scalars
    and thinned audio, and vs the oracle sample by sample."""
    g = golden("c1_full_scalars.npz")
    fs, f_off = float(g["fs"]), float(g["f_off"])
    raw = O.synth_capture_s16(fs, float(g["seconds"]), f_off)
    chunk = int(g["chunk"])
    got, sign, dem, z_got = _gpu_chain(A, raw, fs=fs, f_off=f_off, bw=12500.0, mode=mode, chunk=chunk, block_chunks=5,
                                       want_z=True)

    assert got.size == int(g[mode + "_n"]) == 480_770  # sample count: exact
    assert sign == int(g[mode + "_sign"]) == 1
    want = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, demod_mode=mode, keep_decimated=True)
    db = dem.chunk_rms_dbfs()
    assert len(db) == len(want.rms_dbfs) == 12
    if mode in ("usb", "lsb"):
        # SSB + AGC on this capture (carrier at DC -> DC-blocked residue crossing zero all the time) is
        # ill-conditioned in the REFERENCE: held link by link (z / logic / rounding / sensitivity), with the strict
        # replay bar (>= 99 % of samples within 1e-4, median <= 1e-6, RMS < 1e-4 against the reference's own float32
        # loops on the same input) because this is the capture the north-star bar is stated on.
        from test_gpu_configs import chunk_lens_for, ssb_agc_evidence

        lens = chunk_lens_for(raw.shape[0], chunk, 26, want.decimated.size)
        ev = ssb_agc_evidence(f"C1 {mode} (3 blocks)", z_got, got, want.decimated, want.audio, lens, mode, want.fs_channel,
                              z_tol=2e-7, strict_replay=True)
        assert ev["err"] < 1.5e-2  # ("fast" z, 2.0e-6 off: 3.5e-2; "full" z, 3.3e-8 off -- the float32 grid of z itself: 6.5e-3)
        assert abs(rms(got) - float(g[mode + "_rms"])) < 0.01 * float(g[mode + "_rms"])
        np.testing.assert_allclose(db, want.rms_dbfs, atol=0.5)
        # with the AGC off the same path is well-conditioned and meets the tight bar
        got_off, _, _ = _gpu_chain(A, raw, fs=fs, f_off=f_off, bw=12500.0, mode=mode, chunk=chunk, block_chunks=5, agc=False)
        want_off = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, demod_mode=mode, agc_enabled=False, keep_decimated=False)
        assert rms(got_off - want_off.audio) < 2e-5
        return
    assert rms(got[::97] - g[mode + "_thin"]) < 2e-5
    assert rms(got[:4096] - g[mode + "_head"]) < 2e-5
    assert abs(rms(got) - float(g[mode + "_rms"])) < 1e-5
    assert abs(dem.peak - float(g[mode + "_peak"])) < 2e-4 * max(1.0, float(g[mode + "_peak"]))
    assert rms(got - want.audio) < 2e-5
    np.testing.assert_allclose(db, want.rms_dbfs, atol=1e-3)


def test_resampler_matches_spec(A):
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.processing import Resampler48k

    fs_ch = 2.5e6 / 26
    rng = np.random.default_rng(4)
    t = np.arange(50_000) / 96154.0
    x = (0.4 * np.sin(2 * np.pi * 1000 * t) + 0.05 * rng.normal(size=t.size)).astype(np.float32)
    rs = Resampler48k(fs_ch)
    y = rs.process(D.to_device(x, "float32"))
    want = O.resample_48k(x, fs_ch)
    assert y.numel() == want.size == -(-x.size * 24000 // 48077)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=2e-7)
    pcm = rs.to_pcm16(y).cpu().numpy()
    np.testing.assert_array_equal(pcm, O.float_to_pcm16(y.cpu().numpy()))
    # the one-pass forms: PCM16 straight from the resampler, alone and beside the float32 stream
    y2, pcm2 = rs.process(D.to_device(x, "float32"), want="both")
    np.testing.assert_array_equal(y2.cpu().numpy(), y.cpu().numpy())
    np.testing.assert_array_equal(pcm2.cpu().numpy(), pcm)
    np.testing.assert_array_equal(rs.process(D.to_device(x, "float32"), want="pcm16").cpu().numpy(), pcm)
    with pytest.raises(ValueError):
        rs.process(D.to_device(x, "float32"), want="s24")
    # a later stretch of the output stream through the C ABI (j0 > 0: the residues of one wave wrap around `up`,
    # the one place where the kernel reads its taps' samples straight from memory instead of the staged window)
    from ctypes import c_int32, c_int64

    from iq_to_audio_amd import _native as N

    xd, j0, cnt = D.to_device(x, "float32"), 12_345, 9_000
    part = D.empty(cnt, "float32")
    N.call("iqa_resample", N.ptr(xd), c_int64(x.size), N.ptr(rs.table_dev), c_int32(rs.plan.up), c_int32(rs.plan.down),
           c_int32(rs.plan.half_taps), c_int64(j0), c_int64(cnt), N.ptr(part), N.ptr(None), N.stream_ptr())
    np.testing.assert_array_equal(part.cpu().numpy(), y.cpu().numpy()[j0 : j0 + cnt])
    # stream lengths that are not multiples of 4 samples (the staging DMA moves 16 bytes per lane: its last lane
    # straddles the end of the stream) and one shorter than a filter row
    for m in (49_999, 49_998, 49_997, 77, 5):
        ym, pm = rs.process(D.to_device(x[:m].copy(), "float32"), want="both")
        wm = O.resample_48k(x[:m], fs_ch)
        np.testing.assert_allclose(ym.cpu().numpy(), wm, rtol=0, atol=2e-7)
        np.testing.assert_array_equal(pm.cpu().numpy(), O.float_to_pcm16(ym.cpu().numpy()))
    # C5's rate: gcd(48000, 95969) == 1
    rs5 = Resampler48k(50e6 / 521)
    y5 = rs5.process(D.to_device(x, "float32")).cpu().numpy()
    np.testing.assert_allclose(y5, O.resample_48k(x, 50e6 / 521), rtol=0, atol=2e-7)


@pytest.mark.parametrize("mode,agc", [("nfm", True), ("am", True), ("usb", True), ("lsb", False)])
def test_demodulate_from_reset_equals_reset_then_demodulate(A, mode, agc):
    """iqa_demodulate_from_reset (what ChannelDemod.process calls after a reset) against the explicit way: the pristine
    state image copied over a used decoder's state block, then iqa_demodulate -- audio, outgoing state and peak bit for
    bit, the per-chunk sums to rounding, for a block whose chunk starts make the SSB AGC restart inside it."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.processing import ChannelDemod

    rng = np.random.default_rng(21)
    n, fs_ch = 300_001, 96_153.8
    t = np.arange(n) / fs_ch
    z = (0.3 * np.exp(2j * np.pi * (900.0 * t + 2.0 * np.sin(2 * np.pi * 3.0 * t))) * (1.0 + 0.4 * np.sin(2 * np.pi * 440.0 * t))
         + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))).astype(np.complex64)
    z_dev = D.to_device(z, "complex64")
    starts = np.array([0, 40_330, 80_660, 120_990, 161_320, 201_650, 241_980, 282_310], dtype=np.int64)
    dem = ChannelDemod(mode, fs_ch, deemph_us=300.0, agc_enabled=agc)
    used = D.empty(n, "float32")
    dem.process(z_dev[: n // 2].contiguous() * 0.5, np.array([0], dtype=np.int64), used)  # the decoder has seen something
    # (a) the explicit way
    dem.prepare(n, starts)
    dem._alloc_block(len(starts))  # a pristine block of the right size (uploads the image)
    dem._from_reset, dem._fresh, dem.chunk_sumsq = False, True, []
    a_audio = D.empty(n, "float32")
    dem.process(z_dev, starts, a_audio)
    a_state, a_peak, a_sums = dem.state_dev.clone(), dem.peak, dem.chunk_sumsq[-1][0].clone()
    # (b) a used decoder, reset() (no copy), process (iqa_demodulate_from_reset)
    dem.process(z_dev[: n // 3].contiguous(), np.array([0], dtype=np.int64), used)
    dem.reset()
    assert dem._from_reset and dem.peak == 0.0
    b_audio = D.empty(n, "float32")
    dem.process(z_dev, starts, b_audio)
    assert torch.equal(a_audio, b_audio)
    assert torch.equal(a_state, dem.state_dev) and a_peak == dem.peak
    # (the sums are float64 atomics of several workgroups per slot: equal up to the order of additions)
    assert torch.allclose(a_sums, dem.chunk_sumsq[-1][0], rtol=1e-12, atol=0.0)
    assert a_peak > 0.0 and float(a_sums.sum()) > 0.0


@pytest.mark.parametrize("fs_ch", [150_000.0, 250_000.0, 192_000.0, 48_000.0, 44_100.0, 24_000.0, 8_000.0, 96_000.0, 131_071.0])
def test_resampler_other_ratios(A, fs_ch):
    """Channel rates other than the 96 kHz class: longer polyphase rows (the 24 / 32 / 48 taps-per-lane builds; past the
    one-instruction window the waves read their samples straight from memory), integer ratios, rates below 48 kHz
    (up > down: few residues, many steps) and a prime rate (up = 48000)."""
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.processing import Resampler48k

    rng = np.random.default_rng(int(fs_ch))
    n = 30_011
    t = np.arange(n) / fs_ch
    x = (0.5 * np.sin(2 * np.pi * 700 * t) + 0.1 * rng.normal(size=n)).astype(np.float32)
    rs = Resampler48k(fs_ch)
    y, pcm = rs.process(D.to_device(x, "float32"), want="both")
    want = O.resample_48k(x, fs_ch)
    assert y.numel() == want.size
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=3e-7)
    np.testing.assert_array_equal(pcm.cpu().numpy(), O.float_to_pcm16(y.cpu().numpy()))


def test_pipeline_end_to_end_wav(A, tmp_path):
    """ProcessingPipeline.run on a WAV written to disk: result fields, 48 kHz PCM16 file, sample
    count, and the channel-rate audio against the oracle."""
    from iq_to_audio_amd import iqio
    from iq_to_audio_amd.benchmark import synthetic_iq_s16

    fs, f_off, secs = 2.5e6, 25e3, 1.0
    raw = synthetic_iq_s16(fs, secs, f_off)
    np.testing.assert_array_equal(raw, O.synth_capture_s16(fs, secs, f_off))
    wav = tmp_path / "cap_400000000Hz.wav"
    iqio.write_wav_iq(wav, raw, int(fs), "s16")
    cfg = A.ProcessingConfig(in_path=wav, target_freq=400_025_000.0, output_path=tmp_path / "out.wav")
    pipe = A.ProcessingPipeline(cfg)
    pipe.keep_channel_audio = True
    res = pipe.run()
    assert (res.decimation, res.mix_sign) == (26, 1) and res.center_freq == 400e6 and abs(res.freq_offset - 25e3) < 1e-6
    want = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, keep_decimated=False)
    got = pipe.audio_fs_channel.cpu().numpy()
    assert got.size == want.audio.size
    assert rms(got - want.audio) < 2e-5
    assert abs(res.audio_peak - want.audio_peak) < 1e-5
    pcm, rate = iqio.read_wav_pcm16_mono(tmp_path / "out.wav")
    assert rate == 48000
    ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
    assert pcm.size == ref48.size
    assert np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32))) <= 1
    # errors as the reference raises them
    with pytest.raises(ValueError):
        A.ProcessingPipeline(A.ProcessingConfig(in_path=wav, target_freq=0.0, center_freq=4e8)).run()
    with pytest.raises(ValueError):
        A.ProcessingPipeline(A.ProcessingConfig(in_path=wav, target_freq=4e8, center_freq=4e8, bandwidth=-1)).run()
    with pytest.raises(ValueError):
        A.ProcessingPipeline(A.ProcessingConfig(in_path=wav, target_freq=4e8, center_freq=4e8, iq_order="xx")).run()


@pytest.mark.parametrize("resident", [False, True])
def test_resident_capture_runner_batch_and_sign_speculation(A, resident):
    """batch.ResidentCaptureRunner: a batch of device-resident captures with one set of settings, every
    capture queued without a host sync (speculative mixer sign +1; ``resident``: probes, decoder-state reset and the
    start-up outputs on the aux stream, ahead of the compute stream).  Each capture's 48 kHz PCM16 must equal the
    oracle's (<= 1 LSB), captures must not bleed into each other through the two buffer slots, and a capture whose
    probe picks sign -1 must come out as if the sign had been known."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner

    fs, f_off, secs = 10e6, 25e3, 0.9
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    n = int(round(fs * secs))
    caps = [O.synth_capture_s16(fs, secs, f_off, seed=42),      # sign +1
            O.synth_capture_s16(fs, secs, -f_off, seed=43),     # carrier on the other side: the probe picks -1
            O.synth_capture_s16(fs, secs, f_off, seed=44),
            O.synth_capture_s16(fs, secs, f_off, seed=45)]
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                   n_frames=n, demod_mode="nfm")
    devs = [D.to_device(c.reshape(-1), "int16") for c in caps]
    torch.cuda.synchronize()  # resident=True promises complete captures
    tickets = [runner.submit(x, resident=resident) for x in devs]   # submit 3 and 4 collect 1 and 2 to free their slots
    got = []
    for t in tickets:
        r = runner.collect(t)
        got.append((r["sign"], None if t is not tickets[-1] and t is not tickets[-2] else r["pcm_host"].numpy().copy(), r["kernel"]))
    # collected results of captures whose slot was reused later are only checked through what was copied at collect
    signs = [g[0] for g in got]
    assert signs == [1, -1, 1, 1]
    assert got[-1][2] == "k_channelize_mfma_s16_ring"
    for idx in (2, 3):
        want = O.run_chain(caps[idx], sample_rate=fs, freq_offset=f_off, keep_decimated=False)
        ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
        pcm = got[idx][1]
        assert pcm.size == ref48.size == runner.n48
        assert np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32))) <= 1
    # the mis-speculated capture, alone: same answer as the oracle that is told nothing about the sign
    t = runner.submit(devs[1], resident=resident)
    r = runner.collect(t)
    want = O.run_chain(caps[1], sample_rate=fs, freq_offset=f_off, keep_decimated=False)
    assert r["sign"] == want.mix_sign == -1
    audio = r["audio"].cpu().numpy()
    assert audio.size == want.audio.size and rms(audio - want.audio) < 2e-5
    ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
    assert np.max(np.abs(r["pcm_host"].numpy().astype(np.int32) - ref48.astype(np.int32))) <= 1
    assert abs(r["demod"].peak - want.audio_peak) < 1e-5
    # the same capture inside a padded buffer (readable slack behind it; a lead-in of zeros is supported but not
    # recommended, see padded_capture_frames): the last outputs come from the matrix-core kernel too
    z_plain = runner.collect(runner.submit(devs[2], resident=resident))["z"].clone()
    lead, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = torch.zeros(2 * (lead + n + slack), dtype=torch.int16, device=devs[2].device)
    buf[2 * lead : 2 * (lead + n)] = devs[2]
    buf[2 * (lead + n) :] = 12345  # the slack is read but must never matter
    torch.cuda.synchronize()
    r = runner.collect(runner.submit(buf[2 * lead : 2 * (lead + n)], enclosing=buf, lead_frames=lead, resident=resident))
    want = O.run_chain(caps[2], sample_rate=fs, freq_offset=f_off)
    assert r["sign"] == 1 and r["z"].numel() == want.decimated.size
    z_pad = r["z"].cpu().numpy()
    assert rms(z_pad - want.decimated) < 2e-5 and np.abs(z_pad - want.decimated).max() < 1e-4
    assert np.abs(z_pad[:200] - want.decimated[:200]).max() < 1e-4 and np.abs(z_pad[-200:] - want.decimated[-200:]).max() < 1e-4
    assert float((r["z"] - z_plain).abs().max()) < 1e-4
    ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
    assert np.max(np.abs(r["pcm_host"].numpy().astype(np.int32) - ref48.astype(np.int32))) <= 1


@pytest.mark.parametrize("fs,fmt,secs,bw,precision", [(10e6, "s16", 60.0, 12_500.0, "fast"), (20e6, "s16", 60.0, 12_500.0, "fast"),
                                                      (20e6, "u8", 60.0, 12_500.0, "fast"), (50e6, "s16", 120.0, 12_500.0, "fast"),
                                                      (20e6, "s16", 60.0, 2_800.0, "full"), (10e6, "s16", 60.0, 12_500.0, "fine")])
def test_full_size_capture_windows_match_float32_kernel(A, fs, fmt, secs, bw, precision):
    """BASELINE's full sizes (config 2: 600 M frames = 2.4 GB; config 4's unit: 1.2 G frames = 4.8 GB of int16, 2.4 GB
    of uint8; config 5: 6 G frames = 24 GB, D = 521, three k-step passes, one of its channels): the matrix-core
    channelizer's output over the whole capture against the float32 VALU kernel run on short slices of it
    (``consumed`` = the slice's position in the capture) -- at the start, around the frames whose byte offsets are
    2^31 ... 2^34 (address and index arithmetic: config 5 has more than 2^32 frames), in the middle and at the very
    end.  A size-independent property: the output at position m depends on frames [m D - L + 1, m D] and on the
    absolute sample index only.  The last two cases are this at the finer precisions: config 3's SSB filter (32 769 taps,
    three tap-row groups) at "full" -- six lanes with 64-bit sums in pairs over the 4.8 GB capture -- and config 2 at
    "fine" (taps + residue lanes of the loader-wave kernel), held to 5e-6 (max) instead of 1e-4."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner

    f_off = 25e3
    n = int(round(fs * secs))
    d, _ = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, bw, d)
    ntaps = len(taps)
    unique = O.synth_capture_s16(fs, 0.37, f_off, seed=5)  # 0.37 s: no small period against D or the windows
    if fmt == "u8":
        unique = ((unique.astype(np.int32) >> 8) + 128).astype(np.uint8)
    tile = D.to_device(unique.reshape(-1), "int16" if fmt == "s16" else "uint8")
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, ntaps)
    reps = -(-(n + slack) // (tile.numel() // 2))
    buf = tile.repeat(reps)[: 2 * (n + slack)].contiguous()
    raw = buf[: 2 * n]
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt=fmt, precision=precision)
    z = ch.process(raw, last_block=True, halo=(buf, 0))
    assert ch.precision == precision
    assert ch._kernel.last_kernel == ("k_channelize_mfma_s16_ring" if fmt == "s16" else "k_channelize_mfma_u8_ring")
    n_out = -(-n // d)
    assert z.numel() == n_out
    bytes_per_frame = 4 if fmt == "s16" else 2
    marks = [0, n // 2, n - 1] + [(1 << e) // bytes_per_frame for e in (31, 32, 33, 34) if (1 << e) // bytes_per_frame < n]
    span = 1500  # outputs compared per window
    worst = 0.0
    for mark in marks:
        m_mid = min(max(mark // d, 0), n_out - 1)
        m_lo = max(0, m_mid - span // 2)
        m_hi = min(n_out, m_lo + span)
        f_lo = max(0, m_lo * d - (ntaps - 1))           # first frame any compared output depends on
        f_lo -= f_lo % d                                 # (keeps the slice's decimator phase at 0: simpler indexing)
        f_hi = min(n, (m_hi - 1) * d + 1)
        part = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt=fmt)
        part.consumed = f_lo                             # the slice sits at this position of the capture
        zs = part.process(raw[2 * f_lo : 2 * f_hi], last_block=True)
        assert part._kernel.last_kernel == "k_channelize_v1"  # short block: the float32 kernel
        first = f_lo // d                                # output index of zs[0]
        # outputs that see the zero history of the slice instead of the capture are skipped (none when f_lo == 0)
        m0 = m_lo if f_lo == 0 else max(m_lo, -(-(f_lo + ntaps - 1) // d))
        got, want = z[m0:m_hi], zs[m0 - first : m_hi - first]
        assert got.numel() == want.numel() and got.numel() >= span // 2
        err = float((got - want).abs().max())
        worst = max(worst, err)
        bound = (1e-4 if fmt == "s16" else 2e-4) * max(1.0, float(np.sqrt(ntaps / 6401.0))) if precision == "fast" else 5e-6  # (the float32 kernel it is compared with carries up to ~3e-6 itself on 32 769 taps)
        assert err < bound, (mark, err, precision)
    assert worst > 0.0  # (two different kernels: identical output would mean the comparison compared nothing)


def test_full_size_chain_windows(A):
    """BASELINE config 2 at full size through the batch path (600 M frames -> 5.77 M channel-rate samples -> 2.88 M
    samples at 48 kHz): the demodulator and the resampler at the start, across a reference-chunk boundary, in the middle
    and at the end of the stream against the same stages run on short slices with fresh state (the de-emphasis filter
    forgets its state within ~1000 samples; a 48 kHz output depends on 67 inputs), plus the whole-stream invariants:
    sample counts, per-chunk levels, peak."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner
    from iq_to_audio_amd.processing import ChannelDemod, Resampler48k

    fs, f_off, secs = 10e6, 25e3, 60.0
    n = int(round(fs * secs))
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    unique = O.synth_capture_s16(fs, 0.37, f_off, seed=6)
    tile = D.to_device(unique.reshape(-1), "int16")
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = tile.repeat(-(-(n + slack) // (tile.numel() // 2)))[: 2 * (n + slack)].contiguous()
    torch.cuda.synchronize()
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                   n_frames=n, demod_mode="nfm")
    r = runner.collect(runner.submit(buf[: 2 * n], enclosing=buf, lead_frames=0, resident=True))
    z, audio = r["z"], r["audio"]
    n_dec = -(-n // d)
    assert r["sign"] == 1 and z.numel() == audio.numel() == n_dec == 5_769_231
    pcm = torch.from_numpy(r["pcm_host"].numpy().copy())
    assert pcm.numel() == runner.n48 == 2_879_996
    starts = runner.starts
    assert len(starts) == 144 and len(r["demod"].chunk_rms_dbfs()) == 144
    assert abs(float(audio.abs().max()) - min(r["demod"].peak, 0.99)) < 1e-6
    settle, span = 4000, 3000
    for mid in (0, int(starts[71]), n_dec // 2 + 777, n_dec):  # start, a chunk boundary, middle, end
        lo = max(0, min(mid - span // 2, n_dec - span))
        s0 = max(0, lo - settle)
        dem = ChannelDemod("nfm", fs_ch, deemph_us=300.0, agc_enabled=True)
        part = D.empty(lo + span - s0, "float32")
        dem.process(z[s0 : lo + span], np.array([0], dtype=np.int64), part)
        err = float((part[lo - s0 :] - audio[lo : lo + span]).abs().max())
        assert err < 2e-6, (mid, err)  # float64 scans combined in a different order, float32 outputs
    rs = Resampler48k(fs_ch)
    up, down = rs.plan.up, rs.plan.down
    for mid48 in (0, runner.n48 // 2, runner.n48):
        k = max(0, min(mid48 // up - 1, (n_dec - 2 * down) // down))  # slice starts at input index k * down
        s, j0 = k * down, k * up
        cnt = min(2 * up, runner.n48 - j0)
        y = rs.process(audio[s:], want="pcm16")[:cnt]
        edge = 40 if s > 0 else 0  # outputs whose 67-sample window reaches in front of the slice
        assert torch.equal(y[edge:].cpu(), pcm[j0 + edge : j0 + cnt]), mid48


@pytest.mark.parametrize("mode,agc,fs,bw", [("am", True, 10e6, 10_000.0), ("usb", False, 20e6, 2_800.0), ("nfm", True, 5e6, 12_500.0)])
def test_resident_capture_runner_other_modes_and_rates(A, mode, agc, fs, bw):
    """The batch path for the other demodulators and capture shapes: AM at the C2 rate, USB (AGC off: the AGC case is
    ill-conditioned in the reference itself) on the C3 narrow filter (32769 taps, three tap-row groups
    chained through partial sums), NFM at 5 MS/s (D = 52: four k steps).  48 kHz PCM16 within 1 LSB of the oracle's."""
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner

    f_off, secs = 25e3, 0.45
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, bw, d)
    n = int(round(fs * secs))
    cap = O.synth_capture_s16(fs, secs, f_off, seed=7)
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                   n_frames=n, demod_mode=mode, agc_enabled=agc)
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    import torch

    buf = torch.zeros(2 * (n + slack), dtype=torch.int16, device=D.device())
    buf[: 2 * n] = D.to_device(cap.reshape(-1), "int16")
    r = runner.collect(runner.submit(buf[: 2 * n], enclosing=buf, lead_frames=0))
    want = O.run_chain(cap, sample_rate=fs, freq_offset=f_off, bandwidth=bw, demod_mode=mode, agc_enabled=agc,
                       keep_decimated=False)
    assert r["sign"] == want.mix_sign and r["kernel"] == "k_channelize_mfma_s16_ring"
    audio = r["audio"].cpu().numpy()
    assert audio.size == want.audio.size and rms(audio - want.audio) < 5e-5
    ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
    pcm = r["pcm_host"].numpy()
    assert pcm.size == ref48.size and np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32))) <= 2
    assert abs(r["demod"].peak - want.audio_peak) < 1e-4


@pytest.mark.parametrize("fs,bw", [(2.5e6, 2_800.0), (20e6, 2_800.0)])
def test_batch_runners_run_ssb_with_agc_at_full_precision(A, fs, bw):
    """SSB with the AGC on through the batch path -- ResidentCaptureRunner (direct launches and the captured hipGraph step)
    and ResidentBankRunner -- at the reference's --benchmark rate (D = 26: chained passes of the per-lane kernel) and at
    config 3's (D = 208: lanes with 64-bit sums): the runners pick the "full" precision themselves
    (processing.base_precision), z within 3e-7 of the oracle's, and the audio is what the float64 statement of the decoder
    makes of that very z on EVERY sample (the logic link of DESIGN.md section 5: a wrong restart index or state hand-off
    shows at the 1e-2 level)."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner
    from test_gpu_configs import chunk_lens_for, oracle_ssb_f64_chunks

    f_off, secs = 25e3, 1.1 if fs < 5e6 else 0.5
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, bw, d)
    n = int(round(fs * secs))
    cap = O.synth_capture_s16(fs, secs, f_off, seed=11)
    want = O.run_chain(cap, sample_rate=fs, freq_offset=f_off, bandwidth=bw, demod_mode="usb", agc_enabled=True)
    lens = chunk_lens_for(n, chunk, d, want.decimated.size)
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = torch.zeros(2 * (n + slack), dtype=torch.int16, device=D.device())
    buf[: 2 * n] = D.to_device(cap.reshape(-1), "int16")
    raw = buf[: 2 * n]
    torch.cuda.synchronize()

    def check(label, r):
        z, audio = r["z"].cpu().numpy(), r["audio"].cpu().numpy()
        assert r["precision"] == "full" and r["sign"] == want.mix_sign, label
        assert z.size == want.decimated.size and audio.size == want.audio.size, label
        dz = rms(z - want.decimated)
        logic = np.abs(audio.astype(np.float64) - oracle_ssb_f64_chunks(z, lens, "usb")).max()
        print(f"SSB+AGC through {label} at {fs / 1e6:g} MS/s: z rms diff {dz:.2e}, logic max {logic:.2e}")
        assert dz < 3e-7 and logic < 1e-5, (label, dz, logic)

    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                   n_frames=n, demod_mode="usb", agc_enabled=True)
    assert runner.base_precision == "full"
    check("ResidentCaptureRunner.submit", runner.collect(runner.submit(raw, enclosing=buf, lead_frames=0)))
    for _ in range(2):  # capture, then replay
        check("ResidentCaptureRunner.submit_captured", runner.collect(runner.submit_captured(raw, enclosing=buf, lead_frames=0)))
    bank = ResidentBankRunner([dict(freq_offset=f_off, demod_mode="usb", bandwidth=bw), dict(freq_offset=f_off, demod_mode="nfm")],
                              sample_rate=fs, n_frames=n)
    res = bank.collect(bank.submit(raw, enclosing=buf, lead_frames=0))
    check("ResidentBankRunner", res[0])
    assert res[1]["precision"] == "fast"


def test_pipeline_cancel_removes_partial_output(A, tmp_path):
    """reference tests/test_processing.py:125-151: cancelling raises ProcessingCancelled and leaves no file."""
    from iq_to_audio_amd import iqio
    from iq_to_audio_amd.benchmark import synthetic_iq_s16
    from iq_to_audio_amd.progress import NullProgressSink

    wav = tmp_path / "c.wav"
    iqio.write_wav_iq(wav, synthetic_iq_s16(1e6, 0.3, 10e3), 1_000_000, "s16")
    out = tmp_path / "o.wav"
    out.write_bytes(b"stale")
    pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=wav, target_freq=1.0001e8, center_freq=1e8, output_path=out))

    class Sink(NullProgressSink):
        def status(self, message):
            if message.startswith("channel"):
                pipe.cancel()

    with pytest.raises(A.ProcessingCancelled):
        pipe.run(Sink())
    assert not out.exists()


# ---- int8-MFMA form of the channelizer -----------------------------------------------------------


@pytest.mark.parametrize("fs,d,bw,n", [
    (2.5e6, 26, 12500.0, 4_000_000),      # C1 shape: 1601 taps, one pass
    (10e6, 104, 12500.0, 6_000_000),      # C2 shape: 6401 taps, one pass
    (20e6, 208, 2800.0, 12_000_000),      # C3 narrow channel: 32769 taps = 158 tap rows -> 3 q-groups
    (50e6, 521, 12500.0, 16_000_000),     # C5 shape: 32001 taps, 33 k steps -> 3 k-step ranges
])
@pytest.mark.parametrize("variant", ["plain", "ring"])
def test_mfma_channelizer_vs_valu_and_oracle(A, fs, d, bw, n, variant):
    """The matrix-core path (exact int32 accumulation of int8 pieces, fixed-point taps) against the float32 VALU
    kernel and the oracle: error is the documented tap-quantisation floor (16-bit taps, ~2e-6 of full scale, for
    the per-lane kernel; ~14-bit taps, ~1e-5, for the ring kernel, which keeps 256*S1+S2 in one int32), results
    are bit-reproducible, and head/tail outputs (history / end of block) are seamless.  The ring kernel covers every
    shape: contiguous slots where rows are 16-byte aligned and fit one pass, row-staged slots for C1 (D = 26) and C5
    (D = 521: three k-step ranges of 11, chained through partial sums, each fetching its own third of every row)."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    f_off = 25e3
    raw = O.synth_capture_s16(fs, n / fs, f_off).reshape(-1)
    taps = A.design_channel_filter(fs, bw, d)
    x = D.to_device(raw, "int16")
    old_min, old_variant = PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.mfma_variant
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        PR._ChannelKernel.mfma_variant = variant
        outs = {}
        for use in (False, True):
            PR._ChannelKernel.use_mfma = use
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
            # two ragged blocks: the second one starts with a history and a non-zero decimator phase
            cut = 2 * 1_500_001
            z = torch.cat([ch.process(x[:cut]), ch.process(x[cut:])])
            assert (ch._kernel.last_kernel.startswith("k_channelize_mfma_s16")) == use
            outs[use] = z
        PR._ChannelKernel.use_mfma = True
        ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
        again = torch.cat([ch.process(x[: 2 * 1_500_001]), ch.process(x[2 * 1_500_001 :])])
        assert torch.equal(again, outs[True])  # integer accumulation: bit-reproducible
    finally:
        PR._ChannelKernel.use_mfma = True
        PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.mfma_variant = old_min, old_variant
    is_ring = ch._kernel.last_kernel.endswith("_ring")
    assert is_ring == (variant == "ring")  # contiguous slots for D % 4 == 0, D <= 256; row-staged slots for C1 / C5
    # int32 sums: the tap unit is ~ sum|g| * sqrt(2) / 2^24 for every filter, so the error grows with sqrt(L)
    grow = max(1.0, float(np.sqrt(len(taps) / 6401.0)))
    tol_rms, tol_max = (1.4e-5 * grow, 1e-4 * grow) if is_ring else (4e-6, 2e-5)
    valu, mfma = outs[False].cpu().numpy(), outs[True].cpu().numpy()
    assert valu.shape == mfma.shape == (-(-n // d),)
    n_cpu = 1_000_000
    want = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw[: 2 * n_cpu], "s16"), O.NcoState(f_off, fs), 1),
                                     O.OverlapSaveState(taps, 65536)), O.DecimState(d))
    k = want.size
    assert rms(valu[:k] - want) < 2e-7
    assert rms(mfma[:k] - want) < tol_rms and np.abs(mfma[:k] - want).max() < tol_max
    assert rms(mfma - valu) < tol_rms and np.abs(mfma - valu).max() < tol_max


@pytest.mark.parametrize("mode,fmt", [("nfm", "s16"), ("am", "s16"), ("usb", "s16"), ("nfm", "u8"), ("nfm", "f32")])
def test_pipeline_multi_block_streaming(A, tmp_path, mode, fmt):
    """Several device blocks (history, decimator phase, IIR states and AGC restarts carried across blocks,
    pinned double-buffered staging) on raw captures of every sample format, against the oracle."""
    from iq_to_audio_amd.benchmark import synthetic_iq_s16

    fs, f_off, secs = 2.5e6, 25e3, 2.2
    s16 = synthetic_iq_s16(fs, secs, f_off).reshape(-1)
    if fmt == "s16":
        raw, suffix = s16, ".cs16"
    elif fmt == "u8":
        raw, suffix = ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8), ".cu8"
    else:
        raw, suffix = (s16.astype(np.float32) / np.float32(32768.0)), ".cf32"
    path = tmp_path / f"cap_400000000Hz{suffix}"
    path.write_bytes(raw.tobytes())
    agc = mode != "usb"  # keep the SSB case on the well-conditioned (AGC off) path
    cfg = A.ProcessingConfig(in_path=path, target_freq=400_025_000.0, demod_mode=mode, agc_enabled=agc,
                             input_sample_rate=fs, output_path=tmp_path / "o.wav")
    pipe = A.ProcessingPipeline(cfg)
    pipe.block_frames_target = 2 * 1_048_576  # 3 blocks of 2 chunks for 5.5 M frames
    pipe.keep_channel_audio = True
    res = pipe.run()
    want = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, demod_mode=mode, agc_enabled=agc, fmt=fmt, keep_decimated=False)
    got = pipe.audio_fs_channel.cpu().numpy()
    assert got.size == want.audio.size and res.mix_sign == want.mix_sign
    assert rms(got - want.audio) < 2e-5
    assert abs(res.audio_peak - want.audio_peak) < 1e-4 * max(1.0, want.audio_peak)
    assert len(pipe.chunk_rms_dbfs) == len(want.rms_dbfs)
    np.testing.assert_allclose(pipe.chunk_rms_dbfs, want.rms_dbfs, atol=1e-2)


def test_run_benchmark_contract(A, caplog):
    """reference tests/test_benchmark.py:56-70: run_benchmark returns 0 and logs the 'x realtime' line."""
    import logging

    from iq_to_audio_amd.benchmark import run_benchmark

    with caplog.at_level(logging.INFO):
        rc = run_benchmark(seconds=0.5, sample_rate=2.5e6, freq_offset=25e3, center_freq=None, target_freq=None,
                           base_kwargs={"demod_mode": "nfm", "bandwidth": 12_500.0})
    assert rc == 0
    assert any("Benchmark processed" in r.message and "realtime" in r.message for r in caplog.records)
    with pytest.raises(ValueError):
        run_benchmark(seconds=0.0, sample_rate=2.5e6, freq_offset=25e3, center_freq=None, target_freq=None, base_kwargs=None)
    with pytest.raises(ValueError):
        run_benchmark(seconds=1.0, sample_rate=2.5e6, freq_offset=2e6, center_freq=None, target_freq=None, base_kwargs=None)


@pytest.mark.gpu
@pytest.mark.parametrize(
    "fs,d,bw,n",
    [
        (10e6, 104, 12500.0, 8_000_000),   # C2 shape: 7 k steps, ring of 3 rounds
        (10e6, 104, 12500.0, 5_000_123),   # ragged: odd tile counts, short last block
        (20e6, 208, 12500.0, 12_000_000),  # C3/C4 shape: 13 k steps, ring of 2 rounds
        (20e6, 208, 2800.0, 14_000_000),   # 32769 taps: three q-groups chained through partial sums
        (2.0e6, 20, 12500.0, 2_000_000),   # small decimation: 2 k steps
    ],
)
@pytest.mark.parametrize("acc32", [False, True])
def test_mfma_ring_kernel_matches_the_per_lane_kernel(A, fs, d, bw, n, acc32):
    """The ring kernel (contiguous LDS-DMA stream, tap fragments in registers) against the per-lane kernel and the
    float32 VALU kernel.  Default sums: one int64 (S1 << 32) + S2 per component with the per-lane kernel's own 16-bit
    fragments -- the same integers, so the two agree to float32 rounding of the rotation.  ``ring_acc32``: 256*S1 + S2 in
    one int32 with taps quantised to the ~14 bits that make that overflow-free -- agreement at that quantisation floor.
    Either way bit-reproducible run to run (exact integer accumulation)."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    f_off = 0.113 * fs
    raw = D.to_device(O.synth_capture_s16(fs, n / fs, 25e3).reshape(-1), "int16")
    taps = A.design_channel_filter(fs, bw, d)
    old = (PR._ChannelKernel.mfma_variant, PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma)
    old_acc = PR._ChannelKernel.ring_acc32
    outs = {}
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        PR._ChannelKernel.ring_acc32 = acc32
        for v in ("plain", "ring", "ring2", "valu"):
            PR._ChannelKernel.mfma_variant = "plain" if v == "valu" else v.rstrip("2")
            PR._ChannelKernel.use_mfma = v != "valu"
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
            outs[v] = ch.process(raw)
            want = {"plain": "k_channelize_mfma_s16", "ring": "k_channelize_mfma_s16_ring",
                    "ring2": "k_channelize_mfma_s16_ring", "valu": "k_channelize_v1"}[v]
            assert ch._kernel.last_kernel == want
    finally:
        PR._ChannelKernel.mfma_variant, PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma = old
        PR._ChannelKernel.ring_acc32 = old_acc
    assert torch.equal(outs["ring"], outs["ring2"])
    ring = outs["ring"].cpu().numpy()
    assert ring.shape == (-(-n // d),)
    for other in ("plain", "valu"):
        ref = outs[other].cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1.0)
        if acc32:
            tol = (2e-5, 1e-4)
        elif other == "valu":
            tol = (4e-6, 2e-5)
        else:  # same integers as the per-lane kernel; only the few tail outputs the ring leaves to the VALU kernel differ
            tol = (1e-7, 2e-5)
        assert rms(ring - ref) < tol[0] * scale and np.abs(ring - ref).max() < tol[1] * scale, other


@pytest.mark.parametrize("precision", ["fast", "full"])
@pytest.mark.parametrize("d", [16 * ks for ks in range(1, 17)] + [16 * ks - 3 for ks in (1, 2, 3, 5, 7, 8, 9, 10, 11)] + [100, 212])
def test_ring_kernel_every_kstep_count(A, d, precision):
    """Every instantiation of the ring kernel once: contiguous slots at 1..16 k steps (D = 16 KS: byte planes staged by
    loader waves up to 13 k steps -- tables of offsets up to 8, packed offsets beyond, uneven piece split at 12 and 13 --,
    LDS-DMA by the multiplying waves at 14..16), row-staged slots at 1..11 k steps (odd D: dword-aligned rows), a row that
    does not fill its last k step (D = 100, 212); 32-bit sums ("fast") and 64-bit sums ("full": the variants that spill a
    few registers at 12-13 k steps).  Against the float32 VALU kernel on a ragged capture cut into two blocks."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    fs = 96_000.0 * d
    n = 4500 * d + 1237
    f_off = 0.0917 * fs
    raw = D.to_device(O.synth_capture_s16(fs, n / fs, f_off).reshape(-1), "int16")
    taps = A.design_channel_filter(fs, 12_500.0, d)
    old = (PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma)
    outs = {}
    try:
        PR._ChannelKernel.mfma_min_outputs = 1024
        for use in (False, True):
            PR._ChannelKernel.use_mfma = use
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, precision=precision)
            cut = 2 * (2000 * d + 77)
            outs[use] = torch.cat([ch.process(raw[:cut]), ch.process(raw[cut:])]).cpu().numpy()
            if use:
                assert ch._kernel.last_kernel.startswith("k_channelize_mfma_s16"), ch._kernel.last_kernel
    finally:
        PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma = old
    valu, mfma = outs[False], outs[True]
    assert valu.shape == mfma.shape == (-(-n // d),)
    assert float(np.abs(valu).max()) > 0.05  # (the carrier is in the channel)
    # of full scale: ~14-bit taps for "fast"; the VALU kernel's own float32 rounding bounds what "full" can be held to
    tol = (2e-5, 1.2e-4) if precision == "fast" else (5e-7, 4e-6)
    assert rms(mfma - valu) < tol[0] and np.abs(mfma - valu).max() < tol[1], (rms(mfma - valu), np.abs(mfma - valu).max())


@pytest.mark.parametrize("fs,d,bw,n", [(2.4e6, 25, 12500.0, 6_000_000), (10e6, 104, 12500.0, 9_000_000), (50e6, 521, 12500.0, 16_000_000)])
def test_mfma_ring_kernel_uint8_captures(A, fs, d, bw, n):
    """uint8 I/Q captures (cu8, RTL-SDR style: 2.4 MS/s -> D = 25, an odd row of 50 bytes) on the matrix cores: the
    row-staged ring kernel with one data piece (u ^ 0x80, two MFMAs per k step) against the float32 VALU kernel and the
    oracle.  Also D = 104 and the three-pass D = 521 shape.  Bit-reproducible; error at the tap-quantisation floor."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    f_off = 25e3
    s16 = O.synth_capture_s16(fs, n / fs, f_off).reshape(-1)
    raw = ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)
    taps = A.design_channel_filter(fs, bw, d)
    x = D.to_device(raw, "uint8")
    old_min, old_use = PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma
    outs = {}
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        for use in (False, True, True):
            PR._ChannelKernel.use_mfma = use
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="u8")
            cut = 2 * (n // 3 + 1)
            z = torch.cat([ch.process(x[:cut]), ch.process(x[cut:])])  # second block: history + decimator phase
            assert ch._kernel.last_kernel == ("k_channelize_mfma_u8_ring" if use else "k_channelize_v1")
            outs.setdefault(use, []).append(z)
    finally:
        PR._ChannelKernel.mfma_min_outputs, PR._ChannelKernel.use_mfma = old_min, old_use
    assert torch.equal(outs[True][0], outs[True][1])
    valu, mfma = outs[False][0].cpu().numpy(), outs[True][0].cpu().numpy()
    assert valu.shape == mfma.shape == (-(-n // d),)
    n_cpu = min(n, 1_500_000)
    want = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw[: 2 * n_cpu], "u8"), O.NcoState(f_off, fs), 1),
                                     O.OverlapSaveState(taps, 65536)), O.DecimState(d))
    k = want.size
    grow = max(1.0, float(np.sqrt(len(taps) / 6401.0)))
    assert rms(valu[:k] - want) < 2e-7
    assert rms(mfma[:k] - want) < 1.4e-5 * grow and np.abs(mfma[:k] - want).max() < 1e-4 * grow
    assert rms(mfma - valu) < 1.4e-5 * grow and np.abs(mfma - valu).max() < 1e-4 * grow


def test_multi_channel_single_pass_and_cli(A, tmp_path):
    """BASELINE config 3 pattern: several --ft targets (mixed demodulators, bandwidths) extracted from ONE pass
    over the capture; every channel must equal its own single-target oracle run, and the CLI shim must
    produce one 48 kHz WAV per target."""
    from iq_to_audio_amd import cli, iqio
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    fs, secs, fc = 2.5e6, 1.3, 433.0e6
    carriers = [(25e3, 0.14, "nfm"), (-150e3, 0.14, "am"), (400e3, 0.14, "usb"), (-610e3, 0.14, "nfm")]
    raw = synthetic_multi_iq_s16(fs, secs, carriers)
    wav = tmp_path / "multi_433000000Hz.wav"
    iqio.write_wav_iq(wav, raw, int(fs), "s16")
    specs = [("nfm", 12_500.0, True), ("am", 10_000.0, True), ("usb", 2_800.0, False), ("nfm", 12_500.0, True)]
    cfgs = [A.ProcessingConfig(in_path=wav, target_freq=fc + off, bandwidth=bw, demod_mode=mode, agc_enabled=agc,
                               output_path=tmp_path / f"ch{i}.wav")
            for i, ((off, _, _), (mode, bw, agc)) in enumerate(zip(carriers, specs))]
    multi = A.MultiChannelPipeline(cfgs)
    for o in multi.owners:
        o.keep_channel_audio = True
        o.block_frames_target = 1_048_576  # several blocks
    results = multi.run()
    assert len(results) == 4
    for (off, _, _), (mode, bw, agc), res, owner in zip(carriers, specs, results, multi.owners):
        want = O.run_chain(raw, sample_rate=fs, freq_offset=off, bandwidth=bw, demod_mode=mode, agc_enabled=agc,
                           keep_decimated=False)
        got = owner.audio_fs_channel.cpu().numpy()
        assert got.size == want.audio.size and res.mix_sign == want.mix_sign == 1
        assert abs(res.freq_offset - off) < 1e-6
        assert rms(got - want.audio) < 2e-5, (mode, off)
        assert rms(want.audio) > 1e-3  # the channel really carries its signal
    for i in range(4):
        assert (tmp_path / f"ch{i}.wav").stat().st_size > 100_000
    # mixed inputs are rejected
    other = tmp_path / "other.wav"
    iqio.write_wav_iq(other, raw[:1000], int(fs), "s16")
    with pytest.raises(ValueError):
        A.MultiChannelPipeline([cfgs[0], A.ProcessingConfig(in_path=other, target_freq=fc)])
    # the CLI shim: three targets, explicit --out gets the _<freq> suffix (reference cli.py:523-527)
    out = tmp_path / "cli" / "a.wav"
    rc = cli.main(["--in", str(wav), "--ft", str(fc + 25e3), "--ft", str(fc - 150e3), "--ft", str(fc - 610e3),
                   "--out", str(out)])
    assert rc == 0
    for f in (fc + 25e3, fc - 150e3, fc - 610e3):
        pcm, rate = iqio.read_wav_pcm16_mono(out.with_name(f"a_{int(round(f))}.wav"))
        assert rate == 48000 and pcm.size == -(-(-(-raw.shape[0] // 26)) * 24000 // 48077)
    assert cli.main(["--in", str(tmp_path / "missing.wav"), "--ft", "1e6", "--fc", "1e6"]) == 1


def test_spectrum_psd_and_waterfall_vs_reference_fixtures_and_oracle(A, golden):
    """iq_to_audio_amd.spectrum (window + rocFFT double-complex FFT + dB/fftshift epilogue,
    pairwise waterfall reduction on the device) against the reference's own outputs (tests/golden/spectrum.npz) and,
    at the reference's default nfft = 2^18, against the oracle.  PSD values are compared in dB where the bin is
    above the float64 FFT's own noise (1e-13 of the frame's peak power) and in linear power below it."""
    from iq_to_audio_amd import spectrum as S
    from test_oracle_golden import _spectrum_stream, spectrum_waterfall_chunks

    def close(db, want):
        db, want = np.asarray(db, dtype=np.float64), np.asarray(want, dtype=np.float64)
        lin, wlin = 10.0 ** (db / 10.0), 10.0 ** (want / 10.0)
        floor = 1e-13 * wlin.max()
        strong = wlin > 1e3 * floor
        assert np.max(np.abs(db[strong] - want[strong])) < 1e-6
        assert np.max(np.abs(lin - wlin)) < 10 * floor + 1e-18

    g = golden("spectrum.npz")
    for k in range(4):
        seed, n, nfft, fs = g[f"psd{k}_case"]
        freqs, psd = S.compute_psd(_spectrum_stream(int(seed), int(n)), float(fs), int(nfft))
        np.testing.assert_array_equal(freqs, g[f"psd{k}_freqs"])
        assert psd.dtype == np.float64
        close(psd, g[f"psd{k}_db"])
    for k in range(3):
        nfft, hop, max_slices = (int(v) for v in g[f"wf{k}_case"])
        freqs, avg, wf, frames = S.streaming_waterfall(spectrum_waterfall_chunks(g), 2.0e6, nfft=nfft,
                                                       hop=None if hop < 0 else hop, max_slices=max_slices)
        assert frames == int(g[f"wf{k}_frames"])
        np.testing.assert_array_equal(freqs, g[f"wf{k}_freqs"])
        np.testing.assert_array_equal(wf.times, g[f"wf{k}_times"])
        assert wf.matrix.dtype == np.float32 and wf.matrix.shape == g[f"wf{k}_matrix"].shape
        # rows are float32 dB values; the noise-floor bins differ by the FFT's rounding, compare where it is resolved
        want = g[f"wf{k}_matrix"].astype(np.float64)
        resolved = want > want.max() - 120.0
        assert np.max(np.abs(wf.matrix.astype(np.float64)[resolved] - want[resolved])) < 2e-4
        assert np.max(np.abs(avg - g[f"wf{k}_avg"])[g[f"wf{k}_avg"] > g[f"wf{k}_avg"].max() - 120.0]) < 1e-5
    # the reference's default frame size on the benchmark capture: 2^18-point double-complex FFT
    raw = O.synth_capture_s16(2.5e6, 0.5, 25e3)
    x = O.ingest_to_complex64(raw.reshape(-1), "s16")
    freqs, psd = S.compute_psd(x, 2.5e6)
    wf_freqs, want = O.compute_psd(x, 2.5e6)
    np.testing.assert_array_equal(freqs, wf_freqs)
    close(psd, want)
    assert abs(freqs[int(np.argmax(psd))] - 25e3) < 2.5e6 / (1 << 18) * 1.5
    _, avg, wf, frames = S.streaming_waterfall([x[:400_000], x[400_000:]], 2.5e6, nfft=1 << 16, max_slices=40)
    _, avg_o, times_o, matrix_o, frames_o = O.streaming_waterfall([x[:400_000], x[400_000:]], 2.5e6, nfft=1 << 16, max_slices=40)
    assert frames == frames_o and wf.matrix.shape == matrix_o.shape
    np.testing.assert_array_equal(wf.times, times_o)
    resolved = matrix_o > matrix_o.max() - 110.0
    assert np.max(np.abs(wf.matrix[resolved] - matrix_o[resolved])) < 2e-4
    with pytest.raises(ValueError):
        S.compute_psd(np.empty(0, dtype=np.complex64), 1e6)
    with pytest.raises(ValueError):
        S.streaming_waterfall([np.zeros(10, dtype=np.complex64)], 1e6, nfft=64)
