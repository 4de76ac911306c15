"""GPU parity tests for the BASELINE configurations as WORKLOADS (not just their filter shapes):

* config 3 -- 20 MS/s, five simultaneous targets nfm/am/usb/lsb/nfm with bandwidths 12.5k/10k/2.8k/2.8k/12.5k,
  AGC on -- through :class:`MultiChannelPipeline` against one oracle chain per target;
* config 5's per-GPU unit -- 50 MS/s, D = 521, five NFM channels with de-emphasis out of a 40-carrier capture --
  through channelizer + demodulator + 48 kHz resampler against the oracle;
* SSB with AGC on, where the reference itself is ill-conditioned: the error is localised BY CAUSE instead of being
  bounded loosely (see :func:`ssb_agc_evidence`).

Bars (BASELINE.json north_star): sample counts exact; float audio within 1e-4 RMS.  Asserted tolerances are stated
per test.
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


# ---- SSB + AGC: where the error lives -------------------------------------------------------------


def oracle_demod_chunks(z, chunk_lens, mode, fs_ch, agc=True, clip=True):
    """The oracle's decoder + writer clip over the given per-chunk lengths (AGC gain restarts per chunk,
    reference decoders/ssb.py:72; DC-blocker state carried, decoders/common.py:28-29)."""
    st = O.DemodState(mode, fs_ch, agc_enabled=agc)
    out, pos = [], 0
    for n in chunk_lens:
        y, _ = O.demodulate(z[pos : pos + n], st)
        out.append(np.clip(y, -0.99, 0.99) if clip else y)
        pos += int(n)
    return np.concatenate(out) if out else np.empty(0, np.float32)


def agc_sensitivity(z, chunk_lens, mode, fs_ch, noise_rms, seed=123):
    """kappa(noise_rms): how far the ORACLE's own clipped SSB+AGC output moves (RMS) when its input z moves by white
    noise of the given RMS.  The reference's AGC adds 0.001*(target/|s| - gain) per sample for |s| down to 1e-6
    (decoders/ssb.py:75-77): a perturbation d of a sample near a zero crossing changes the gain by
    ~2.5e-4 * d / s^2 and that change decays with a 1000-sample time constant."""
    rng = np.random.default_rng(seed)
    dz = (rng.normal(size=z.size) + 1j * rng.normal(size=z.size)) * (noise_rms / np.sqrt(2))
    a = oracle_demod_chunks(z, chunk_lens, mode, fs_ch)
    b = oracle_demod_chunks((z + dz).astype(np.complex64), chunk_lens, mode, fs_ch)
    return rms(a - b)


def oracle_ssb_f64_chunks(z, chunk_lens, mode, agc=True):
    """The SSB decoder's two recurrences in float64 (oracle.cpu_ref.ssb_demod_f64 -- NOT the reference's float32
    arithmetic), per-chunk gain restarts, clipped like the writer: the comparison target for LOGIC."""
    st, out, pos = O.DcState(), [], 0
    for n in chunk_lens:
        out.append(np.clip(O.ssb_demod_f64(z[pos : pos + n], st, lsb=(mode == "lsb"), agc_enabled=agc), -0.99, 0.99))
        pos += int(n)
    return np.concatenate(out) if out else np.empty(0, np.float32)


def ssb_agc_evidence(label, z_gpu, audio_gpu, z_ref, audio_ref, chunk_lens, mode, fs_ch, z_tol, strict_replay=False, e2e_max=None):
    """SSB with AGC on: localise the GPU-vs-oracle difference BY CAUSE.

    It cannot be localised in TIME: the gain carries every near-zero sample's error for ~1000 samples, so an input
    perturbation of 1e-7 RMS already moves the reference's own output by more than 1e-4 on about half of all samples
    (--benchmark capture: 54 % within 1e-4 at 1e-7, 49 % at 3e-7; tests/test_host_logic.py).  The chain is therefore
    held link by link:

      1. z (channelizer output) against the oracle's z: the usual tight bar ``z_tol`` -- the only place where the two
         implementations see different numbers;
      2. LOGIC: the decoder's two recurrences restated in float64 (``oracle_ssb_f64_chunks``), fed with the GPU's own
         z, must reproduce the GPU's audio to 1e-5 on EVERY sample (same input, same precision: what is left is the
         order of float64 operations).  A wrong AGC restart index, a wrong state hand-off between blocks or a wrong
         threshold decision shows up here at the 1e-2 level;
      3. ROUNDING: the reference runs those recurrences in sequential float32; replaying the GPU's z through THAT
         arithmetic differs from the GPU's audio by the reference's own rounding amplified by 1/|s| -- measured and
         printed; with ``strict_replay`` (the --benchmark capture, on which the north-star bar is stated) it must meet
         the bar itself: >= 99 % of samples within 1e-4, median <= 1e-6, RMS < 1e-4;
      4. the end-to-end error is then the reference's sensitivity to the z difference of link 1: it must stay below
         2 x kappa evaluated AT that difference (never below the 3e-7 of float32 rounding; kappa is a one-realisation
         Monte-Carlo estimate, the GPU's z difference is another realisation of the same size: measured ratio 0.05 .. 1.54).
    Returns the measured numbers (DESIGN.md section 5 quotes them)."""
    assert z_gpu.shape == z_ref.shape and audio_gpu.shape == audio_ref.shape == (int(np.sum(chunk_lens)),)
    dz = rms(z_gpu - z_ref)
    assert dz < z_tol, (label, "z", dz)
    logic = np.abs(audio_gpu.astype(np.float64) - oracle_ssb_f64_chunks(z_gpu, chunk_lens, mode))
    replay = oracle_demod_chunks(z_gpu, chunk_lens, mode, fs_ch)
    e = np.abs(audio_gpu.astype(np.float64) - replay)
    frac, med, e_rms = float(np.mean(e < 1e-4)), float(np.median(e)), float(np.sqrt(np.mean(e * e)))
    err = rms(audio_gpu - audio_ref)
    kappa = agc_sensitivity(z_ref, chunk_lens, mode, fs_ch, max(dz, 3e-7))
    print(f"SSB+AGC {label}: z rms diff {dz:.2e} | logic (float64 recurrences on the GPU's z): max {logic.max():.2e}, "
          f"rms {np.sqrt(np.mean(logic ** 2)):.2e} | rounding (reference float32 loops on the GPU's z): rms {e_rms:.2e}, "
          f"median {med:.2e}, within 1e-4: {100 * frac:.3f} %, max {e.max():.2e} | end-to-end rms {err:.2e} vs "
          f"kappa({max(dz, 3e-7):.1e}) = {kappa:.2e}")
    assert logic.max() < 1e-5, (label, "logic", float(logic.max()))
    if strict_replay:
        assert frac >= 0.99, (label, "replay fraction", frac)
        assert med <= 1e-6, (label, "replay median", med)
        assert e_rms < 1e-4, (label, "replay rms", e_rms)
    else:
        assert e_rms < 1e-3 and frac > 0.9, (label, "replay", e_rms, frac)
    assert err < 2.0 * kappa + 2e-5, (label, "end-to-end", err, kappa)
    if e2e_max is not None:  # what is measured at the product's precision for this case, with margin
        assert err < e2e_max, (label, "end-to-end (absolute)", err, e2e_max)
    return dict(dz=dz, logic_max=float(logic.max()), replay_rms=e_rms, replay_median=med, replay_frac=frac, err=err, kappa=kappa)


def chunk_lens_for(n_frames, chunk, d, n_dec):
    starts = -(-np.arange(0, n_frames, chunk, dtype=np.int64) // d)
    return np.diff(np.append(starts, n_dec))


def test_ssb_agc_same_input_full_c1(A, golden):
    """BASELINE config 1 at full size, USB and LSB with AGC on, through the fused block path (Channelizer +
    ChannelDemod, two device blocks so that the DC-blocker state crosses a block edge and the AGC restarts fall both
    inside and at the start of a block).  Held by cause (``ssb_agc_evidence``), plus the gain trajectory: the
    pluggable decoder (unclipped output) fed with the oracle's own z chunk by chunk must end every chunk with the
    oracle's gain."""
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.processing import ChannelDemod

    g = golden("c1_full_scalars.npz")
    fs, f_off, chunk = float(g["fs"]), float(g["f_off"]), int(g["chunk"])
    raw = O.synth_capture_s16(fs, float(g["seconds"]), f_off)
    n = raw.shape[0]
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    flat = raw.reshape(-1)
    for mode in ("usb", "lsb"):
        want = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, demod_mode=mode, keep_decimated=True)
        # SSB with the AGC on runs at the "full" precision (processing.base_precision): z within float32 rounding of the oracle's
        from iq_to_audio_amd.processing import base_precision

        assert base_precision(mode, True) == "full" and base_precision(mode, False) == "fast" and base_precision("nfm", True) == "fast"
        ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, precision=base_precision(mode, True))
        assert ch.precision == "full"
        dem = ChannelDemod(mode, fs_ch, deemph_us=300.0, agc_enabled=True)
        n_dec = -(-n // d)
        audio, zs, pos = D.empty(n_dec, "float32"), [], 0
        step = 7 * chunk  # 7 + 5 chunks
        for lo in range(0, n, step):
            hi = min(lo + step, n)
            z = ch.process(D.to_device(flat[2 * lo : 2 * hi], "int16"))
            dem.process(z, P.chunk_output_starts(chunk, d, lo, hi - lo), audio[pos : pos + z.numel()])
            zs.append(z)
            pos += z.numel()
        import torch

        z_gpu = torch.cat(zs).cpu().numpy()
        got = audio.cpu().numpy()
        assert got.size == want.audio.size == int(g[mode + "_n"]) == 480_770
        lens = chunk_lens_for(n, chunk, d, n_dec)
        assert len(lens) == 12
        ev = ssb_agc_evidence(f"C1 {mode}", z_gpu, got, want.decimated, want.audio, lens, mode, fs_ch, z_tol=2e-7,
                              strict_replay=True, e2e_max=1.5e-2)  # (6.5e-3 measured; 3.4e-2 at "fast")
        assert ev["dz"] < 2e-7  # ("full": the per-lane kernel, taps + their residue as chained passes; "fast" measured 2.0e-6)
        assert ch._kernel.last_kernel == "k_channelize_mfma_s16"
        assert abs(rms(got) - float(g[mode + "_rms"])) < 0.01 * float(g[mode + "_rms"])
        np.testing.assert_allclose(dem.chunk_rms_dbfs(), want.rms_dbfs, atol=0.5)
        # gain trajectory: same input (the ORACLE's z), unclipped outputs of the pluggable decoder, every chunk.
        # Against the float64 recurrences the gain at each chunk's end must agree to 1e-5 (logic); against the
        # reference's float32 loops the unclipped audio must agree within 1e-4 (relative to the output's own excursion,
        # peak 42 on this capture) on >= 99 % of the samples (rounding).
        dec = A.create_decoder(mode, deemph_us=300.0, agc_enabled=True)
        dec.setup(fs_ch)
        st32, st64 = O.DemodState(mode, fs_ch), O.DcState()
        pos, close = 0, []
        for k, ln in enumerate(lens):
            zc = want.decimated[pos : pos + ln]
            y_gpu, _ = dec.process(zc)
            y_ref, _ = O.demodulate(zc, st32)
            y_64 = O.ssb_demod_f64(zc, st64, lsb=(mode == "lsb"))
            s_gpu = dec.intermediates()["dc_block"][0]
            tail = slice(ln - 64, ln)
            ok = np.abs(s_gpu[tail]) > 1e-4
            gain_gpu = np.median(y_gpu[tail][ok].astype(np.float64) / s_gpu[tail][ok])
            gain_64 = np.median(y_64[tail][ok].astype(np.float64) / s_gpu[tail][ok])
            assert abs(gain_gpu - gain_64) <= 1e-5 * abs(gain_64), (mode, k, gain_gpu, gain_64)
            assert np.abs(y_gpu.astype(np.float64) - y_64).max() <= 1e-5 * max(1.0, float(np.abs(y_64).max())), (mode, k)
            e = np.abs(y_gpu.astype(np.float64) - y_ref)
            close.append(e < 1e-4 * np.maximum(1.0, np.abs(y_ref)))
            assert np.mean(close[-1]) >= 0.95, (mode, k)  # (the float64 statement itself: 97.1 % on the worst chunk)
            pos += int(ln)
        assert np.mean(np.concatenate(close)) >= 0.99, mode


# ---- BASELINE config 3 as a workload ---------------------------------------------------------------

C3_FS = 20e6
C3_TARGETS = [  # (offset Hz, amplitude, generator mode), (demod mode, bandwidth): SURVEY.md section 8(d)
    ((25e3, 0.14, "nfm"), ("nfm", 12_500.0)),
    ((-150e3, 0.14, "am"), ("am", 10_000.0)),
    ((400e3, 0.14, "usb"), ("usb", 2_800.0)),
    ((-1.1e6, 0.14, "lsb"), ("lsb", 2_800.0)),
    ((2.3e6, 0.14, "nfm"), ("nfm", 12_500.0)),
]


def test_config3_workload_five_mixed_targets_agc_on(A, tmp_path):
    """BASELINE config 3: 20 MS/s, five simultaneous --ft targets (NFM 12.5 k / AM 10 k / USB 2.8 k / LSB 2.8 k /
    NFM 12.5 k: 12 801 / 16 001 / 32 769 / 32 769 / 12 801 taps, i.e. 1 + 2 + 3 + 3 + 1 tap-row groups), AGC on,
    D = 208, chunk 4 194 304 -- 0.55 s of the build-defined five-carrier capture through
    MultiChannelPipeline in three device blocks, every channel against its own single-target oracle chain: exact
    counts, signs, NFM/AM audio < 2e-5 RMS, SSB+AGC held by cause, 48 kHz PCM16 of NFM/AM within 1 LSB."""
    from iq_to_audio_amd import iqio
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    fs, secs, fc = C3_FS, 0.55, 433.0e6
    raw = synthetic_multi_iq_s16(fs, secs, [c for c, _ in C3_TARGETS])
    n = raw.shape[0]
    wav = tmp_path / "c3_433000000Hz.wav"
    iqio.write_wav_iq(wav, raw, int(fs), "s16")
    cfgs = [A.ProcessingConfig(in_path=wav, target_freq=fc + off, bandwidth=bw, demod_mode=mode, agc_enabled=True,
                               output_path=tmp_path / f"c3_{i}.wav", dump_iq_path=tmp_path / f"c3_{i}.cf32")
            for i, ((off, _, _), (mode, bw)) in enumerate(C3_TARGETS)]
    multi = A.MultiChannelPipeline(cfgs)
    for o in multi.owners:
        o.keep_channel_audio = True
        o.block_frames_target = 4_194_304  # one reference chunk per device block: 3 blocks
    from iq_to_audio_amd import processing as PR

    old_min = PR._ChannelKernel.mfma_min_outputs
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096  # 20 165 outputs per block: matrix-core kernels, as on a 64 Mi-frame block
        results = multi.run()
    finally:
        PR._ChannelKernel.mfma_min_outputs = old_min
    assert len(results) == 5
    # NFM / AM: lanes of the shared ring launch with int32 sums ("fast"); USB / LSB with the AGC on: "full" precision -- every
    # tap-row group and the residue of its quantisation as lanes of a SECOND shared launch, with 64-bit sums (lane pairs)
    assert [o.channelizer_precision for o in multi.owners] == ["fast", "fast", "full", "full", "fast"]
    assert all(o.channelizer_kernel == "k_channelize_mfma_s16_ring" for o in multi.owners)
    assert multi.banks[0].launches == [dict(lanes=1 + 2 + 1, launches=1, combines=1, pairs=2), dict(lanes=2 * (3 + 3), launches=1, combines=2, pairs=6)]
    chunk = 4_194_304
    for i, (((off, _, _), (mode, bw)), res, owner) in enumerate(zip(C3_TARGETS, results, multi.owners)):
        want = O.run_chain(raw, sample_rate=fs, freq_offset=off, bandwidth=bw, demod_mode=mode, agc_enabled=True)
        got = owner.audio_fs_channel.cpu().numpy()
        assert res.decimation == want.decimation == 208 and want.chunk == chunk
        assert got.size == want.audio.size == -(-n // 208)  # sample count: exact
        assert res.mix_sign == want.mix_sign
        assert abs(res.freq_offset - off) < 1e-3
        assert want.ntaps == {12_500.0: 12_801, 10_000.0: 16_001, 2_800.0: 32_769}[bw]
        z_gpu = np.fromfile(tmp_path / f"c3_{i}.cf32", dtype=np.complex64)
        assert z_gpu.size == want.decimated.size
        if mode in ("usb", "lsb"):
            lens = chunk_lens_for(n, chunk, 208, got.size)
            ev = ssb_agc_evidence(f"C3 {mode} {off:+.0f} Hz", z_gpu, got, want.decimated, want.audio, lens, mode,
                                  want.fs_channel, z_tol=3e-7, e2e_max=5e-4)  # (1.1e-4 / 1.7e-4 measured; 1.8e-2 / 7.7e-2 at "fast")
            assert ev["dz"] < 3e-7, (mode, ev["dz"])  # ("fast" measured 3.2e-6)
            assert rms(want.audio) > 0.05
            continue
        assert rms(z_gpu - want.decimated) < 3e-5, (mode, off)
        err = rms(got - want.audio)
        print(f"C3 {mode} {off:+.0f} Hz: audio rms err {err:.2e} (signal rms {rms(want.audio):.3f})")
        assert err < 1e-4, (mode, off, err)  # the north-star bar
        assert err < 2e-5, (mode, off, err)  # what we hold
        assert rms(want.audio) > 1e-3  # the channel carries its signal
        assert abs(res.audio_peak - want.audio_peak) < 2e-4 * max(1.0, want.audio_peak)
        pcm, rate = iqio.read_wav_pcm16_mono(tmp_path / f"c3_{i}.wav")
        ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
        assert rate == 48000 and pcm.size == ref48.size
        assert np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32))) <= 1


# ---- BASELINE config 5's per-GPU unit --------------------------------------------------------------

C5_FS = 50e6


def c5_carriers():
    """40 NFM carriers on a 100 kHz raster from -1.95 MHz, amplitude 0.02 each."""
    return [(-1.95e6 + 100e3 * k, 0.02, "nfm") for k in range(40)]


def test_config5_unit_five_nfm_channels(A, tmp_path):
    """BASELINE config 5, one GPU's share: 50 MS/s, D = 521 (fs_ch 95 969.29 Hz, 32 001 taps, 33 k steps in three
    passes), five of the 40 NFM channels (first, second, the two around DC, last) with de-emphasis, 0.42 s, through
    MultiChannelPipeline (demodulator + 48 kHz resampler with up/down = 48 000/95 969) against the oracle.
    The carriers are weak (0.02 of full scale, 34 dB below the wideband total): this is also the weak-channel case
    for the fixed-point channelizer."""
    from iq_to_audio_amd import iqio
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    fs, secs, fc = C5_FS, 0.42, 1.0e9
    carriers = c5_carriers()
    raw = synthetic_multi_iq_s16(fs, secs, carriers)
    n = raw.shape[0]
    path = tmp_path / "c5_1000000000Hz.cs16"
    path.write_bytes(raw.tobytes())
    picks = [0, 1, 19, 20, 39]
    cfgs = [A.ProcessingConfig(in_path=path, target_freq=fc + carriers[k][0], bandwidth=12_500.0, demod_mode="nfm",
                               deemph_us=300.0, input_sample_rate=fs, output_path=tmp_path / f"c5_{k}.wav",
                               dump_iq_path=tmp_path / f"c5_{k}.cf32") for k in picks]
    multi = A.MultiChannelPipeline(cfgs)
    for o in multi.owners:
        o.keep_channel_audio = True
        o.block_frames_target = 3 * 4_194_304  # two device blocks (3 + 3 chunks; 21 M frames = 5.01 chunks)
    from iq_to_audio_amd import processing as PR

    old_min = PR._ChannelKernel.mfma_min_outputs
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096  # 24 151 outputs per block: the row-staged ring kernel, three k-step passes
        results = multi.run()
    finally:
        PR._ChannelKernel.mfma_min_outputs = old_min
    assert all(o.channelizer_kernel == "k_channelize_mfma_s16_ring" for o in multi.owners)
    problems = []  # tolerance misses are collected so that one run shows every channel's numbers
    for k, res, owner in zip(picks, results, multi.owners):
        off = carriers[k][0]
        want = O.run_chain(raw, sample_rate=fs, freq_offset=off, bandwidth=12_500.0, demod_mode="nfm")
        assert (res.decimation, want.decimation, want.ntaps) == (521, 521, 32_001)
        assert abs(res.fs_channel - 50e6 / 521) < 1e-9
        got = owner.audio_fs_channel.cpu().numpy()
        assert got.size == want.audio.size == -(-n // 521)
        assert res.mix_sign == want.mix_sign  # (the raster is symmetric: the mirror carrier has the same power)
        z_gpu = np.fromfile(tmp_path / f"c5_{k}.cf32", dtype=np.complex64)
        dz = rms(z_gpu - want.decimated)
        err = rms(got - want.audio)
        print(f"C5 channel {k} ({off:+.0f} Hz): z rms diff {dz:.2e} (|z| rms {rms(want.decimated):.3e}), "
              f"audio rms err {err:.2e} (signal rms {rms(want.audio):.3f})")
        if not dz < 1e-5:  # (5.0e-6 measured: the row-staged ring kernel, ~14-bit taps, 32 001 of them)
            problems.append((k, "z", dz))
        if not err < 1e-4:  # the north-star bar on a channel 34 dB below full scale
            problems.append((k, "audio", err))
        assert rms(want.audio) > 1e-2
        pcm, rate = iqio.read_wav_pcm16_mono(tmp_path / f"c5_{k}.wav")
        ref48 = O.resample_48k(want.audio, want.fs_channel)
        assert rate == 48000 and pcm.size == ref48.size == -(-got.size * 48000 // 95969)
        # PCM16 of the GPU's own audio through the oracle's resampler: the resampler/quantiser link, 1 LSB
        own48 = O.float_to_pcm16(O.resample_48k(got, want.fs_channel))
        assert np.max(np.abs(pcm.astype(np.int32) - own48.astype(np.int32))) <= 1
        assert rms(pcm.astype(np.float64) / 32768.0 - ref48) < 1e-4 + (1.0 if problems else 0.0)
    assert not problems, problems


# ---- shared ingest: several channels in one launch ------------------------------------------------


@pytest.mark.parametrize("fs,fmt,specs,n", [
    # C3: D = 208, contiguous slots, 1 + 2 + 3 + 3 + 1 tap-row groups = 10 lanes, three filters finished by the combine kernel
    (20e6, "s16", [(25e3, 12_500.0, 1), (-150e3, 10_000.0, 1), (400e3, 2_800.0, 1), (-1.1e6, 2_800.0, -1), (2.3e6, 12_500.0, 1)], 12_000_000),
    # D = 208, lane pairs across tap-row groups: 1 + 3 + 2 groups = 6 lanes in descending group order (2,1), (1,0), (0,0)
    (20e6, "s16", [(25e3, 12_500.0, 1), (400e3, 2_800.0, -1), (-150e3, 10_000.0, 1)], 9_000_000),
    # D = 208, an odd number of lanes: one pair and a workgroup whose second half idles
    (20e6, "s16", [(25e3, 12_500.0, 1), (-1.3e6, 12_500.0, -1), (2.3e6, 12_500.0, 1)], 9_000_000),
    # C2 rate: D = 104 (loader waves), three single-group lanes
    (10e6, "s16", [(25e3, 12_500.0, 1), (-1.3e6, 12_500.0, 1), (3.1e6, 12_500.0, -1)], 7_000_000),
    # C5's unit: D = 521, row-staged slots, five channels x three k-step passes chained through partial sums
    (50e6, "s16", [(-1.95e6, 12_500.0, 1), (-1.85e6, 12_500.0, 1), (-50e3, 12_500.0, 1), (50e3, 12_500.0, -1), (1.95e6, 12_500.0, 1)], 22_000_000),
    # uint8 capture, odd decimation (RTL-SDR rate)
    (2.4e6, "u8", [(25e3, 12_500.0, 1), (-300e3, 12_500.0, 1)], 5_000_000),
])
def test_channel_bank_equals_one_channel_at_a_time(A, fs, fmt, specs, n):
    """ChannelBank (one launch of the ring kernel for all channels of a capture, iqa_channelize_mfma_multi +
    iqa_mfma_combine) against the same channelizers run one by one: the same integers are added in the same order, so
    the decimated streams must agree to the last bit of the float32 rotation -- over two ragged blocks (history,
    decimator phase) and with mixed mixer signs.  One channel is also held against the oracle so that the pair is not jointly wrong."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd import processing as PR

    d, _ = P.choose_decimation(fs, 96_000.0)
    s16 = O.synth_capture_s16(fs, n / fs, specs[0][0], seed=3).reshape(-1)
    raw = s16 if fmt == "s16" else ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)
    x = D.to_device(raw, "int16" if fmt == "s16" else "uint8")
    cut = 2 * (n // 2 + 12_345)
    old_min = PR._ChannelKernel.mfma_min_outputs
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096

        def make():
            return [A.Channelizer(A.design_channel_filter(fs, bw, d), sample_rate=fs, freq_offset=off, mix_sign=sign, decimation=d, fmt=fmt)
                    for off, bw, sign in specs]

        singles = [[c.process(x[:cut]), c.process(x[cut:])] for c in make()]
        bank = A.ChannelBank(make())
        first = bank.process(x[:cut])
        info = dict(bank.last_launch)
        banked = [[a_, b_] for a_, b_ in zip(first, bank.process(x[cut:]))]
    finally:
        PR._ChannelKernel.mfma_min_outputs = old_min
    groups = [max(1, -(-(-(-len(A.design_channel_filter(fs, bw, d)) // d)) // 64)) for _, bw, _ in specs]
    kranges = 1 if (d % 4 == 0 and d <= 256 and fmt == "s16") else -(-(-(-2 * d // 32)) // 11)
    # lanes of equal tap-row group go two to a workgroup where the kernel offers it (int16, contiguous slots, 9..16 k steps)
    ks = -(-2 * d // 32)
    can_pair = fmt == "s16" and d % 4 == 0 and 9 <= ks <= 16 and ks != 15
    pairs = -(-sum(groups) // 2) if can_pair else 0  # lanes in descending group order, two by two; one launch either way
    assert info == dict(lanes=sum(groups), launches=kranges, combines=sum(g > 1 for g in groups), pairs=pairs), info
    if fs == 20e6:
        assert pairs == {10: 5, 6: 3, 3: 2}[sum(groups)]
    # The matrix-core interior of a block starts 64 outputs per tap-row group behind the block's first output and ends
    # ~30 outputs before its last: a bank uses the interior common to its channels, so a channel with fewer groups gets
    # a few more of its first outputs from the float32 kernel than it would alone.  Inside: bit-identical.
    edge = 64 * max(groups) + 64 + 8192 // d
    for i, (one, many) in enumerate(zip(singles, banked)):
        assert sum(t.numel() for t in one) == sum(t.numel() for t in many) == -(-n // d)
        for blk, (a_, b_) in enumerate(zip(one, many)):
            assert a_.numel() == b_.numel()
            # same integer sums; the float64 rotation recurrence starts at each workgroup's first output and a bank cuts
            # the launch into other ranges than a single channel does: where cos/sin sit on a float32 rounding boundary
            # the last bit of an output may differ
            inner = (a_ - b_)[edge:-edge]
            assert float(inner.abs().max()) <= 1.2e-7 * float(a_.abs().max()), (i, blk, float(inner.abs().max()))
            assert float((inner != 0).float().mean()) < 1e-3, (i, blk)
            assert float((a_ - b_).abs().max()) < 2e-4  # the edges: float32 kernel vs fixed-point kernel
    banked = [torch.cat(pair) for pair in banked]
    off, bw, sign = specs[0]
    n_cpu = min(n, 1_500_000)
    taps = A.design_channel_filter(fs, bw, d)
    want = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw[: 2 * n_cpu], fmt), O.NcoState(off, fs), sign),
                                     O.OverlapSaveState(taps, 65536)), O.DecimState(d))
    got = banked[0].cpu().numpy()[: want.size]
    assert rms(got - want) < 1.4e-5 * max(1.0, float(np.sqrt(len(taps) / 6401.0)))
    assert rms(want) > 0.05


def test_resident_bank_runner_matches_the_oracle_per_target(A):
    """batch.ResidentBankRunner (what bench.py's config-3 entry and batch.demodulate_sharded run): four targets of one
    resident capture at the C2 rate -- two of them multi-group filters (2 and 5 tap-row groups), one whose carrier sits on the other side so that
    its probe overrules the speculative sign +1 -- every target's channel-rate audio and 48 kHz PCM16 against its own
    oracle chain; two captures back to back through the two buffer slots."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.batch import ResidentBankRunner
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    fs, secs = 10e6, 0.9
    targets = [dict(freq_offset=25e3, demod_mode="nfm", bandwidth=12_500.0), dict(freq_offset=-150e3, demod_mode="am", bandwidth=10_000.0),
               dict(freq_offset=400e3, demod_mode="usb", bandwidth=2_800.0, agc_enabled=False),
               dict(freq_offset=1.1e6, demod_mode="nfm", bandwidth=12_500.0)]
    caps = []
    for seed in (42, 43):
        carriers = [(25e3, 0.2, "nfm"), (-150e3, 0.2, "am"), (400e3, 0.2, "usb"), (-1.1e6, 0.2, "nfm")]  # last one mirrored
        caps.append(synthetic_multi_iq_s16(fs, secs, carriers, seed=seed))
    n = caps[0].shape[0]
    runner = ResidentBankRunner(targets, sample_rate=fs, n_frames=n)
    devs = [D.to_device(c.reshape(-1), "int16") for c in caps]
    torch.cuda.synchronize()
    tickets = [runner.submit(x) for x in devs]
    assert tickets[0]["launch"] == dict(lanes=1 + 2 + 5 + 1, launches=1, combines=2, pairs=0)  # (7 k steps: no lane pairs) 6401 / 8001 / 28571 / 6401 taps at D = 104
    for cap, ticket in zip(caps, tickets):
        res = runner.collect(ticket)
        for spec, r in zip(targets, res):
            want = O.run_chain(cap, sample_rate=fs, freq_offset=spec["freq_offset"], bandwidth=spec["bandwidth"],
                               demod_mode=spec["demod_mode"], agc_enabled=spec.get("agc_enabled", True), keep_decimated=False)
            assert r["sign"] == want.mix_sign == (-1 if spec["freq_offset"] == 1.1e6 else 1)
            audio = r["audio"].cpu().numpy()
            assert audio.size == want.audio.size
            assert rms(audio - want.audio) < 2e-5, (spec, rms(audio - want.audio))
            ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
            pcm = r["pcm_host"].numpy()
            assert pcm.size == ref48.size == runner.n48
            assert np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32))) <= 2
            assert abs(r["demod"].peak - want.audio_peak) < 1e-4 * max(1.0, want.audio_peak)


def test_captured_step_replays_to_the_same_audio(A):
    """ResidentCaptureRunner.submit_captured: the whole per-capture step (probes, channelizer, demodulator, resampler,
    PCM16 copy) captured into a hipGraph once per fixed buffer and replayed with one host call.  Replays must give exactly
    what the ordinary submit gives; a buffer whose capture has its carrier on the other side (probe says -1 while the
    captured step assumed +1) must come out right as well; new data in the same buffer must give new audio."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner

    fs, f_off, secs = 2.5e6, 25e3, 1.1
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    n = int(round(fs * secs))
    caps = [O.synth_capture_s16(fs, secs, f_off, seed=42), O.synth_capture_s16(fs, secs, -f_off, seed=43), O.synth_capture_s16(fs, secs, f_off, seed=44)]
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk, n_frames=n)
    bufs = [D.to_device(c.reshape(-1), "int16") for c in caps[:2]]
    torch.cuda.synchronize()
    want = []
    for x in bufs:
        r = runner.collect(runner.submit(x))
        want.append((r["sign"], r["pcm_host"].numpy().copy(), r["audio"].clone()))
    assert [w[0] for w in want] == [1, -1]
    for rep in range(3):  # first round captures (per buffer and slot), later rounds replay
        tickets = [runner.submit_captured(x) for x in bufs]
        for (sign, pcm, audio), t in zip(want, tickets):
            r = runner.collect(t)
            assert r["sign"] == sign
            assert np.array_equal(r["pcm_host"].numpy(), pcm), rep
            assert torch.equal(r["audio"], audio)
    assert 2 <= len(runner._graphs) <= 2 * runner.SLOTS  # one graph per (buffer, slot) the rounds above have touched
    # the same fixed buffer with another capture in it: the replay reads what is there now
    bufs[0].copy_(D.to_device(caps[2].reshape(-1), "int16"))
    torch.cuda.synchronize()
    r = runner.collect(runner.submit_captured(bufs[0]))
    ref = O.run_chain(caps[2], sample_rate=fs, freq_offset=f_off, keep_decimated=False)
    assert r["sign"] == 1 and rms(r["audio"].cpu().numpy() - ref.audio) < 2e-5
    ref48 = O.float_to_pcm16(O.resample_48k(ref.audio, ref.fs_channel))
    assert np.max(np.abs(r["pcm_host"].numpy().astype(np.int32) - ref48.astype(np.int32))) <= 1
    assert abs(r["demod"].peak - ref.audio_peak) < 1e-5 and len(r["demod"].chunk_rms_dbfs()) == len(ref.rms_dbfs)
    # two captures per graph (submit_captured_batch): one graph launch for both, the same audio -- including the buffer
    # whose probe says -1 (redone the ordinary way at collect)
    bufs[0].copy_(D.to_device(caps[0].reshape(-1), "int16"))
    torch.cuda.synchronize()
    for rep in range(3):
        tickets = runner.submit_captured_batch([(x, None, 0) for x in bufs])
        for (sign, pcm, audio), t in zip(want, tickets):
            r = runner.collect(t)
            assert r["sign"] == sign and np.array_equal(r["pcm_host"].numpy(), pcm) and torch.equal(r["audio"], audio), rep
    with pytest.raises(ValueError):
        runner.submit_captured_batch([(bufs[0], None, 0)] * (runner.SLOTS + 1))
    # replays on two streams (graph_streams=2: consecutive captures' chains side by side), eight slots: the same PCM16 and
    # audio, the buffer whose probe says -1 redone at collect as before
    multi = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk, n_frames=n, slots=8,
                                  graph_streams=2)
    for rep in range(3):
        tickets = [multi.submit_captured(bufs[k % 2]) for k in range(10)]
        for k, t in enumerate(tickets):
            sign, pcm, audio = want[k % 2]
            r = multi.collect(t)
            assert r["sign"] == sign and np.array_equal(r["pcm_host"].numpy(), pcm) and torch.equal(r["audio"], audio), (rep, k)


# ---- dynamic range: a full-scale interferer beside an empty channel and beside a weak one ----------


def test_stop_band_leakage_and_weak_channel_by_kernel_precision(A):
    """What the fixed-point channelizers do to a channel that holds (almost) nothing while the capture is full scale.

    Capture (10 MS/s, D = 104, 6401 taps, 0.4 s): a 0.95-of-full-scale tone at +1.3 MHz, an NFM signal at -70 dBFS
    (3.16e-4) at +1.0 MHz -- 300 kHz beside the tone -- and one LSB of noise.  Channel E ("empty") is centred at +1.6 MHz:
    300 kHz on the other side of the tone, nothing in it but the tone's leakage through the reference's own 80 dB Kaiser
    filter.  Channel W ("weak") is the -70 dBFS signal.  Every kernel form against the oracle:

      float32 VALU kernel | per-lane int8-MFMA, 16-bit taps | ring, 64-bit sums, 16-bit taps | ring, int32 sums, ~14-bit taps (default)

    The tap quantisation error is ABSOLUTE -- a fraction of the wideband level, here the 0.95 tone -- whatever the
    channel holds: the asserted bounds are per kernel form, the measured figures are printed for
    DESIGN.md section 5, and the weak channel's NFM audio shows what each form leaves of a signal 70 dB below full scale.
    """
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    fs, d, n = 10e6, 104, 4_000_000
    t = np.arange(n, dtype=np.float64) / fs
    f_tone, f_weak, f_empty = 1.3e6, 1.0e6, 1.6e6
    msg = np.sin(2 * np.pi * 1000.0 * t)
    weak = 10 ** (-70 / 20) * np.exp(1j * (2 * np.pi * f_weak * t + 3.0 * (1.0 - np.cos(2 * np.pi * 1000.0 * t))))
    x = 0.95 * np.exp(2j * np.pi * f_tone * t) + weak
    rng = np.random.default_rng(8)
    iq = np.column_stack((x.real, x.imag)) + rng.normal(scale=1.0 / 32768.0, size=(n, 2))
    raw = np.rint(np.clip(iq, -0.999, 0.999) * 32767.0).astype(np.int16).reshape(-1)
    del msg
    taps = A.design_channel_filter(fs, 12_500.0, d)
    dev = D.to_device(raw, "int16")
    forms = [("float32 VALU", dict(use_mfma=False, mfma_variant="ring", ring_acc32=True)),
             ("per-lane MFMA, 16-bit taps", dict(use_mfma=True, mfma_variant="plain", ring_acc32=True)),
             ("ring, 64-bit sums, 16-bit taps", dict(use_mfma=True, mfma_variant="ring", ring_acc32=False)),
             ("ring, int32 sums, ~14-bit taps", dict(use_mfma=True, mfma_variant="ring", ring_acc32=True))]
    keep = {k: getattr(PR._ChannelKernel, k) for k in ("use_mfma", "mfma_variant", "ring_acc32", "mfma_min_outputs")}
    rows = {}
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        for ch_name, f_c, mode in (("empty", f_empty, "am"), ("weak", f_weak, "nfm")):
            x64 = O.ingest_to_complex64(raw, "s16")
            z_ref = O.decimate(O.overlap_save(O.nco_mix(x64, O.NcoState(f_c, fs), 1), O.OverlapSaveState(taps, 65536)), O.DecimState(d))
            settled = slice(200, None)  # past the filter's start-up
            a_ref = np.clip(O.demodulate(z_ref, O.DemodState(mode, fs / d))[0], -0.99, 0.99)
            level = 20 * np.log10(rms(z_ref[settled]) + 1e-30)
            for name, flags in forms:
                for k, v in flags.items():
                    setattr(PR._ChannelKernel, k, v)
                PR._KERNEL_CACHE.clear()
                chz = A.Channelizer(taps, sample_rate=fs, freq_offset=f_c, mix_sign=1, decimation=d)
                z = chz.process(dev).cpu().numpy()
                a_gpu = np.clip(O.demodulate(z, O.DemodState(mode, fs / d))[0], -0.99, 0.99)  # the SAME (oracle) demodulator on both
                rows[(ch_name, name)] = (level, rms((z - z_ref)[settled]), rms((a_gpu - a_ref)[settled]), chz._kernel.last_kernel)
    finally:
        for k, v in keep.items():
            setattr(PR._ChannelKernel, k, v)
        PR._KERNEL_CACHE.clear()
    for (ch_name, name), (level, dz, da, kern) in rows.items():
        print(f"dynamic range, {ch_name:5s} channel ({level:7.1f} dBFS in the oracle) | {name:32s} {kern:28s}: z rms err {dz:.2e} "
              f"({20 * np.log10(dz + 1e-30):6.1f} dBFS), audio rms err {da:.2e}")
    for ch_name in ("empty", "weak"):
        assert rows[(ch_name, "float32 VALU")][1] < 3e-7
        assert rows[(ch_name, "per-lane MFMA, 16-bit taps")][1] < 5e-6
        assert rows[(ch_name, "ring, 64-bit sums, 16-bit taps")][1] < 5e-6
        assert rows[(ch_name, "ring, int32 sums, ~14-bit taps")][1] < 4e-5
        assert rows[(ch_name, "ring, int32 sums, ~14-bit taps")][3].endswith("_ring") and rows[(ch_name, "float32 VALU")][3] == "k_channelize_v1"
    assert rows[("weak", "float32 VALU")][2] < 1e-4  # the float32 kernel holds the north-star bar 70 dB below full scale


def dynamic_range_capture(n=4_000_000, fs=10e6, weak_db=-70.0):
    """A 0.95-of-full-scale tone at +1.3 MHz, an NFM signal at ``weak_db`` dBFS at +1.0 MHz, one LSB of noise: int16 frames."""
    t = np.arange(n, dtype=np.float64) / fs
    weak = 10 ** (weak_db / 20) * np.exp(1j * (2 * np.pi * 1.0e6 * t + 3.0 * (1.0 - np.cos(2 * np.pi * 1000.0 * t))))
    x = 0.95 * np.exp(2j * np.pi * 1.3e6 * t) + weak
    iq = np.column_stack((x.real, x.imag)) + np.random.default_rng(8).normal(scale=1.0 / 32768.0, size=(n, 2))
    return np.rint(np.clip(iq, -0.999, 0.999) * 32767.0).astype(np.int16)


@pytest.mark.parametrize("container", ["cs16", "cf32"])
def test_precision_guard_routes_a_very_weak_nfm_channel_to_a_finer_precision(A, tmp_path, container):
    """The pipeline's precision guard (processing.pick_precision): the dynamic-range capture of the test above
    (0.95 tone, an NFM signal at -70 dBFS 300 kHz beside it) as a file, two NFM targets in one run -- the weak signal and
    the tone itself.  The weak channel's probed level is below guard x (expected z error of the "fast" kernel at this
    wideband level), so it runs at "fine" and meets the north-star bar (1e-4) where "fast" would be 2.8e-4 off; the strong
    channel stays "fast".  ``cf32``: the same capture stored as float32 (values k / 32768): its blocks run as int16 on the
    matrix cores (``f32_integer_path``) and the guard must judge THAT kernel, not the float32 one the format maps to."""
    from iq_to_audio_amd import iqio

    fs, n, fc = 10e6, 4_000_000, 1.0e9
    raw = dynamic_range_capture(n, fs)
    path = tmp_path / f"dyn_1000000000Hz.{container}"
    path.write_bytes(raw.tobytes() if container == "cs16" else (raw.astype(np.float32) / 32768.0).astype(np.float32).tobytes())
    cfgs = [A.ProcessingConfig(in_path=path, target_freq=fc + off, demod_mode="nfm", input_sample_rate=fs,
                               output_path=tmp_path / f"g{i}.wav") for i, off in enumerate((1.0e6, 1.3e6))]
    multi = A.MultiChannelPipeline(cfgs)
    for o in multi.owners:
        o.keep_channel_audio = True
    res = multi.run()
    if container == "cf32":
        assert multi.integer_blocks >= 1  # the blocks ran as int16
    # the weak target clears the guard at "fine" (two lanes per tap-row group on the same ring kernel); the strong one stays "fast"
    assert [o.channelizer_precision for o in multi.owners] == ["fine", "fast"]
    assert [o.channelizer_kernel for o in multi.owners] == ["k_channelize_mfma_s16_ring", "k_channelize_mfma_s16_ring"]
    for off, r, o in zip((1.0e6, 1.3e6), res, multi.owners):
        want = O.run_chain(raw, sample_rate=fs, freq_offset=off, keep_decimated=False)
        got = o.audio_fs_channel.cpu().numpy()
        assert got.size == want.audio.size and r.mix_sign == want.mix_sign
        err = rms(got - want.audio)
        print(f"precision guard: target {off:+.0f} Hz through {o.channelizer_kernel}: audio rms err {err:.2e}")
        assert err < 1e-4, (off, err)


def test_channelizer_precision_ladder_against_the_oracle(A):
    """Channelizer(precision=...): "fast" (ring, one int32 sum, ~14-bit taps), "fine" (the same kernels, each tap-row
    group as two lanes: high-byte-only taps + their residue), "full" (per-lane kernel, 16-bit taps + residue as chained
    passes) and "float32" (VALU) on the dynamic-range capture's weak channel (D = 104, one tap-row group) and on a
    2.8 kHz filter at 20 MS/s (D = 208, three groups): z against the oracle, each step of the ladder at least 5x closer
    than "fast", "full" at the float32 kernel's level; the plan's own error prediction within a factor of 6 (tonal capture)."""
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import processing as PR

    keep = PR._ChannelKernel.mfma_min_outputs
    rows = {}
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        for label, fs, d, bw, f_c, n in (("weak NFM channel, D=104", 10e6, 104, 12_500.0, 1.0e6, 4_000_000),
                                         ("2.8 kHz filter, D=208, 3 groups", 20e6, 208, 2_800.0, 1.0e6, 6_000_000)):
            raw = dynamic_range_capture(n, fs).reshape(-1)
            taps = A.design_channel_filter(fs, bw, d)
            z_ref = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw, "s16"), O.NcoState(f_c, fs), 1),
                                              O.OverlapSaveState(taps, 65536)), O.DecimState(d))
            dev = D.to_device(raw, "int16")
            settled = slice(64 * 3 + 200, -64)
            wide = rms(raw.astype(np.float64) / 32768.0) * np.sqrt(2.0)
            for prec in PR._ChannelKernel.PRECISIONS:
                ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_c, mix_sign=1, decimation=d, precision=prec)
                assert ch.precision == prec
                z = ch.process(dev).cpu().numpy()
                err = rms((z - z_ref)[settled])
                pred = ch._kernel.fixed_point_error_rms(wide)
                rows[(label, prec)] = (err, pred, ch._kernel.last_kernel)
                print(f"precision ladder | {label:32s} | {prec:8s} {ch._kernel.last_kernel:28s}: z rms err {err:.2e} (predicted {pred:.2e})")
            e = {p_: rows[(label, p_)][0] for p_ in PR._ChannelKernel.PRECISIONS}
            assert rows[(label, "fast")][2].endswith("_ring") and rows[(label, "fine")][2].endswith("_ring")
            # ("full" on contiguous ring slots: lanes with 64-bit sums; the per-lane kernel only where those do not apply)
            assert rows[(label, "full")][2] == "k_channelize_mfma_s16_ring" and rows[(label, "float32")][2] == "k_channelize_v1"
            assert e["fast"] < 4e-5 and e["fine"] * 5.0 < e["fast"] and e["full"] * 5.0 < e["fine"], e
            assert e["full"] < 3e-8 and e["float32"] < 3e-7, e  # (z itself is float32: ~1e-8 of rounding on either side)
            for p_ in ("fast", "fine"):
                assert rows[(label, p_)][0] < 6.0 * rows[(label, p_)][1], (label, p_, rows[(label, p_)])
    finally:
        PR._ChannelKernel.mfma_min_outputs = keep
    # the other slot forms: row-staged ring slots (D = 26: the --benchmark rate; "fine" = residue lanes of the rows kernel,
    # "full" = chained passes of the per-lane kernel) and uint8 captures (D = 25, one data piece: "fine" products are exact)
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096
        for fmt, fs, d in (("s16", 2.5e6, 26), ("u8", 2.4e6, 25)):
            s16 = O.synth_capture_s16(fs, 1.2, 25e3, seed=5).reshape(-1)
            raw = s16 if fmt == "s16" else ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)
            taps = A.design_channel_filter(fs, 12_500.0, d)
            z_ref = O.decimate(O.overlap_save(O.nco_mix(O.ingest_to_complex64(raw, fmt), O.NcoState(25e3, fs), 1),
                                              O.OverlapSaveState(taps, 65536)), O.DecimState(d))
            dev = D.to_device(raw, "int16" if fmt == "s16" else "uint8")
            errs = {}
            for prec in ("fast", "fine", "full"):
                ch = A.Channelizer(taps, sample_rate=fs, freq_offset=25e3, mix_sign=1, decimation=d, fmt=fmt, precision=prec)
                z = ch.process(dev).cpu().numpy()
                errs[prec] = (rms((z - z_ref)[300:-64]), ch.precision, ch._kernel.last_kernel)
                print(f"precision ladder | {fmt} D={d:3d} | {prec:5s} -> {ch.precision:5s} {ch._kernel.last_kernel:28s}: z rms err {errs[prec][0]:.2e}")
            assert errs["fast"][2].endswith("_ring") and errs["fine"][2].endswith("_ring")
            assert errs["fine"][0] * 5.0 < errs["fast"][0] < 2e-5, errs
            if fmt == "s16":
                assert errs["full"][1:] == ("full", "k_channelize_mfma_s16") and errs["full"][0] < 5e-8, errs
            else:
                assert errs["full"][1] == "fine" and errs["full"][0] == errs["fine"][0], errs
    finally:
        PR._ChannelKernel.mfma_min_outputs = keep
    # uint8 captures have no per-lane kernel: "full" is "fine" there; float32 captures have the VALU kernel only
    taps = A.design_channel_filter(2.4e6, 12_500.0, 25)
    assert A.Channelizer(taps, sample_rate=2.4e6, freq_offset=1e5, mix_sign=1, decimation=25, fmt="u8", precision="full").precision == "fine"
    assert A.Channelizer(taps, sample_rate=2.4e6, freq_offset=1e5, mix_sign=1, decimation=25, fmt="f32", precision="fine").precision == "float32"
    with pytest.raises(ValueError):
        A.Channelizer(taps, sample_rate=2.4e6, freq_offset=1e5, mix_sign=1, decimation=25, precision="exactly")


@pytest.mark.parametrize("precision", ["fine", "full"])
@pytest.mark.parametrize("fs,d,bw,fmt,order,sign,n", [
    (10e6, 104, 12_500.0, "s16", "iq", 1, 5_000_000),        # loader waves, one group: two lanes of the multi kernel
    (20e6, 208, 2_800.0, "s16", "qi_inv", -1, 9_000_000),    # three groups: six lanes in pairs (int32 / 64-bit sums)
    (20e6, 208, 10_000.0, "s16", "iq_inv", 1, 9_000_000),    # two groups, conjugated sums
    (5e6, 52, 12_500.0, "s16", "qi", 1, 4_000_000),          # 4 k steps
    (50e6, 521, 12_500.0, "s16", "iq", -1, 16_000_000),      # row-staged slots, three chained k-step ranges per lane
    (1.7e6, 18, 12_500.0, "s16", "iq", 1, 3_000_000),        # D % 4 != 0: row-staged, two k steps
    (2.4e6, 25, 12_500.0, "u8", "iq", 1, 5_000_000),         # uint8: one data piece ("full" is "fine")
])
def test_finer_precisions_over_ragged_blocks_and_slot_forms(A, fs, d, bw, fmt, order, sign, n, precision):
    """Channelizer(precision="fine" / "full") on every slot form of the ring kernels, every iq_order fold and both mixer
    signs, over TWO ragged blocks (history carry, decimator phase, the float32 edges): against the float32 kernel on the
    same frames -- at least 5x closer than "fast" on the same capture and within the plan's own prediction."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd import processing as PR

    assert P.choose_decimation(fs, 96_000.0)[0] == d
    taps = A.design_channel_filter(fs, bw, d)
    s16 = O.synth_capture_s16(fs, n / fs, 0.11 * fs, seed=9).reshape(-1)
    raw = s16 if fmt == "s16" else ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)
    x = D.to_device(raw, "int16" if fmt == "s16" else "uint8")
    cut = 2 * (n // 2 + 4_321)
    keep = PR._ChannelKernel.mfma_min_outputs
    try:
        PR._ChannelKernel.mfma_min_outputs = 4096

        def run(prec):
            ch = A.Channelizer(taps, sample_rate=fs, freq_offset=0.11 * fs, mix_sign=sign, decimation=d, fmt=fmt, iq_order=order, precision=prec)
            z = torch.cat([ch.process(x[:cut]), ch.process(x[cut:])])
            return z, ch

        z_ref, _ = run("float32")
        z_fast, _ = run("fast")
        z, ch = run(precision)
    finally:
        PR._ChannelKernel.mfma_min_outputs = keep
    assert z.numel() == z_ref.numel() == -(-n // d)
    settled = slice(64 * (len(taps) // (64 * d) + 2), None)
    e = float((z - z_ref)[settled].abs().pow(2).mean().sqrt())
    e_fast = float((z_fast - z_ref)[settled].abs().pow(2).mean().sqrt())
    wide = rms(s16.astype(np.float64) / 32768.0) * np.sqrt(2.0)
    pred = ch._kernel.fixed_point_error_rms(wide)
    print(f"{precision} -> {ch.precision} at fs={fs / 1e6:g} MS/s D={d} {fmt} {order} sign {sign:+d} ({ch._kernel.last_kernel}): "
          f"z rms err {e:.2e} (fast {e_fast:.2e}, predicted {pred:.2e})")
    assert ch._kernel.last_kernel.startswith("k_channelize_mfma_")
    assert e * 5.0 < e_fast and e_fast < 3e-5, (e, e_fast)
    assert e < 6.0 * pred + 8e-8, (e, pred)  # (+ the float32 grid of z itself: ~3.4e-8 RMS at |z| ~ 0.7, on either side)


def test_precision_guard_in_the_batch_runners(A):
    """The benchmarked paths keep the 1e-4 bar on a weak channel too (round-2 finding: they had no guard).  The -70 dBFS
    NFM signal beside a full-scale tone through ResidentCaptureRunner (direct launches and the captured hipGraph step) and
    through ResidentBankRunner (weak + strong target): the first capture runs at the speculated "fast", its probe's
    level against the wideband level of the warm-up block asks for "fine" and ``collect`` re-runs it (like a
    mis-speculated sign); the second capture is speculated at "fine" and is not re-run.  Audio < 1e-4 RMS vs the oracle."""
    import torch

    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner

    fs, n = 10e6, 4_000_000
    raw = dynamic_range_capture(n, fs)
    dev = D.to_device(raw.reshape(-1), "int16")
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    want = {off: O.run_chain(raw, sample_rate=fs, freq_offset=off, keep_decimated=False) for off in (1.0e6, 1.3e6)}
    torch.cuda.synchronize()
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=1.0e6, decimation=d, fs_channel=fs_ch,
                                   chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n)
    for k in range(2):
        r = runner.collect(runner.submit(dev, resident=bool(k)))
        err = rms(r["audio"].cpu().numpy() - want[1.0e6].audio)
        print(f"batch guard, ResidentCaptureRunner capture {k}: precision {r['precision']}, audio rms err {err:.2e}, redone {runner.redone}")
        assert r["precision"] == "fine" and r["sign"] == want[1.0e6].mix_sign and err < 1e-4, (k, r["precision"], err)
    assert runner.redone == dict(sign=0, precision=1)  # the first capture only
    # the captured step: a fresh runner captures at the speculated "fast", its replay is redone at "fine"; the next
    # submit captures a NEW graph at "fine" (the speculation is part of the graph key) and its replays stand
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=1.0e6, decimation=d, fs_channel=fs_ch,
                                   chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n)
    for k in range(4):
        r = runner.collect(runner.submit_captured(dev))
        err = rms(r["audio"].cpu().numpy() - want[1.0e6].audio)
        assert r["precision"] == "fine" and err < 1e-4, (k, r["precision"], err)
    assert runner.redone["precision"] <= 2 and runner.redone["sign"] == 0, runner.redone
    # with the guard off the same capture stays at "fast" and misses the bar (what round 2 shipped)
    off_runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=1.0e6, decimation=d, fs_channel=fs_ch,
                                       chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n, precision_guard=0.0)
    r = off_runner.collect(off_runner.submit(dev))
    err_fast = rms(r["audio"].cpu().numpy() - want[1.0e6].audio)
    print(f"batch guard off: precision {r['precision']}, audio rms err {err_fast:.2e}")
    assert r["precision"] == "fast" and err_fast > 1e-4
    bank = ResidentBankRunner([dict(freq_offset=1.0e6), dict(freq_offset=1.3e6)], sample_rate=fs, n_frames=n)
    for k in range(2):
        res = bank.collect(bank.submit(dev))
        assert [r_["precision"] for r_ in res] == ["fine", "fast"]
        for off, r_ in zip((1.0e6, 1.3e6), res):
            err = rms(r_["audio"].cpu().numpy() - want[off].audio)
            print(f"batch guard, ResidentBankRunner capture {k} target {off:+.0f} Hz: precision {r_['precision']}, audio rms err {err:.2e}")
            assert r_["sign"] == want[off].mix_sign and err < 1e-4, (k, off, err)
    assert bank.redone == dict(sign=0, precision=1)


def test_demodulate_sharded_single_process_both_axes(A):
    """batch.demodulate_sharded -- the function an N-GPU job calls on every rank (dist.run_sharded underneath; its
    collectives are driven over gloo in tests/test_dist_gloo.py) -- without a process group, i.e. as rank 0 of 1: the
    channel axis (one capture, the targets as one bank) and the capture axis (two captures, every target each) must give
    each unit's 48 kHz PCM16 as the oracle's chain + resampler does."""
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd.batch import demodulate_sharded
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    fs, secs = 2.5e6, 0.8
    targets = [dict(freq_offset=25e3, demod_mode="nfm"), dict(freq_offset=-150e3, demod_mode="am", bandwidth=10_000.0)]
    caps = [synthetic_multi_iq_s16(fs, secs, [(25e3, 0.3, "nfm"), (-150e3, 0.3, "am")], seed=s) for s in (42, 43)]
    n = caps[0].shape[0]

    def ref48(cap, spec):
        w = O.run_chain(cap, sample_rate=fs, freq_offset=spec["freq_offset"], bandwidth=spec.get("bandwidth", 12_500.0),
                        demod_mode=spec["demod_mode"], keep_decimated=False)
        return O.float_to_pcm16(O.resample_48k(w.audio, w.fs_channel)), w.audio_peak

    got, peak = demodulate_sharded(targets, sample_rate=fs, n_frames=n, axis="channels", capture=D.to_device(caps[0].reshape(-1), "int16"))
    assert sorted(got) == [0, 1]
    peaks = []
    for i, spec in enumerate(targets):
        want, pk = ref48(caps[0], spec)
        peaks.append(pk)
        assert got[i].dtype == np.int16 and got[i].size == want.size
        assert np.max(np.abs(got[i].astype(np.int32) - want.astype(np.int32))) <= 1
    assert abs(peak - max(peaks)) < 1e-4
    loads = []

    def loader(k):
        def load():
            loads.append(k)  # a capture is loaded by the rank that owns it, once
            return D.to_device(caps[k].reshape(-1), "int16")
        return load

    got, _ = demodulate_sharded(targets, sample_rate=fs, n_frames=n, axis="captures", captures=[loader(0), loader(1)])
    assert sorted(got) == [0, 1] and loads == [0, 1]
    for k in (0, 1):
        assert got[k].shape[0] == len(targets)
        for i, spec in enumerate(targets):
            want, _ = ref48(caps[k], spec)
            assert np.max(np.abs(got[k][i].astype(np.int32) - want.astype(np.int32))) <= 1
    with pytest.raises(ValueError):
        demodulate_sharded(targets, sample_rate=fs, n_frames=n, axis="files")
