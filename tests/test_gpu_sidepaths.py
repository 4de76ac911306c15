"""GPU tests of the pipeline's side paths: the pass-through slice writer
(``--demod none``), ``--dump-iq``, preview truncation (``max_input_seconds``) and ``probe_only``.

Reference behaviour restated here: pass-through writes the DECIMATED channel in the input's own container and codec
(processing.py:693-695, 1014-1015, 1114-1121; encodings :527-539 for headerless files, libsndfile for WAV --
third-party, parity unpinned); ``--dump-iq`` is the decimated stream as interleaved float32 (:363-378); a preview keeps
``max(1, floor(seconds * fs))`` input frames (:839-844, 1072-1081); ``probe_only`` stops after the mixer-sign probe
with ``audio_peak = 0`` and writes nothing (:1044-1065).
"""
from __future__ import annotations

import numpy as np
import pytest

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import iq_to_audio_amd as pkg

    pkg.native.lib()
    pkg.native.require_gpu()
    return pkg


FS, F_OFF, FC = 2.5e6, 25e3, 400e6


def _capture(fmt: str, secs: float = 0.9):
    s16 = O.synth_capture_s16(FS, secs, F_OFF).reshape(-1)
    if fmt == "s16":
        return s16
    if fmt == "u8":
        return ((s16.astype(np.int32) >> 8) + 128).astype(np.uint8)
    return s16.astype(np.float32) / np.float32(32768.0)


def _oracle_z(raw, fmt):
    want = O.run_chain(raw, sample_rate=FS, freq_offset=F_OFF, fmt=fmt, keep_decimated=True)
    return want


def _oracle_slice(z, codec, container):
    """The reference's encodings, written independently of the product's encoder (explicit loops over the rule)."""
    pairs = np.empty(2 * z.size, dtype=np.float32)
    pairs[0::2], pairs[1::2] = z.real, z.imag
    if codec == "pcm_f32le":
        return pairs
    if container == "wav":  # libsndfile float -> PCM with normalisation (unpinned, see module docstring)
        if codec == "pcm_s16le":
            return np.clip(np.rint(pairs.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
        return np.clip(np.rint(pairs.astype(np.float64) * 127.0) + 128, 0, 255).astype(np.uint8)
    if codec == "pcm_s16le":
        return (np.clip(pairs, -1.0, 0.999969) * 32767.0).astype(np.int16)  # truncation toward zero
    return np.round((np.clip(pairs, -1.0, 1.0) + 1.0) * 127.5).astype(np.uint8)


@pytest.mark.parametrize("container,fmt,suffix", [("wav", "s16", ".wav"), ("raw", "s16", ".cs16"), ("raw", "u8", ".cu8"),
                                                   ("raw", "f32", ".cf32"), ("wav", "f32", ".wav"), ("wav", "u8", ".wav")])
def test_pass_through_slice_keeps_container_and_codec(A, tmp_path, container, fmt, suffix):
    from iq_to_audio_amd import iqio

    raw = _capture(fmt)
    src = tmp_path / f"cap_{int(FC)}Hz{suffix}"
    if container == "wav":
        iqio.write_wav_iq(src, raw, int(FS), fmt)
    else:
        src.write_bytes(raw.tobytes())
    cfg = A.ProcessingConfig(in_path=src, target_freq=FC + F_OFF, demod_mode="none",
                             input_sample_rate=FS if container == "raw" else None)
    pipe = A.ProcessingPipeline(cfg)
    pipe.block_frames_target = 1_048_576  # several device blocks: the slice is assembled across them
    res = pipe.run()
    out = src.with_name(f"slice_{int(FC + F_OFF)}{suffix}")  # reference processing.py:1215-1233 default naming
    assert out.exists()
    want = _oracle_z(raw, fmt)
    z = want.decimated
    assert (res.decimation, res.mix_sign) == (26, want.mix_sign)
    assert abs(res.audio_peak - float(np.max(np.abs(z)))) < 1e-5  # the slice writer's peak is max |z| (:568-571)
    codec = iqio.FMT_TO_CODEC[fmt]
    if container == "wav":
        info = iqio.probe_capture(out)
        assert (info.container, info.codec, info.n_frames) == ("wav", codec, z.size)
        assert info.sample_rate == round(FS / 26)  # the slice declares the channel rate (:551)
        got = np.array(iqio.map_frames(info))
    else:
        got = np.fromfile(out, dtype=iqio.NP_DTYPE[fmt])
    ref = _oracle_slice(z, codec, container)
    assert got.size == ref.size == 2 * z.size  # sample count: exact
    if fmt == "f32":
        # (this cf32 capture is an int16 capture in disguise -> matrix-core channelizer, ~2e-6 RMS of full scale)
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5)
        assert float(np.sqrt(np.mean((got - ref) ** 2))) < 5e-6
    else:
        # z differs from the oracle's by ~2e-6 RMS (matrix-core channelizer): a value next to a quantisation step may
        # land on the other side -- 3.5 % of the int16 values (step 3.05e-5), 0.03 % of the uint8 ones
        diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff != 0) < (0.08 if fmt == "s16" else 2e-3)


def test_dump_iq_is_the_decimated_stream(A, tmp_path):
    """--dump-iq: interleaved float32 of the decimated channel, beside the normal audio output."""
    from iq_to_audio_amd import iqio

    raw = _capture("s16")
    src = tmp_path / f"cap_{int(FC)}Hz.wav"
    iqio.write_wav_iq(src, raw, int(FS), "s16")
    dump = tmp_path / "z.cf32"
    pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=src, target_freq=FC + F_OFF, dump_iq_path=dump,
                                                   output_path=tmp_path / "a.wav"))
    pipe.block_frames_target = 1_048_576
    pipe.keep_channel_audio = True
    pipe.run()
    want = _oracle_z(raw, "s16")
    z = np.fromfile(dump, dtype=np.complex64)
    assert z.size == want.decimated.size
    np.testing.assert_allclose(z, want.decimated, rtol=0, atol=3e-5)
    assert float(np.sqrt(np.mean(np.abs(z - want.decimated) ** 2))) < 1e-5
    # and the audio next to it is the audio of exactly that stream
    audio = pipe.audio_fs_channel.cpu().numpy()
    st = O.DemodState("nfm", want.fs_channel)
    replay = np.clip(O.demodulate(z, st)[0], -0.99, 0.99)
    assert np.abs(audio - replay).max() < 2e-6
    assert (tmp_path / "a.wav").exists()


@pytest.mark.parametrize("seconds", [0.25, 1e-9, 5.0])
def test_preview_truncation_counts(A, tmp_path, seconds):
    """max_input_seconds keeps max(1, floor(s * fs)) frames (a tiny value keeps one frame; a long one the whole file),
    and what is kept is processed exactly like a capture of that length."""
    from iq_to_audio_amd import iqio

    raw = _capture("s16", 0.6)
    n_all = raw.size // 2
    src = tmp_path / f"cap_{int(FC)}Hz.wav"
    iqio.write_wav_iq(src, raw, int(FS), "s16")
    cfg = A.ProcessingConfig(in_path=src, target_freq=FC + F_OFF, max_input_seconds=seconds, output_path=tmp_path / "p.wav")
    pipe = A.ProcessingPipeline(cfg)
    pipe.keep_channel_audio = True
    res = pipe.run()
    keep = min(n_all, max(1, int(np.floor(seconds * FS))))
    audio = pipe.audio_fs_channel.cpu().numpy()
    assert audio.size == -(-keep // 26)
    want = O.run_chain(raw[: 2 * keep], sample_rate=FS, freq_offset=F_OFF, keep_decimated=False)
    assert audio.size == want.audio.size and res.mix_sign == want.mix_sign
    assert float(np.sqrt(np.mean((audio - want.audio) ** 2))) < 2e-5
    pcm, rate = iqio.read_wav_pcm16_mono(tmp_path / "p.wav")
    assert rate == 48000 and pcm.size == -(-audio.size * 24000 // 48077)


def test_probe_only_reports_and_writes_nothing(A, tmp_path):
    from iq_to_audio_amd import iqio

    for off in (F_OFF, -F_OFF):  # the second capture has its carrier on the other side: the probe must say -1
        raw = O.synth_capture_s16(FS, 0.5, off)
        src = tmp_path / f"cap{'pos' if off > 0 else 'neg'}_{int(FC)}Hz.wav"
        iqio.write_wav_iq(src, raw, int(FS), "s16")
        out = tmp_path / f"never_{int(off)}.wav"
        res = A.ProcessingPipeline(A.ProcessingConfig(in_path=src, target_freq=FC + F_OFF, probe_only=True, output_path=out)).run()
        want = O.run_chain(raw, sample_rate=FS, freq_offset=F_OFF, keep_decimated=False)
        assert res.mix_sign == want.mix_sign == (1 if off > 0 else -1)
        assert (res.decimation, res.audio_peak) == (26, 0.0)
        assert abs(res.fs_channel - FS / 26) < 1e-9 and abs(res.freq_offset - F_OFF) < 1e-6
        assert not out.exists()
    # probe_only needs no target (reference processing.py:846-849: the check is skipped)
    res = A.ProcessingPipeline(A.ProcessingConfig(in_path=src, target_freq=0.0, probe_only=True)).run()
    assert res.audio_peak == 0.0 and res.target_freq == FC


def test_cli_preview_and_pass_through_naming(A, tmp_path):
    """--preview writes ``<stem>_preview<suffix>`` next to the would-be output (reference preview.py:15-21 builds the
    audio-style name whatever the demodulator; cli.py:645-658) from the first ``seconds`` of input."""
    from iq_to_audio_amd import cli, iqio

    raw = _capture("s16", 0.6)
    src = tmp_path / f"cap_{int(FC)}Hz.wav"
    iqio.write_wav_iq(src, raw, int(FS), "s16")
    assert cli.main(["--in", str(src), "--ft", str(FC + F_OFF), "--preview", "0.2"]) == 0
    prev = src.with_name(f"audio_{int(FC + F_OFF)}_48k_preview.wav")
    pcm, rate = iqio.read_wav_pcm16_mono(prev)
    assert rate == 48000 and pcm.size == -(-(-(-int(0.2 * FS) // 26)) * 24000 // 48077)
    assert not src.with_name(f"audio_{int(FC + F_OFF)}_48k.wav").exists()
    assert cli.main(["--in", str(src), "--ft", str(FC + F_OFF), "--demod", "none", "--out", str(tmp_path / "s.wav")]) == 0
    info = iqio.probe_capture(tmp_path / "s.wav")
    assert info.n_frames == -(-raw.size // 2 // 26) and info.codec == "pcm_s16le"


def test_float32_capture_that_is_an_integer_capture_takes_the_matrix_cores(A, tmp_path):
    """A cf32 capture whose values are all k / 32768 (what SDR software writes for int16 ADC samples) is re-packed to
    int16 block by block on the device (iqa_f32_to_s16_exact checks every value) and runs on the matrix-core
    channelizers: its audio is EXACTLY the audio of the same capture stored as cs16.  A capture with a single value that
    is not of that form in its last block switches to the float32 kernel there -- state carried along -- and still meets
    the oracle."""
    import torch

    fs, secs = 2.5e6, 4 * 1_048_576 / 2.5e6  # four device blocks of one reference chunk each
    s16 = O.synth_capture_s16(FS, secs, F_OFF).reshape(-1)
    f32 = s16.astype(np.float32) / np.float32(32768.0)
    outs = {}
    for name, data, suffix in (("s16", s16, ".cs16"), ("f32", f32, ".cf32")):
        path = tmp_path / f"cap_{int(FC)}Hz{suffix}"
        path.write_bytes(data.tobytes())
        pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=path, target_freq=FC + F_OFF, input_sample_rate=fs,
                                                       output_path=tmp_path / f"{name}.wav"))
        pipe.block_frames_target = 1_048_576  # four device blocks
        pipe.keep_channel_audio = True
        pipe.run()
        outs[name] = (pipe.audio_fs_channel.clone(), pipe.channelizer_kernel, pipe._multi.integer_blocks)
    assert outs["f32"][1] == outs["s16"][1] == "k_channelize_mfma_s16_ring"
    assert (outs["s16"][2], outs["f32"][2]) == (0, 4)
    assert torch.equal(outs["f32"][0], outs["s16"][0])
    # one value off the 2^-15 grid in the last block: that block runs as TWO int16 planes (hi + lo / 32768), still on the
    # matrix cores; one value outside +-1 in the last block: that one the planes cannot hold -> float32 kernel, state carried
    for label, delta, blocks, kernel in (("bent", np.float32(1e-6), (3, 1), "k_channelize_mfma_s16_ring"),
                                         ("over", np.float32(1.5), (3, 0), "k_channelize_v1")):
        bent = f32.copy()
        bent[2 * 3_500_000 + 1] += delta
        path = tmp_path / f"{label}_{int(FC)}Hz.cf32"
        path.write_bytes(bent.tobytes())
        pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=path, target_freq=FC + F_OFF, input_sample_rate=fs, output_path=tmp_path / "b.wav"))
        pipe.block_frames_target = 1_048_576
        pipe.keep_channel_audio = True
        pipe.run()
        assert (pipe._multi.integer_blocks, pipe._multi.split_blocks) == blocks and pipe.channelizer_kernel == kernel, label
        want = O.run_chain(bent, sample_rate=fs, freq_offset=F_OFF, fmt="f32", keep_decimated=False)
        got = pipe.audio_fs_channel.cpu().numpy()
        assert got.size == want.audio.size
        assert float(np.sqrt(np.mean((got - want.audio) ** 2))) < 2e-5, label
    # and the switch itself
    pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=tmp_path / f"cap_{int(FC)}Hz.cf32", target_freq=FC + F_OFF,
                                                   input_sample_rate=fs, output_path=tmp_path / "off.wav"))
    pipe.f32_integer_path = False
    pipe.run()
    assert pipe._multi.integer_blocks == 0 and pipe.channelizer_kernel == "k_channelize_v1"


@pytest.mark.parametrize("grid", ["rtlsdr", "k32767", "float"])
def test_fractional_float32_captures_run_as_two_int16_planes(A, tmp_path, grid):
    """cf32 captures that are NOT on the 2^-15 grid -- RTL-SDR's (u - 127.5) / 127.5 (reaches +-1: one bit of headroom),
    int16 samples scaled by 1 / 32767, and genuinely continuous floats -- run as two int16 planes on the matrix-core
    channelizers (iqa_f32_split_s16: x = 2^shift (hi + lo / 32768) / 32768, z = 2^shift (z(hi) + 2^-15 z(lo))) instead
    of the float32 VALU kernel: NFM audio within 2e-5 RMS of the oracle's float32 ingest of the very same values, two
    device blocks (the low plane's channelizers carry their own history)."""
    fs, secs = 2.5e6, 2.9 * 1_048_576 / 2.5e6  # (the last block long enough for the matrix-core kernels: 36 k outputs)
    s16 = O.synth_capture_s16(FS, secs, F_OFF).reshape(-1)
    if grid == "rtlsdr":
        u8 = np.clip((s16.astype(np.int32) >> 8) + 128, 0, 255).astype(np.uint8)
        u8[:2] = (255, 0)  # the grid's end points: +1.0 and -1.0
        f32 = ((u8.astype(np.float32) - np.float32(127.5)) / np.float32(127.5)).astype(np.float32)
    elif grid == "k32767":
        f32 = (s16.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    else:
        rng = np.random.default_rng(17)
        f32 = (s16.astype(np.float64) / 32768.0 * 0.83 + rng.normal(scale=1e-5, size=s16.size)).astype(np.float32)
    path = tmp_path / f"frac_{int(FC)}Hz.cf32"
    path.write_bytes(f32.tobytes())
    pipe = A.ProcessingPipeline(A.ProcessingConfig(in_path=path, target_freq=FC + F_OFF, input_sample_rate=fs, output_path=tmp_path / "f.wav"))
    pipe.block_frames_target = 1_048_576
    pipe.keep_channel_audio = True
    res = pipe.run()
    n_blocks = -(-(f32.size // 2) // 1_048_576)
    assert (pipe._multi.integer_blocks, pipe._multi.split_blocks) == (0, n_blocks) and pipe.channelizer_kernel == "k_channelize_mfma_s16_ring"
    want = O.run_chain(f32, sample_rate=fs, freq_offset=F_OFF, fmt="f32", keep_decimated=False)
    got = pipe.audio_fs_channel.cpu().numpy()
    assert got.size == want.audio.size and res.mix_sign == want.mix_sign
    err = float(np.sqrt(np.mean((got - want.audio) ** 2)))
    print(f"fractional float32 capture ({grid}): {n_blocks} blocks as two int16 planes, audio rms err {err:.2e}")
    assert err < 2e-5, (grid, err)
