"""Command-line shim for the hot path.

Only the reference flags that feed ``ProcessingConfig`` and ``--benchmark*`` exist here
(reference ``cli.py:151-412, 500-578, 661-741``); the GUI, ``digital`` docker sub-command,
``--audio-post`` squelch mode and ``--plot-stages`` are out of scope.  Same exit codes
(0 ok / cancelled, 1 processing error, 2 usage error via argparse), same limits (at most five
``--ft`` targets, duplicates within 0.5 Hz rejected), same output naming
(``audio_<ft>_48k.wav``, ``_<freq>`` suffix on explicit ``--out`` with several targets,
``*_preview`` for ``--preview``).

    python -m iq_to_audio_amd.cli --in capture.wav --ft 400025000 --demod nfm
    python -m iq_to_audio_amd.cli --benchmark
"""
from __future__ import annotations

import argparse
import dataclasses
import logging
import math
import sys
from pathlib import Path

from .benchmark import run_benchmark
from .processing import MultiChannelPipeline, ProcessingCancelled, ProcessingConfig, ProcessingPipeline

LOG = logging.getLogger("iq_to_audio_amd")

_CODECS = {"u8": "pcm_u8", "cu8": "pcm_u8", "pcm_u8": "pcm_u8", "s16": "pcm_s16le", "cs16": "pcm_s16le",
           "s16le": "pcm_s16le", "pcm_s16le": "pcm_s16le", "f32": "pcm_f32le", "cf32": "pcm_f32le",
           "f32le": "pcm_f32le", "pcm_f32le": "pcm_f32le"}


def positive_float(text: str) -> float:
    value = float(text)
    if value <= 0:
        raise argparse.ArgumentTypeError("must be positive")
    return value


def parse_user_format(text: str) -> tuple[str | None, str]:
    """``[wav:|raw:]<codec>`` -> (container or None, codec)."""
    container = None
    if ":" in text:
        container, text = text.split(":", 1)
        container = container.lower()
        if container not in ("wav", "raw"):
            raise ValueError(f"unknown container '{container}'")
    codec = _CODECS.get(text.lower())
    if codec is None:
        raise ValueError(f"unknown sample format '{text}'")
    return container, codec


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="iq-to-audio-amd",
                                description="Extract and demodulate narrowband channels from SDR I/Q captures on an MI355X.")
    p.add_argument("--in", dest="input_path", type=Path, help="Input capture (WAV PCM_U8/PCM_16/FLOAT or raw .cu8/.cs16/.cf32).")
    p.add_argument("--ft", dest="target_freqs", type=positive_float, action="append", default=None,
                   help="Target RF frequency in Hz. Supply up to five times to batch additional channels.")
    p.add_argument("--bw", dest="bandwidth", type=positive_float, default=12_500.0)
    p.add_argument("--fc", dest="center_freq", type=positive_float, help="Centre frequency in Hz if filename parsing fails.")
    p.add_argument("--fs-ch", dest="fs_ch", type=positive_float, default=96_000.0)
    p.add_argument("--demod", dest="demod", choices=["nfm", "am", "usb", "lsb", "ssb", "none"], default="nfm")
    p.add_argument("--deemph", dest="deemph_us", type=positive_float, default=300.0)
    p.add_argument("--no-agc", dest="agc_enabled", action="store_false")
    p.add_argument("--out", dest="output_path", type=Path)
    p.add_argument("--dump-iq", dest="dump_iq", type=Path)
    p.add_argument("--chunk", dest="chunk_size", type=int, default=1_048_576)
    p.add_argument("--fft-workers", dest="fft_workers", type=int, help="Accepted for compatibility; unused (no FFT on this path).")
    p.add_argument("--filter-block", dest="filter_block", type=int, default=65_536)
    p.add_argument("--iq-order", dest="iq_order", choices=["iq", "qi", "iq_inv", "qi_inv"], default="iq")
    p.add_argument("--input-format", dest="input_format")
    p.add_argument("--input-sample-rate", dest="input_sample_rate", type=positive_float)
    p.add_argument("--mix-sign", dest="mix_sign", type=int, choices=[-1, 1])
    p.add_argument("--probe-only", dest="probe_only", action="store_true")
    p.add_argument("--preview", dest="preview_seconds", type=positive_float)
    p.add_argument("--benchmark", dest="benchmark", action="store_true")
    p.add_argument("--benchmark-seconds", dest="benchmark_seconds", type=positive_float, default=5.0)
    p.add_argument("--benchmark-sample-rate", dest="benchmark_sample_rate", type=positive_float, default=2_500_000.0)
    p.add_argument("--benchmark-offset", dest="benchmark_offset", type=float, default=25_000.0)
    p.add_argument("--cli", dest="cli", action="store_true", help="Accepted for compatibility (there is no GUI here).")
    p.add_argument("--verbose", dest="verbose", action="store_true")
    return p


def _preview_output_path(config: ProcessingConfig) -> Path:
    """reference preview.py:15-21"""
    base = config.output_path or config.in_path.with_name(f"audio_{int(config.target_freq)}_48k.wav")
    return base.with_name(f"{base.stem}_preview{base.suffix}")


def main(argv: list[str] | None = None) -> int:
    parser = build_parser()
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO, format="%(levelname)s %(message)s")
    frequencies = list(args.target_freqs or [])
    container = codec = None
    if args.input_format:
        try:
            container, codec = parse_user_format(args.input_format)
        except ValueError as exc:
            parser.error(f"--input-format: {exc}")
    if len(frequencies) > 5:
        parser.error("At most five target frequencies are supported per run.")
    for i, f in enumerate(frequencies):
        if any(math.isclose(f, g, rel_tol=0.0, abs_tol=0.5) for g in frequencies[:i]):
            parser.error("Duplicate target frequencies are not allowed.")

    shared = dict(bandwidth=args.bandwidth, center_freq=args.center_freq,
                  center_freq_source="cli" if args.center_freq is not None else None, demod_mode=args.demod,
                  fs_ch_target=args.fs_ch, deemph_us=args.deemph_us, agc_enabled=args.agc_enabled,
                  chunk_size=args.chunk_size, filter_block=args.filter_block, iq_order=args.iq_order,
                  probe_only=args.probe_only, mix_sign_override=args.mix_sign, fft_workers=args.fft_workers,
                  input_format=codec, input_container=container, input_format_source="cli" if codec else None,
                  input_sample_rate=args.input_sample_rate)

    if args.benchmark:
        try:
            return run_benchmark(seconds=args.benchmark_seconds, sample_rate=args.benchmark_sample_rate,
                                 freq_offset=args.benchmark_offset, center_freq=args.center_freq,
                                 target_freq=frequencies[0] if frequencies else None, base_kwargs=shared)
        except Exception as exc:  # noqa: BLE001 - user-facing exit code, as the reference does
            LOG.error("Benchmark failed: %s", exc)
            return 1

    if args.input_path is None:
        parser.error("--in is required (or use --benchmark).")
    if not frequencies and not args.probe_only:
        parser.error("Provide at least one --ft target frequency.")

    def annotate(base: Path | None, freq: float) -> Path | None:
        if base is None or len(frequencies) <= 1:
            return base
        return base.with_name(f"{base.stem}_{int(round(freq))}{base.suffix}")

    configs = []
    for freq in frequencies or [0.0]:
        config = ProcessingConfig(in_path=args.input_path, target_freq=freq, output_path=annotate(args.output_path, freq),
                                  dump_iq_path=annotate(args.dump_iq, freq), **shared)
        if args.preview_seconds:
            config = dataclasses.replace(config, max_input_seconds=args.preview_seconds,
                                         output_path=_preview_output_path(config))
        configs.append(config)
    LOG.info("=== Processing %d target(s) in one pass over %s ===", len(configs), args.input_path)
    try:
        # the reference loops whole pipelines over the targets (cli.py:683-710); here the capture is read once
        results = MultiChannelPipeline(configs).run(progress_sink=None) if len(configs) > 1 else [
            ProcessingPipeline(configs[0]).run(progress_sink=None)]
    except ProcessingCancelled:
        LOG.info("Processing cancelled by user.")
        return 0
    except Exception as exc:  # noqa: BLE001
        LOG.error("Processing failed: %s", exc)
        if args.verbose:
            LOG.exception("Debug traceback")
        return 1
    for config, result in zip(configs, results):
        LOG.info("%.0f Hz: decimation %d -> %.2f Hz, mixer sign %+d, audio peak %.4f", config.target_freq,
                 result.decimation, result.fs_channel, result.mix_sign, result.audio_peak)
    return 0


if __name__ == "__main__":
    sys.exit(main())
