#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path.

Metric (BASELINE.json): complex IQ MS/s end-to-end, device-resident int16 capture ->
48 kHz PCM16 audio on the host, for BASELINE config 2: synthetic 60 s @ 10 MS/s, one NFM
channel (+25 kHz offset, bw 12.5 kHz, de-emphasis 300 us, requested chunk 1 048 576 ->
effective 4 194 304, D = 104, 6401 taps).

One *step* = one pass of the whole hot path over the capture:
  mixer-sign probe -> fused ingest+mix+FIR+decimate -> discriminator -> de-emphasis scan ->
  peak/clip -> 48 kHz polyphase resample -> PCM16 -> D2H of the audio [-> RCCL gather at N>1].
Steps are queued through batch.ResidentCaptureRunner (a batch of captures with one set of settings):
no host<->device synchronisation inside a step, the channelizer runs speculatively for mixer sign +1
behind the probes (checked when the capture is collected), and the D2H of one capture overlaps the
kernels of the next.
The capture is resident in HBM before the timed region (the first 5 s are the reference's
seed-42 generator, tiled to 60 s in HBM -- SURVEY.md section 8(d)).

The one JSON line also carries (rank 0, N = 1; all bounded so the default run stays within minutes):
  value_host_resident -- the same step with the 2.4 GB capture starting in PINNED HOST memory every time (H2D inside
                         the timed region, double-buffered): the PCIe-inclusive rate.  Never `value`.
  configs             -- BASELINE config 1 (the reference's own --benchmark capture, 5 s @ 2.5 MS/s), config 3
                         (60 s @ 20 MS/s, five simultaneous targets nfm/am/usb/lsb/nfm, AGC on; ONE pass of the
                         channelizer for the "fast" targets, the SSB+AGC targets at "full" precision behind it), config 4's
                         per-GPU unit (60 s @ 20 MS/s, one channel) and config 5's (1.2 G frames @ 50 MS/s, five NFM
                         channels, D = 521), each with ms_per_step, dominant kernel, roofline fraction and parity.
  file_to_wav         -- the LITERAL metric: a PCM16 WAV on disk (tmpfs) -> iq_to_audio_amd ProcessingPipeline.run -> 48 kHz
                         PCM16 WAV on disk, wall clock around run() as the reference's --benchmark times it
                         (benchmark.py:104-120), "x realtime" as it prints it, the oracle's DSP time beside it.
  roofline.traffic    -- HBM bytes per launch from separate rocprofv3 --pmc passes of this command (profiles/), tagged
                         with its source; null when no matching profile is committed.  It is NOT measured in this run.

N > 1 (launched by torch.distributed.run): every rank processes its own independent capture
(BASELINE config 4 pattern, seeds 42+rank), no data-path collective, only the finished 48 kHz
PCM16 audio is gathered to rank 0 over RCCL (iq_to_audio_amd.dist.AudioGather: asynchronously, overlapping
the next step).  scaling = "weak".

--axis channels (any N): BASELINE config 5 -- ONE 120 s @ 50 MS/s capture (24 GB) on rank 0, replicated with one RCCL
broadcast (timed apart: config.broadcast_s), its 40 NFM channels in contiguous shares of ceil(40/N) per rank, every
rank's share extracted as one bank per step (iq_to_audio_amd.dist.ShardedJob + batch.ResidentBankRunner), every
channel's 48 kHz PCM16 gathered on rank 0 per step.  Total work is fixed: scaling = "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

C3_TARGETS = [  # offsets, generator / demodulator modes, bandwidths of BASELINE config 3
    dict(freq_offset=25e3, demod_mode="nfm", bandwidth=12_500.0),
    dict(freq_offset=-150e3, demod_mode="am", bandwidth=10_000.0),
    dict(freq_offset=400e3, demod_mode="usb", bandwidth=2_800.0),
    dict(freq_offset=-1.1e6, demod_mode="lsb", bandwidth=2_800.0),
    dict(freq_offset=2.3e6, demod_mode="nfm", bandwidth=12_500.0),
]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults long enough to time the SUSTAINED rate: after an idle->load edge the part runs ~8 captures at boost
    # clocks, overshoots its 1400 W package limit, is clamped for ~25 captures (channelizer 0.84 ms instead of 0.57)
    # and settles by capture ~40 (profiles/power_trace.sh); 100 + 1000 captures take 0.8 s.
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle", type=int, default=60,
                    help="the untimed phase is at least this many captures long: max(0, settle - warmup) extra untimed "
                         "captures run in front of the warm-up steps (the power-management transient above)")
    ap.add_argument("--seconds", type=float, default=60.0, help="capture length (config 2: 60 s)")
    ap.add_argument("--sample-rate", type=float, default=10e6)
    ap.add_argument("--unique-seconds", type=float, default=5.0, help="seed-42 prefix generated on the host, then tiled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip value_host_resident, the configs array and file_to_wav")
    ap.add_argument("--axis", choices=("captures", "channels"), default="captures",
                    help="captures (default): one independent capture per GPU (BASELINE configs 2 / 4); channels: BASELINE "
                         "config 5 -- one 50 MS/s capture broadcast to every GPU, its 40 channels sharded")
    ap.add_argument("--channels", type=int, default=40, help="--axis channels: NFM channels of the capture (config 5: 40)")
    return ap.parse_args()


class quiet_gc:
    """Around a timed region: collect now, then no garbage collection until the region ends (as timeit does).  A generation-2
    collection of this process takes 35-55 ms (profiles/r03_c1_gc_stall.txt: ONE of them inside 400 replays of 53 us each made
    config 1's step read 0.19 ms instead of 0.053)."""

    def __enter__(self):
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            gc.enable()

    def __call__(self, fn):  # as a decorator: the whole sub-bench (reference counting still frees its arrays)
        import functools

        @functools.wraps(fn)
        def inner(*a, **k):
            with quiet_gc():
                return fn(*a, **k)
        return inner


def rms_err(a, b) -> float:
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))


def padded_resident(host_i16: np.ndarray, n_total: int, slack: int):
    """The capture tiled to ``n_total`` frames inside a slightly larger device buffer (readable slack behind it)."""
    import torch

    from iq_to_audio_amd import _dev as D

    tile = torch.from_numpy(host_i16).to(D.device())
    buf = torch.zeros(2 * (n_total + slack), dtype=torch.int16, device=D.device())
    buf[: 2 * n_total] = tile.repeat(-(-2 * n_total // tile.numel()))[: 2 * n_total]
    del tile
    return buf


@quiet_gc()
def sub_bench_c1(steps: int = 400, warm: int = 100) -> dict:
    """BASELINE config 1: the reference's --benchmark capture (5 s @ 2.5 MS/s, NFM, +25 kHz) through the same runner."""
    import torch

    import iq_to_audio_amd as A
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_iq_s16
    from oracle import cpu_ref as O

    fs, secs, f_off = 2.5e6, 5.0, 25e3
    n = int(round(fs * secs))
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    host = synthetic_iq_s16(fs, secs, f_off).reshape(-1)  # (the product's generator = the reference's, benchmark.py:19-38)
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = padded_resident(host, n, slack)
    raw = buf[: 2 * n]
    torch.cuda.synchronize()
    # eight captures in flight: with two, every 85 us step is a host round trip (event wait + wake-up + graph launch),
    # which the round-2 driver box took 0.236 ms for where the builder's boxes took 0.085
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch,
                                   chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n, slots=8)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    ts = [runner.submit(raw, enclosing=buf, lead_frames=0, resident=True) for _ in range(warm)]
    for t in ts:
        runner.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [runner.submit(raw, events=ev[i], enclosing=buf, lead_frames=0, resident=True) for i in range(steps)]
    res = [runner.collect(t) for t in ts][-1]
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / steps
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the same step captured into a hipGraph per (buffer, slot) and replayed: one host call per capture
    for t in [runner.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(warm)]:
        runner.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts, host_us = [], []
    for _ in range(steps):
        h0 = time.perf_counter()
        ts.append(runner.submit_captured(raw, enclosing=buf, lead_frames=0))  # (waits for the capture `slots` back first)
        host_us.append((time.perf_counter() - h0) * 1e6)
    res = [runner.collect(t) for t in ts][-1]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # ... the same replays on FOUR streams, round-robin by slot: a 50 MB capture's step is a chain of ten dependent kernels of a
    # few microseconds each (80 us of GPU time per capture: latency, not work), so the chains of several captures side by side
    # fill each other's gaps -- independent captures, the same audio (two streams: 54 us per capture, three 47, four 39)
    runner2 = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch,
                                    chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n, slots=8, graph_streams=4)
    for t in [runner2.submit_captured(raw, enclosing=buf, lead_frames=0) for _ in range(warm)]:
        runner2.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts2, host_us2 = [], []
    for _ in range(steps):
        h0 = time.perf_counter()
        ts2.append(runner2.submit_captured(raw, enclosing=buf, lead_frames=0))
        host_us2.append((time.perf_counter() - h0) * 1e6)
    res2 = [runner2.collect(t) for t in ts2][-1]
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / steps
    audio2 = res2["audio"].cpu().numpy()
    # ... and two captures per graph (a batch of captures in fixed buffers: one graph launch for both)
    pair = [(raw, buf, 0)] * 2
    for _ in range(warm // 2):
        for t in runner.submit_captured_batch(pair):
            runner.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batches = -(-steps // 2)
    for _ in range(batches):
        for t in runner.submit_captured_batch(pair):
            res_b = runner.collect(t)
    torch.cuda.synchronize()
    dt_batch = (time.perf_counter() - t0) / (batches * 2)
    want = O.run_chain(host, sample_rate=fs, freq_offset=f_off, keep_decimated=False)
    audio = res["audio"].cpu().numpy()
    audio_b = res_b["audio"].cpu().numpy()
    algo = (4.0 + 4.0 * 48_000.0 / fs) * n
    return {
        "workload": "BASELINE config 1 (the reference's --benchmark capture): 5 s @ 2.5 MS/s int16 I/Q, 1 NFM channel, +25 kHz, "
                    f"bw 12.5 kHz, D={d}, {len(taps)} taps",
        "value": round(n / dt2 / 1e6, 1), "unit": "MS/s", "ms_per_step": round(dt2 * 1e3, 4), "steps": steps,
        "step": f"captured into a hipGraph per (buffer, slot) and replayed (ResidentCaptureRunner.submit_captured), {runner2.SLOTS} captures in "
                "flight, replays on four streams (independent captures side by side)",
        "ms_per_step_one_stream": round(dt * 1e3, 4), "four_streams_audio_identical": bool(np.array_equal(audio, audio2)),
        "host_us_per_replay": {"mean": round(float(np.mean(host_us)), 1), "median": round(float(np.median(host_us)), 1),
                               "max": round(float(np.max(host_us)), 1),
                               "note": "host time inside submit_captured: wait for the capture `slots` back + hipGraphLaunch + event record"},
        "host_us_per_replay_four_streams": {"mean": round(float(np.mean(host_us2)), 1), "median": round(float(np.median(host_us2)), 1),
                                           "max": round(float(np.max(host_us2)), 1),
                                           "by_quarter": [round(float(np.mean(q)), 1) for q in np.array_split(np.asarray(host_us2), 4)]},
        "ms_per_step_direct_launches": round(dt_eager * 1e3, 4), "replays_redone": dict(runner.redone),
        "ms_per_step_two_captures_per_graph": round(dt_batch * 1e3, 4), "two_per_graph_audio_identical": bool(np.array_equal(audio, audio_b)),
        "roofline": {"kernel": res["kernel"], "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": algo,
                     "achieved": round(algo / (kern_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
        "parity": {"rms_err_vs_oracle_fs_channel": rms_err(audio, want.audio), "samples_compared": int(want.audio.size),
                   "sample_count_exact": bool(audio.size == want.audio.size), "bar": 1e-4},
    }


def _bank_traffic_ratio(tag_key: str):
    """(ratio, source) of measured HBM traffic over algorithmic bytes for a bank workload, from the committed rocprofv3 PMC
    summary (profiles/pmc_bank.sh); (None, reason) when no matching profile is committed."""
    for name in ("r03b_bank_pmc_summary.json", "r03bf_bank_pmc_summary.json", "r03_bank_pmc_summary.json", "r02c_bank_pmc_summary.json"):
        path = ROOT / "profiles" / name
        if not path.exists():
            continue
        try:
            rec = json.loads(path.read_text())
        except ValueError:
            continue
        ratio = rec.get("channelizer_stage_traffic_over_algorithmic")
        if ratio is not None and rec.get("workload_key", "c3_all_fast") == tag_key:
            return float(ratio), f"profiles/{name} (separate rocprofv3 --pmc passes of profiles/bench_bank.py; NOT measured in this run)"
    return None, f"no committed PMC profile for workload {tag_key!r}"


@quiet_gc()
def sub_bench_c3(steps: int = 20, warm: int = 6, cpu_seconds_of_signal: float = 0.55) -> dict:
    """BASELINE config 3: 60 s @ 20 MS/s, five simultaneous targets (nfm/am/usb/lsb/nfm, bw 12.5k/10k/2.8k/2.8k/12.5k),
    AGC on (ResidentBankRunner).  Timed twice: at the precisions the product chooses (USB / LSB with the AGC on at "full":
    chained passes of the per-lane kernel behind the shared pass of the three "fast" targets -- what keeps their audio at
    the 1e-4 bar) and with every target forced to "fast" (round 2's launch: ONE pass for all)."""
    import torch

    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16
    from oracle import cpu_ref as O

    fs, secs, uniq = 20e6, 60.0, 2.0
    n = int(round(fs * secs))
    d, _ = P.choose_decimation(fs, 96_000.0)
    carriers = [(t["freq_offset"], 0.14, t["demod_mode"]) for t in C3_TARGETS]
    host = synthetic_multi_iq_s16(fs, uniq, carriers).reshape(-1)
    slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32_769)[1], 8192)
    buf = padded_resident(host, n, slack)
    raw = buf[: 2 * n]
    torch.cuda.synchronize()

    def timed(targets):
        runner = ResidentBankRunner(targets, sample_rate=fs, n_frames=n)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for t in [runner.submit(raw, enclosing=buf, lead_frames=0) for _ in range(warm)]:
            runner.collect(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [runner.submit(raw, events=ev[i], enclosing=buf, lead_frames=0) for i in range(steps)]
        res = [runner.collect(t) for t in ts][-1]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return dt, float(np.mean([a.elapsed_time(b) for a, b in ev])), ts[-1], res, runner

    dt_fast, chan_fast, tk_fast, _, r_fast = timed([dict(t, precision="fast") for t in C3_TARGETS])
    del r_fast
    torch.cuda.empty_cache()
    dt, chan_ms, tk, res, runner = timed(C3_TARGETS)
    launch = getattr(runner, "last_bank_launches", None) or [tk["launch"]]  # (one entry per shared-ingest launch of a capture)
    # parity per target on the un-tiled prefix (whole reference chunks are not needed: the oracle sees the same frames)
    n_cpu = int(round(cpu_seconds_of_signal * fs))
    parity = []
    for spec, r in zip(C3_TARGETS, res):
        want = O.run_chain(host[: 2 * n_cpu], sample_rate=fs, freq_offset=spec["freq_offset"], bandwidth=spec["bandwidth"],
                           demod_mode=spec["demod_mode"], agc_enabled=True)
        k = want.audio.size - 64  # (the capture continues behind the oracle's sample: its last outputs see other frames)
        z_err = rms_err(np.abs(r["z"][:k].cpu().numpy() - want.decimated[:k]), 0.0)
        entry = {"target": f'{spec["demod_mode"]} {spec["freq_offset"]:+.0f} Hz bw {spec["bandwidth"]:.0f}', "sign": r["sign"],
                 "precision": r["precision"], "z_rms_err": z_err, "samples_compared": int(k),
                 "audio_rms_err": rms_err(r["audio"][:k].cpu().numpy(), want.audio[:k])}
        if spec["demod_mode"] in ("usb", "lsb"):
            entry["note"] = ("SSB + AGC is ill-conditioned in the reference (DESIGN.md section 5: its own output moves by ~2e-3 RMS for a "
                             "3e-7 change of z); held link by link in tests/test_gpu_configs.py")
        parity.append(entry)
    algo = 4.0 * n + len(C3_TARGETS) * 4.0 * 48_000.0 / fs * n
    # Algorithmic matrix work: the decimating FIR needs 4 L / D real 16 x 16-bit MACs per input frame and channel; on int8
    # pieces one such MAC is three int8 MACs (q1 hi, q1 lo, q2 hi) = 6 int8 ops.  (The launches execute more: tap rows are
    # allocated in groups of 64 per component, and a "full"-precision target runs every group twice: taps + their residue.)
    taps = [len(P.design_channel_filter(fs, t["bandwidth"], d)) for t in C3_TARGETS]
    int8_ops = 6.0 * 4.0 * sum(taps) / d * n
    ratio, ratio_src = _bank_traffic_ratio("c3_product_precisions")
    ratio_fast, ratio_fast_src = _bank_traffic_ratio("c3_all_fast")
    return {
        "workload": "BASELINE config 3: 60 s @ 20 MS/s int16 I/Q, 5 simultaneous targets nfm/am/usb/lsb/nfm "
                    f"({'/'.join(str(t) for t in taps)} taps), AGC on, D={d}; precisions {'/'.join(r['precision'] for r in res)}",
        "value": round(n / dt / 1e6, 1), "unit": "MS/s of capture", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
        "channel_samples_per_s": round(len(C3_TARGETS) * n / dt / 1e9, 2),
        "replays_redone": dict(runner.redone),
        "roofline": {"bound": "mfma", "kernel": "k_channelize_mfma_s16_ring" + ("_pairs" if any(l_ and l_.get("pairs") for l_ in launch) else "_multi")
                                               + " (int32 sums: the 'fast' targets) + the same with 64-bit sums (the 'full' targets: taps + residue lanes)",
                     "launch": launch, "kernel_ms": round(chan_ms, 4),
                     "note": "kernel_ms = every channelizer launch of a capture on the caller's stream, by events: the shared multi-lane "
                             "pass of the 'fast' targets and the one of the 'full' targets (the combine launches run on "
                             "the side stream and are not in it); achieved = algorithmic int8 ops (3 int8 MACs per 16x16-bit tap x sample "
                             "MAC) / kernel time; peak = dense int8 MFMA (2 x the 2.5 PFLOP/s bf16 figure of MI355X_MICROARCH.md)",
                     "achieved": round(int8_ops / (chan_ms * 1e-3) / 1e12, 1), "peak": 5000.0, "unit": "TOP/s",
                     "frac": round(int8_ops / (chan_ms * 1e-3) / 1e12 / 5000.0, 5), "algorithmic_int8_ops_per_launch": int8_ops,
                     "hbm": {"algorithmic_bytes_per_launch": algo, "achieved_gb_per_s": round(algo / (chan_ms * 1e-3) / 1e9, 2),
                             "frac_of_8_tb_per_s": round(algo / (chan_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                             "traffic_over_algorithmic": ratio, "traffic_source": ratio_src}},
        "all_targets_fast": {"note": "the same capture with every target forced to precision 'fast' (round 2's launch: ONE multi-lane pass for "
                                     "all five targets; SSB+AGC audio then 2e-2 .. 8e-2 RMS off the reference)",
                             "ms_per_step": round(dt_fast * 1e3, 3), "kernel_ms": round(chan_fast, 4), "launch": tk_fast["launch"],
                             "mfma_frac": round(int8_ops / (chan_fast * 1e-3) / 1e12 / 5000.0, 5),
                             "traffic_over_algorithmic": ratio_fast, "traffic_source": ratio_fast_src},
        "parity": {"bar": 1e-4, "per_target": parity},
    }


@quiet_gc()
def sub_bench_c4_unit(steps: int = 60, warm: int = 60, cpu_seconds_of_signal: float = 0.5) -> dict:
    """BASELINE config 4's per-GPU unit: 60 s @ 20 MS/s int16, one NFM channel (D = 208, 12 801 taps, 13 k steps: the
    ring kernel without loader waves), through the same runner as the headline."""
    import torch

    import iq_to_audio_amd as A
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_iq_s16
    from oracle import cpu_ref as O

    fs, secs, f_off, uniq = 20e6, 60.0, 25e3, 2.0
    n = int(round(fs * secs))
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    taps = A.design_channel_filter(fs, 12_500.0, d)
    host = synthetic_iq_s16(fs, uniq, f_off).reshape(-1)
    _, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = padded_resident(host, n, slack)
    raw = buf[: 2 * n]
    torch.cuda.synchronize()
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch,
                                   chunk=P.tune_chunk_size(fs, 1_048_576), n_frames=n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for t in [runner.submit(raw, enclosing=buf, lead_frames=0, resident=True) for _ in range(warm)]:
        runner.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [runner.submit(raw, events=ev[i], enclosing=buf, lead_frames=0, resident=True) for i in range(steps)]
    res = [runner.collect(t) for t in ts][-1]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    n_cpu = int(round(cpu_seconds_of_signal * fs))
    want = O.run_chain(host[: 2 * n_cpu], sample_rate=fs, freq_offset=f_off, keep_decimated=False)
    k = want.audio.size - 64
    algo = (4.0 + 4.0 * 48_000.0 / fs) * n
    return {
        "workload": f"BASELINE config 4, one GPU's unit: 60 s @ 20 MS/s int16 I/Q, 1 NFM channel, +25 kHz, bw 12.5 kHz, D={d}, {len(taps)} taps",
        "value": round(n / dt / 1e6, 1), "unit": "MS/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "warmup": warm,
        "roofline": {"bound": "hbm", "kernel": res["kernel"], "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": algo,
                     "achieved": round(algo / (kern_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
        "parity": {"rms_err_vs_oracle_fs_channel": rms_err(res["audio"][:k].cpu().numpy(), want.audio[:k]), "samples_compared": int(k),
                   "precision": res["precision"], "sign": res["sign"], "bar": 1e-4},
    }


def c5_targets(n_channels: int = 40) -> list:
    """BASELINE config 5: NFM carriers on a 100 kHz raster from -1.95 MHz, amplitude 0.02 each."""
    return [dict(freq_offset=-1.95e6 + 100e3 * k, demod_mode="nfm", bandwidth=12_500.0) for k in range(n_channels)]


@quiet_gc()
def sub_bench_c5_unit(steps: int = 20, warm: int = 6, uniq: float = 0.2) -> dict:
    """BASELINE config 5's per-GPU unit: five of the 40 NFM channels (first, second, the two around DC, last) of a
    50 MS/s capture, D = 521 (row-staged ring slots, 33 k steps in three chained launches of five lanes), on 1.2 G frames
    (24 s of the 120 s capture: the same per-frame work, a fifth of its length)."""
    import torch

    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16
    from oracle import cpu_ref as O

    fs, n = 50e6, 1_200_000_000
    d, _ = P.choose_decimation(fs, 96_000.0)
    every = c5_targets(40)
    picks = [0, 1, 19, 20, 39]
    targets = [every[k] for k in picks]
    host = synthetic_multi_iq_s16(fs, uniq, [(t["freq_offset"], 0.02, "nfm") for t in every]).reshape(-1)
    slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32_001)[1], 8192)
    buf = padded_resident(host, n, slack)
    raw = buf[: 2 * n]
    torch.cuda.synchronize()
    runner = ResidentBankRunner(targets, sample_rate=fs, n_frames=n)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for t in [runner.submit(raw, enclosing=buf, lead_frames=0) for _ in range(warm)]:
        runner.collect(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [runner.submit(raw, events=ev[i], enclosing=buf, lead_frames=0) for i in range(steps)]
    res = [runner.collect(t) for t in ts][-1]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    chan_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    launch = ts[-1]["launch"]
    parity = []
    for spec, r in list(zip(targets, res))[:: len(targets) - 1]:  # first and last channel (the oracle takes ~3 s per channel)
        want = O.run_chain(host, sample_rate=fs, freq_offset=spec["freq_offset"], bandwidth=spec["bandwidth"], demod_mode="nfm")
        k = want.audio.size - 64
        parity.append({"target": f'nfm {spec["freq_offset"]:+.0f} Hz', "sign": r["sign"], "precision": r["precision"], "samples_compared": int(k),
                       "z_rms_err": rms_err(np.abs(r["z"][:k].cpu().numpy() - want.decimated[:k]), 0.0),
                       "audio_rms_err": rms_err(r["audio"][:k].cpu().numpy(), want.audio[:k])})
    algo = 4.0 * n + len(targets) * 4.0 * 48_000.0 / fs * n
    ntaps = len(P.design_channel_filter(fs, 12_500.0, d))
    int8_ops = 6.0 * 4.0 * len(targets) * ntaps / d * n
    return {
        "workload": f"BASELINE config 5, one GPU's unit: 1.2 G frames (24 s) @ 50 MS/s int16 I/Q, {len(targets)} of the 40 NFM channels, "
                    f"D={d}, {ntaps} taps, 33 k steps in three chained launches",
        "value": round(n / dt / 1e6, 1), "unit": "MS/s of capture", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
        "channel_samples_per_s": round(len(targets) * n / dt / 1e9, 2), "replays_redone": dict(runner.redone),
        "roofline": {"bound": "mfma", "kernel": "k_channelize_mfma_s16_ring_rows_multi", "launch": launch, "kernel_ms": round(chan_ms, 4),
                     "achieved": round(int8_ops / (chan_ms * 1e-3) / 1e12, 1), "peak": 5000.0, "unit": "TOP/s",
                     "frac": round(int8_ops / (chan_ms * 1e-3) / 1e12 / 5000.0, 5), "algorithmic_int8_ops_per_launch": int8_ops,
                     "hbm": {"algorithmic_bytes_per_launch": algo, "achieved_gb_per_s": round(algo / (chan_ms * 1e-3) / 1e9, 2),
                             "frac_of_8_tb_per_s": round(algo / (chan_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}},
        "parity": {"bar": 1e-4, "per_target": parity},
    }


@quiet_gc()
def file_to_wav_legs() -> list:
    """The literal metric -- file in, 48 kHz WAV out -- as the reference's --benchmark times it (benchmark.py:104-120): a
    PCM16 stereo WAV on disk -> ProcessingPipeline.run -> 48 kHz PCM16 WAV on disk, wall clock around run(), generation of
    the capture outside.  In a tmpfs (/dev/shm) so that the figure is the pipeline's, not a disk's.  Beside it the oracle's
    DSP time on the same frames (in memory: the reference's own run() needs ffmpeg, which no box here has)."""
    import shutil

    from iq_to_audio_amd.benchmark import synthetic_iq_s16, timed_file_run
    from oracle import cpu_ref as O

    tmp_root = "/dev/shm" if (os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)) else None
    legs = []
    for label, fs, secs, repeats in (("BASELINE config 1 (the reference's --benchmark): 5 s @ 2.5 MS/s", 2.5e6, 5.0, 3),
                                     ("config 2's rate: 10 s @ 10 MS/s", 10e6, 10.0, 2)):
        f_off, fc = 25e3, 400e6
        try:
            free = shutil.disk_usage(tmp_root or "/tmp").free
            if free < 3 * 4 * fs * secs:
                raise RuntimeError(f"not enough room in {tmp_root or '/tmp'} ({free} bytes free)")
            run = timed_file_run(seconds=secs, sample_rate=fs, tone_offset=f_off, center_freq=fc, target_freq=fc + f_off,
                                 settings=dict(bandwidth=12_500.0, chunk_size=1_048_576), mode="nfm", tmp_root=tmp_root, repeats=repeats,
                                 keep_audio=True)
            raw = synthetic_iq_s16(fs, secs, f_off)
            t1 = time.perf_counter()
            want = O.run_chain(raw, sample_rate=fs, freq_offset=f_off, keep_decimated=False)
            ref48 = O.float_to_pcm16(O.resample_48k(want.audio, want.fs_channel))
            cpu_s = time.perf_counter() - t1
            pcm = run["audio_48k"]
            same = pcm.size == ref48.size
            legs.append({
                "workload": f"{label}, int16 stereo WAV in {'tmpfs' if tmp_root else '/tmp'} -> ProcessingPipeline.run -> 48 kHz PCM16 WAV",
                "frames": run["frames"], "wall_s_first_run": round(run["runs_s"][0], 4), "wall_s_later_runs": [round(t, 4) for t in run["runs_s"][1:]],
                "value": round(run["frames"] / min(run["runs_s"]) / 1e6, 1), "unit": "MS/s (best run)",
                "value_first_run": round(run["frames"] / run["runs_s"][0] / 1e6, 1),
                "x_realtime_first_run": round(secs / run["runs_s"][0], 1), "x_realtime_best": round(secs / min(run["runs_s"]), 1),
                "note": "wall clock around run(): header parse, memory-mapped read, pinned staging, H2D, probes, channelizer, demodulator, "
                        "48 kHz resampler, PCM16, D2H, WAV write; the first run also pays plan creation, tap uploads and pinning",
                "oracle": {"dsp_wall_s": round(cpu_s, 3), "value": round(run["frames"] / cpu_s / 1e6, 2), "unit": "MS/s", "cores": 1,
                           "x_realtime": round(secs / cpu_s, 2),
                           "note": "oracle/cpu_ref.run_chain + resample_48k on the same frames in memory (no file I/O, no ffmpeg processes)"},
                "parity": {"sample_count_exact": bool(same), "mix_sign": int(run["result"].mix_sign),
                           "pcm16_max_abs_diff_lsb": int(np.max(np.abs(pcm.astype(np.int32) - ref48.astype(np.int32)))) if same else None},
            })
        except Exception as exc:  # noqa: BLE001 - an extra must never take the headline down
            legs.append({"workload": label, "error": repr(exc)})
    return legs


@quiet_gc()
def host_resident_leg(runner, host_pinned, bufs, n_total: int, captures: int = 8) -> dict:
    """The step with the capture starting in pinned host memory: its upload (2.4 GB over PCIe) is inside the timed
    region, double-buffered against the previous capture's kernels."""
    import torch

    from iq_to_audio_amd import _dev as D

    up = D.side_stream("upload")

    def step(i):
        buf = bufs[i % 2]
        with torch.cuda.stream(up):  # upload of capture i beside the kernels of capture i-1
            buf[: 2 * n_total].copy_(host_pinned, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        torch.cuda.current_stream().wait_event(ev)
        return runner.submit(buf[: 2 * n_total], enclosing=buf, lead_frames=0)

    def run(count):
        # capture i reuses the buffer of capture i - 2: that one is collected (its kernels have read the buffer) before the
        # upload of capture i is queued
        tickets = []
        for i in range(count):
            if i >= 2:
                runner.collect(tickets[i - 2])
            tickets.append(step(i))
        for t in tickets[-2:]:
            runner.collect(t)

    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(captures)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / captures
    return {"value": round(n_total / dt / 1e6, 1), "unit": "MS/s", "ms_per_step": round(dt * 1e3, 2), "steps": captures,
            "pcie_gb_per_s": round(4.0 * n_total / dt / 1e9, 1),
            "note": "capture in pinned host memory at the start of every step (H2D inside the timed region, double-buffered); "
                    "bounded by the PCIe link, not the GPU"}


def main_channel_axis(args, json_fd: int) -> None:
    """``--axis channels``: BASELINE config 5 -- one 120 s @ 50 MS/s capture, its 40 NFM channels sharded over the ranks
    (5 per GPU at N = 8).  What a rank executes is ``dist.ShardedJob`` (one broadcast, then steps of stage + gather: the
    functions tests/test_dist_gloo.py drives at world 2 and 3) around ``batch.ResidentBankRunner`` (one shared-ingest pass
    per step for the rank's channels).  ref: the reference's sequential loop over --ft targets, each re-reading the file
    (cli.py:683-710)."""
    import torch

    import iq_to_audio_amd as A
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dist as DS
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentBankRunner, ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

    rank, world, local_rank = DS.check_launch_env(args.gpus, torch.cuda.device_count())
    A.native.lib()
    distributed = world > 1 or bool(os.environ.get("IQA_FORCE_DIST") or os.environ.get("IQA_BENCH_FORCE_DIST"))
    if distributed:
        os.environ.setdefault("IQA_FORCE_DIST", "1")
    torch.cuda.set_device(local_rank)
    A.native.require_gpu()
    if distributed:
        DS.init_from_env("nccl", high_priority=True)
    fs = 50e6 if args.sample_rate == 10e6 else float(args.sample_rate)  # (the default --sample-rate is config 2's)
    secs = 120.0 if args.seconds == 60.0 else float(args.seconds)
    n_total = int(round(fs * secs))
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    targets = c5_targets(args.channels)
    slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32_001)[1], 8192)
    numel = 2 * (n_total + slack)
    buf0 = None
    if rank == 0:
        uniq = min(secs, 0.2)
        host = synthetic_multi_iq_s16(fs, uniq, [(t["freq_offset"], 0.02, "nfm") for t in targets]).reshape(-1)
        buf0 = padded_resident(host, n_total, slack)
        torch.cuda.synchronize()
    job = DS.ShardedJob(targets, shared=dict(tensor=buf0, numel=numel, dtype=torch.int16, device=D.device()), sync=torch.cuda.synchronize)
    buf = job.common
    raw = buf[: 2 * n_total]
    mine = job.my_units()
    runner = ResidentBankRunner(mine, sample_rate=fs, n_frames=n_total) if mine else None

    def stage(units, capture):
        res = runner.collect(runner.submit(raw, enclosing=buf, lead_frames=0))
        return [(r["pcm"], r["demod"].peak) for r in res]  # (device PCM16, valid until this slot's next capture: gathered right away)

    steps = args.steps if args.steps != 1000 else 5  # (the defaults of the capture axis are sized for 0.7 ms steps)
    warm = args.warmup if args.warmup != 100 else 2
    for _ in range(warm):
        job.step(stage)
    with quiet_gc():
        DS.fence(None, sync=torch.cuda.synchronize)
        t0 = time.perf_counter()
        got = peak = None
        for _ in range(steps):
            got, peak = job.step(stage)
        DS.fence(None, sync=torch.cuda.synchronize)
        elapsed = DS.max_over_ranks(time.perf_counter() - t0)
    ms_per_step = elapsed / steps * 1e3
    algo = 4.0 * n_total + len(targets) * 4.0 * 48_000.0 / fs * n_total
    out = {
        "metric": "complex IQ MS/s end-to-end (ingest->48 kHz audio)", "value": round(n_total / (elapsed / steps) / 1e6, 1), "unit": "MS/s",
        "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
        "config": {
            "workload": f"BASELINE config 5: ONE synthetic {secs:g} s @ {fs/1e6:g} MS/s int16 I/Q capture ({numel * 2 / 1e9:.1f} GB), "
                        f"{len(targets)} NFM channels (100 kHz raster, bw 12.5 kHz, de-emphasis), D={d}",
            "axis": "channels", "frames": n_total, "channels": len(targets), "channels_per_gpu": -(-len(targets) // world),
            "parallelism": f"the capture on every GPU (one RCCL broadcast from rank 0, {job.broadcast_s:.3f} s, outside the timed steps); "
                           f"{-(-len(targets) // world)} channels per GPU extracted in one pass per step; RCCL gather of 48 kHz PCM16 only",
            "broadcast_s": round(job.broadcast_s, 4),
            "broadcast_gb_per_s": round(numel * 2 / job.broadcast_s / 1e9, 1) if (world > 1 and job.broadcast_s > 0) else None,
            "channel_samples_per_s_G": round(len(targets) * n_total / (elapsed / steps) / 1e9, 2),
            "audio_units_gathered": len(got) if got is not None else None, "audio_peak": peak,
            "value_is": "frames of the capture per second with ALL channels extracted (total work fixed as N grows)",
        },
        "roofline": {"bound": "hbm", "kernel": "k_channelize_mfma_s16_ring_rows_multi", "achieved": round(algo / (elapsed / steps) / 1e9, 2),
                     "peak": HBM_PEAK_GBPS * world, "unit": "GB/s", "frac": round(algo / (elapsed / steps) / 1e9 / (HBM_PEAK_GBPS * world), 5),
                     "traffic": None, "note": "whole step (all launches, demodulators, resamplers, gather) against N x 8 TB/s on the algorithmic "
                                              "bytes 4 B/frame + 4 B x 48 kHz per channel; the per-kernel figure of this shape is configs[3] of the default run"},
    }
    if distributed:
        import torch.distributed as dist

        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


def main() -> None:
    args = parse_args()
    # stdout carries exactly one line, the JSON: whatever native libraries print while the job runs (RCCL's version
    # banner at communicator creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # (see iq_to_audio_amd/__init__.py; before the runtime initialises)
    if args.axis == "channels":
        return main_channel_axis(args, json_fd)
    import torch

    import iq_to_audio_amd as A
    from iq_to_audio_amd import _dev as D
    from iq_to_audio_amd import dist as DS
    from iq_to_audio_amd import dsp_plan as P
    from iq_to_audio_amd.batch import ResidentCaptureRunner
    from iq_to_audio_amd.benchmark import synthetic_iq_s16

    # the launch is validated before anything touches a GPU (device_count() does not initialise one on this image)
    rank, world, local_rank = DS.check_launch_env(args.gpus, torch.cuda.device_count())
    A.native.lib()
    # IQA_FORCE_DIST=1: rehearse the N > 1 code path (process group, per-capture gather, barrier, max over ranks)
    # with a single rank on a one-GPU box
    distributed = world > 1 or bool(os.environ.get("IQA_FORCE_DIST") or os.environ.get("IQA_BENCH_FORCE_DIST"))
    if distributed:
        os.environ.setdefault("IQA_FORCE_DIST", "1")
        # RCCL's send/recv kernel needs 19.7 KB of LDS and ~280 VGPRs per workgroup (librccl's gfx950 code object): it
        # cannot share a CU with a channelizer workgroup (149 KB of LDS at D = 104), and a channelizer launch that wants
        # all 256 CUs waits for every CU a gather still sits on.  So with a gather in the job the capture-long launches
        # use 240 workgroups (measured cost at N = 1: 1.5 %, the clock gives most of it back) and the gather -- 5.76 MB
        # per peer and capture, nowhere near needing more -- is kept to 8 channels = 8 workgroups.
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "8")
        from iq_to_audio_amd import processing as _PR

        _PR._ChannelKernel.launch_blocks = int(os.environ.get("IQA_BENCH_RING_BLOCKS", "240"))
    torch.cuda.set_device(local_rank)
    A.native.require_gpu()
    if distributed:
        DS.init_from_env("nccl", high_priority=True)

    fs, f_off, bw = float(args.sample_rate), 25e3, 12_500.0
    n_total = int(round(fs * args.seconds))
    n_unique = min(n_total, int(round(fs * args.unique_seconds)))
    d, fs_ch = P.choose_decimation(fs, 96_000.0)
    chunk = P.tune_chunk_size(fs, 1_048_576)
    taps = A.design_channel_filter(fs, bw, d)

    # ---- synthetic capture, resident in HBM ---------------------------------------------------
    host = synthetic_iq_s16(fs, n_unique / fs, f_off, seed=42 + rank).reshape(-1)
    # the capture sits inside a slightly larger buffer: a few KB of readable slack behind it make the last outputs
    # interior outputs of the matrix-core kernel too (no VALU tail launch)
    lead, slack = ResidentCaptureRunner.padded_capture_frames(d, len(taps))
    buf = padded_resident(host, n_total, slack)
    raw = buf[: 2 * n_total]
    torch.cuda.synchronize()

    # one runner per configuration: plans are made once, every step queues probes + channelizer + demod/resample/PCM16
    # (compute stream) and the D2H (egress stream) without a host<->device synchronisation
    runner = ResidentCaptureRunner(taps, sample_rate=fs, freq_offset=f_off, decimation=d, fs_channel=fs_ch, chunk=chunk,
                                   n_frames=n_total, demod_mode="nfm", deemph_us=300.0, agc_enabled=True, fmt="s16")
    n48 = runner.n48
    # finished 48 kHz PCM16 audio of every capture is gathered on rank 0 (as bytes: RCCL has no int16 type), on the
    # runner's egress stream, at most one gather in flight
    gather = DS.AudioGather(2 * n48, dst=0, stream=runner.egress, device=D.device()) if distributed else None
    settle = max(0, args.settle - args.warmup)
    n_untimed = settle + args.warmup
    ev_k0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + n_untimed)]
    ev_k1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + n_untimed)]
    tickets = []
    lag = []  # the capture whose audio has not been handed to the gather yet

    def step(i: int):
        # resident: the capture was complete in HBM before the timed region (the metric's premise)
        t = runner.submit(raw, events=(ev_k0[i], ev_k1[i]), enclosing=buf, lead_frames=lead, resident=True)
        tickets.append(t)
        if gather is not None:
            # the PREVIOUS capture's audio goes to the gather now: this capture's first timing event lies behind its
            # last kernel, so no event of its own is needed (an event record costs the compute stream ~7 us)
            if lag:
                gather.queue(lag.pop()["pcm"], after=ev_k0[i])
            lag.append(t)
        return t

    def fence():
        if lag:  # the last capture's audio
            t = lag.pop()
            gather.queue(t["pcm"], after=runner.tail_event(t))
        for t in tickets:
            runner.collect(t)
        del tickets[:]
        DS.fence(gather, sync=torch.cuda.synchronize)

    # Everything slow on the host happens BEFORE the first capture: a generation-2 collection in the middle of a 0.7 ms
    # step is a 10-70 ms stall, and that much idle GPU starts the power-management transient all over again.
    gc.collect()
    gc.disable()
    for i in range(n_untimed):
        if i == n_untimed - 2:
            fence()  # one untimed step runs right after a fence, exactly like the first timed step will
        step(i)
    stats0 = torch.cuda.memory_stats()
    fence()
    t0 = time.perf_counter()
    marks = []
    last = None
    for i in range(args.steps):
        last = step(n_untimed + i)
        marks.append(time.perf_counter() - t0)
    fence()  # (queues the last capture's gather before its buffers go back to the runner)
    res = runner.collect(last)
    sign, audio, kernel_name = res["sign"], res["audio"], [res["kernel"]]
    if os.environ.get("IQA_BENCH_DEBUG"):
        stats1 = torch.cuda.memory_stats()
        keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
        print("host-side step completion times (ms):", [round(m * 1e3, 2) for m in marks[:80]],
              "after fence:", round((time.perf_counter() - t0) * 1e3, 2),
              "allocator deltas:", {k: stats1.get(k, 0) - stats0.get(k, 0) for k in keys}, file=sys.stderr)
    elapsed = DS.max_over_ranks(time.perf_counter() - t0)
    gc.enable()

    kern_ms = [ev_k0[n_untimed + i].elapsed_time(ev_k1[n_untimed + i]) for i in range(args.steps)]
    if os.environ.get("IQA_BENCH_DEBUG"):
        # GPU-side series over warm-up + timed steps: period between consecutive channelizer starts, channelizer time
        n_all = n_untimed + args.steps
        period = [ev_k0[i].elapsed_time(ev_k0[i + 1]) for i in range(n_all - 1)]
        kern = [ev_k0[i].elapsed_time(ev_k1[i]) for i in range(n_all)]
        pick = (list(range(n_all - 1)) if n_all <= 200 else
                sorted(set(list(range(0, min(n_all - 1, 40))) + list(range(40, n_all - 1, max(1, n_all // 60))))))
        print("step: period_ms kernel_ms", " | ".join(f"{i}: {period[i]:.3f} {kern[i]:.3f}" for i in pick), file=sys.stderr)
    kern_avg_ms = float(np.mean(kern_ms))
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_total / (elapsed / args.steps) / 1e6  # MS/s, whole job

    # algorithmic bytes per complex input sample: int16 I+Q read once + 48 kHz f32 out
    bytes_per_sample = 4.0 + 1 * 4.0 * 48_000.0 / fs
    algo_bytes = bytes_per_sample * n_total
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic = traffic_source = None
    pmc = ROOT / "profiles" / "pmc_summary.json"
    if pmc.exists():
        with pmc.open() as fh:
            rec = json.load(fh)
        if rec.get("workload_frames") == n_total and kernel_name[0] in (rec.get("kernel") or ""):
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_source = ("profiles/pmc_summary.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              f"command ({rec.get('tag', 'untagged')}), (2 x FETCH_SIZE + WRITE_SIZE) x 1024; NOT measured in this run")

    label = ("BASELINE config 1 (the reference's --benchmark capture)" if (fs, args.seconds) == (2.5e6, 5.0) else
             "BASELINE config 2" if (fs, args.seconds) == (10e6, 60.0) else
             "BASELINE config 4, one GPU's capture" if (fs, args.seconds) == (20e6, 60.0) else "non-BASELINE shape")
    out = {
        "metric": "complex IQ MS/s end-to-end (ingest->48 kHz audio)",
        "value": round(value, 1),
        "unit": "MS/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i8",
        "data": "synthetic",
        "config": {
            "workload": f"{label}: synthetic {args.seconds:g} s @ {fs/1e6:g} MS/s int16 I/Q, 1 NFM channel, "
                        f"+25 kHz, bw 12.5 kHz, D={d}, {len(taps)} taps, chunk {chunk}",
            "frames_per_gpu": n_total,
            "parallelism": f"{world} independent capture(s), one per GPU; RCCL gather of 48 kHz audio only",
            "audio_samples_48k": int(n48),
            "mix_sign": int(sign),
            "settle_steps": settle,
            "residency": "capture resident in HBM before the timed region (value); see value_host_resident for the PCIe-inclusive rate",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name[0],
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel_ms": round(kern_avg_ms, 4),
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_gsps": round(n_total / (kern_avg_ms * 1e-3) / 1e9, 2),
        },
    }
    if gather is not None:
        out["config"]["gathers"] = gather.count

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_ref as O

        # bounded sample: ~cpu_seconds of CPU work at the ~19 MS/s this port runs at, in whole reference chunks,
        # of the SAME tiled capture the GPU processed (so the parity check below covers the whole sample)
        n_cpu = min(n_total, max(1, int(round(args.cpu_seconds * 19.0e6 / chunk))) * chunk)
        host_cpu = np.tile(host, -(-2 * n_cpu // host.size))[: 2 * n_cpu]
        t1 = time.perf_counter()
        ref = O.run_chain(host_cpu, sample_rate=fs, freq_offset=f_off, keep_decimated=False)
        cpu_s = time.perf_counter() - t1
        # parity of the benchmarked GPU output against the oracle on the same sample
        got = audio[: ref.audio.size].cpu().numpy()
        out["cpu_baseline"] = {
            "value": round(n_cpu / cpu_s / 1e6, 2),
            "unit": "MS/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {n_cpu} frames ({n_cpu / fs:.2f} s of signal) of the same capture, oracle/cpu_ref.run_chain "
                      f"(fp64 NCO + complex128 131072-pt scipy.fft overlap-save + slice decimate + NFM), "
                      f"{cpu_s:.1f} s on 1 of {os.cpu_count()} host cpus (1-D FFTs are single-threaded)",
            "note": "the REFERENCE's own stage classes, timed in the survey container (8 vCPU Xeon 2.1 GHz, DSP stages only, no ffmpeg; "
                    "SURVEY.md section 6): 9.4 MS/s at config 1's parameters, 4.2 MS/s at 10 MS/s parameters (this config), 3.4 MS/s at "
                    "50 MS/s parameters -- the reference cannot run on the GPU box (it does not travel)",
        }
        out["parity"] = {"rms_err_vs_oracle_fs_channel": rms_err(got, ref.audio), "samples_compared": int(ref.audio.size), "bar": 1e-4}

    if rank == 0 and world == 1 and not args.no_extras and not distributed:
        # -- the same step from pinned host memory (PCIe-inclusive; never `value`) --------------------------------
        try:
            host_pinned = torch.from_numpy(np.tile(host, -(-2 * n_total // host.size))[: 2 * n_total]).pin_memory()
            second = torch.zeros_like(buf)
            out["value_host_resident"] = host_resident_leg(runner, host_pinned, [buf, second], n_total)
            del host_pinned, second
        except Exception as exc:  # noqa: BLE001 - an extra must never take the headline down
            out["value_host_resident"] = {"error": repr(exc)}
        del runner, raw, buf
        torch.cuda.empty_cache()
        # -- the other single-GPU BASELINE configurations ------------------------------------------------------
        out["configs"] = []
        for fn in (sub_bench_c1, sub_bench_c3, sub_bench_c4_unit, sub_bench_c5_unit):
            try:
                out["configs"].append(fn())
            except Exception as exc:  # noqa: BLE001
                out["configs"].append({"workload": fn.__name__, "error": repr(exc)})
            gc.collect()
            torch.cuda.empty_cache()
        # -- the literal metric: file in, 48 kHz WAV out, as the reference's --benchmark times it ----------------------
        out["file_to_wav"] = file_to_wav_legs()
        gc.collect()
        torch.cuda.empty_cache()

    if distributed:
        import torch.distributed as dist

        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
