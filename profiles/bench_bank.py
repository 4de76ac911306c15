"""Channelizer stage of BASELINE config 3 (60 s @ 20 MS/s, five targets: 10 lanes) and of config 5's per-GPU unit
(50 MS/s, D = 521, five NFM channels x three k-step passes) on a resident capture: ChannelBank (one launch for all
channels, shared ingest) against the same channelizers one at a time.  Times by CUDA events around the channelizer work
only (no demodulator): the figures for DESIGN.md section 6.   K=20 WARM=5 python profiles/bench_bank.py [c3|c5] [bank|single]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _native as NATIVE
if os.environ.get("IQA_LIB"):  # an experiment build of the library
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
from iq_to_audio_amd import _dev as D, dsp_plan as P
from iq_to_audio_amd.batch import ResidentCaptureRunner
from iq_to_audio_amd.benchmark import synthetic_multi_iq_s16

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
modes = sys.argv[2:] or ["bank", "single"]
if which == "c3":
    fs, secs, uniq = 20e6, 60.0, 2.0
    targets = [(25e3, "nfm", 12500.0), (-150e3, "am", 10000.0), (400e3, "usb", 2800.0), (-1.1e6, "lsb", 2800.0), (2.3e6, "nfm", 12500.0)]
elif which == "nfm4":  # four single-group NFM channels at config 3's rate: two lane pairs of equal tap-row group
    fs, secs, uniq = 20e6, 60.0, 2.0
    targets = [(25e3, "nfm", 12500.0), (-1.3e6, "nfm", 12500.0), (2.3e6, "nfm", 12500.0), (-3.3e6, "nfm", 12500.0)]
else:
    fs, secs, uniq = 50e6, float(os.environ.get("SECS", "24")), 1.0  # 24 s = 4.8 GB of the 120 s capture
    targets = [(-1.95e6 + 100e3 * k, "nfm", 12500.0) for k in (0, 1, 19, 20, 39)]
n_total = int(fs * secs)
host = synthetic_multi_iq_s16(fs, uniq, [(o, 0.14 if which in ("c3", "nfm4") else 0.02, m) for o, m, _ in targets]).reshape(-1)
d, fs_ch = P.choose_decimation(fs, 96000.0)
slack = max(ResidentCaptureRunner.padded_capture_frames(d, 32769)[1], 8192)
buf = torch.zeros(2 * (n_total + slack), dtype=torch.int16, device="cuda")
buf[: 2 * n_total] = torch.from_numpy(host).cuda().repeat(-(-n_total // (host.size // 2)))[: 2 * n_total]
raw = buf[: 2 * n_total]
n_dec = -(-n_total // d)
outs = [D.empty(n_dec, "complex64") for _ in targets]

PREC = os.environ.get("PREC", "fast")  # "fast": every target at the default precision (round 2's workload); "product": what the
                                       # pipeline picks per demodulator (USB / LSB with the AGC on at "full": processing.base_precision)


def make():
    from iq_to_audio_amd.processing import base_precision
    return [A.Channelizer(A.design_channel_filter(fs, bw, d), sample_rate=fs, freq_offset=off, mix_sign=1, decimation=d,
                          precision=base_precision(mode, True) if PREC == "product" else "fast") for off, mode, bw in targets]

def run(mode):
    chans = make()
    if mode in ("bank", "nopair"):  # nopair: one lane per workgroup throughout (the launch before lane pairs existed)
        bank = A.ChannelBank(chans)
        bank.pair_lanes = mode == "bank"
        zs = bank.process(raw, outs=outs, last_block=True, halo=(buf, 0))
        return zs, bank.last_launch
    A.Channelizer.lanes_for_groups = mode != "chained"  # chained: a filter's tap-row groups as passes over the capture, one after the other
    return [c.process(raw, out_dev=o, last_block=True, halo=(buf, 0)) for c, o in zip(chans, outs)], None

K, WARM = int(os.environ.get("K", "20")), int(os.environ.get("WARM", "5"))
keep = {}
for mode in modes:
    for _ in range(WARM):
        zs, info = run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        zs, info = run(mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    keep[mode] = [z.clone() for z in zs]
    algo = 4.0 * n_total + len(targets) * 8.0 * n_dec
    print(f"{which} {mode}: {ms:.3f} ms per capture of {n_total/1e6:.0f} M frames ({len(targets)} channels, D={d}) = {n_total/ms/1e6:.1f} GS/s of capture; "
          f"algorithmic {algo/1e9:.2f} GB -> {algo/ms/1e6:.0f} GB/s = {algo/ms/1e6/8000:.3f} of 8 TB/s; launch info {info}", flush=True)
if "bank" in keep and "single" in keep:
    e = 1024  # (the first and last outputs come from each channel's float32 kernel; a bank's common interior is a little shorter)
    print("bank vs single inside the common matrix-core interior: max |diff| =",
          max(float((a[e:-e] - b[e:-e]).abs().max()) for a, b in zip(keep["bank"], keep["single"])),
          "(same integer sums; the float rotation recurrence restarts per workgroup range)")
