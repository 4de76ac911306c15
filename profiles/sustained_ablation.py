"""Ring-kernel ablations in the SUSTAINED (power-limited) regime: every variant runs N back-to-back launches and the
median of the second half is reported, together with rocm-smi's clock/power at that point.  Debug bits (KS = 7 or, with FS=20e6, 13; int32
sums): 1 = no scatter, 16 = no DMA, 32 = no matrix work, 4 = every second multiplying wave without fragment reads and byte
splits, 8 = no byte splits.  Usage: python profiles/sustained_ablation.py [N]"""
import re, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import os
import iq_to_audio_amd as A
from iq_to_audio_amd import _native as NATIVE
if os.environ.get("IQA_LIB"):  # an experiment build of the library (e.g. another cache policy for the LDS-DMA loads)
    NATIVE.LIB_PATH = Path(os.environ["IQA_LIB"]).resolve()
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
fs = float(os.environ.get("FS", "10e6"))  # 10e6: D = 104, 7 k steps (config 2); 20e6: D = 208, 13 k steps (configs 3, 4)
d, bw, f_off = int(round(fs / 96153.846)), 12500., 25e3
n_total = int(fs * 60)
host = synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)
raw = torch.from_numpy(host).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, bw, d)
z = D.empty(-(-n_total // d), "complex64")
PR._ChannelKernel.mfma_variant = "ring"
PR._ChannelKernel.launch_blocks = int(os.environ.get("RING_BLOCKS", "256"))  # fewer than 256: some CUs stay free
PR._ChannelKernel.ring_acc32 = os.environ.get("ACC32", "1") == "1"  # 0: int64 sums, 16-bit taps (no ablation builds: DBGS=0)

def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        p = re.search(r"Power \(W\): ([\d.]+)", t); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz", t)
        return f"{p.group(1) if p else '?'} W, sclk {c.group(1) if c else '?'} MHz"
    except Exception as e:
        return f"rocm-smi failed: {e}"

names = {0: "everything", 1: "no scatter", 16: "no DMA", 32: "no matrix work", 33: "DMA + LDS reads only", 17: "matrix work + LDS reads only",
         4: "half the fragment reads + splits", 8: "no byte splits", 5: "half reads + splits, no scatter", 13: "half reads, no splits, no scatter"}
DBGS = [int(v) for v in os.environ.get("DBGS", "0,1,16,32,33,17").split(",")]
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for dbg in DBGS:
        PR._KERNEL_CACHE.clear()
        ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
        ch.plan_ahead(); ch._kernel.mfma_params[0].reserved |= dbg
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
        for i in range(N):
            ch.consumed = 0; ch._hist = None
            ch.process(raw, out_dev=z, events=evs[i], last_block=True)
            if i == N - 200:
                pass
        status = smi()  # sampled while the queue is still draining
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) for a, b in evs])
        print(f"round {rnd} {names[dbg]:30s}: first 5 {np.round(ts[:5], 3)}  median of last half {np.median(ts[N // 2:]):.4f} ms  [{status}]", flush=True)
