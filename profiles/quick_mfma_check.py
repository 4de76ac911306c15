"""Scratch: MFMA channelizer timing experiments on the full C2 workload (debug flags skip parts of the kernel)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import iq_to_audio_amd as A
from iq_to_audio_amd import _dev as D, processing as PR
from iq_to_audio_amd.benchmark import synthetic_iq_s16

fs, d, bw, f_off = 10e6, 104, 12500., 25e3
n_total = 600_000_000
host = synthetic_iq_s16(fs, 1.0, f_off).reshape(-1)
raw = torch.from_numpy(host).to("cuda").repeat(60)[: 2 * n_total].contiguous()
taps = A.design_channel_filter(fs, bw, d)
z = D.empty(-(-n_total // d), "complex64")
import os
outs = {}
PR._ChannelKernel.ring_acc32 = os.environ.get('ACC32', '1') == '1'  # the ablation instantiations exist for the int32 sums
for dbg in [int(x) for x in os.environ.get('DBG', '0,4,12').split(',')]:
    PR._ChannelKernel.mfma_variant = {0: "plain", 64: "ring"}[dbg & 64]
    PR._KERNEL_CACHE.clear()
    ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
    ch.plan_ahead(); ch._kernel.mfma_params[0].reserved |= (dbg & ~64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for it in range(4):
        ch.consumed = 0; ch._hist = None
        ch.process(raw, out_dev=z, events=(e0, e1)); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    if not (dbg & ~64): outs[dbg] = z.clone()
    print(f"debug={dbg:3d} ({PR._ChannelKernel.mfma_variant}; skip: {'scatter ' if dbg&1 else ''}{'dma ' if dbg&16 else ''}{'mfma' if dbg&32 else ''}): kernel ms {[round(t,3) for t in ts]}")
print('variants bitwise equal:', {k: bool(torch.equal(outs[k], list(outs.values())[0])) for k in outs}, 'max abs diff vs first:', {k: float((outs[k] - list(outs.values())[0]).abs().max()) for k in outs})
if os.environ.get('NOSTAMPS'): sys.exit(0)
PR._ChannelKernel.mfma_variant = 'plain'
PR._KERNEL_CACHE.clear()
# cycle anatomy from in-kernel stamps (diagnostic build path, debug bit 1)
ch = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d)
ch.plan_ahead()
nblk = -(-(n_total // d) // 512) + 2
st = torch.zeros(nblk * 12 * 8, dtype=torch.int64, device="cuda")
ch._kernel.mfma_params[0].reserved = 2
ch._kernel.mfma_params[0].debug_stamps = st.data_ptr()
ch.process(raw, out_dev=z); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 8)
s = s[s[:, 4] > 0]
print("outputs/block:", ch._kernel.mfma_params[0].outputs_per_block); print("waves:", len(s), "median cycles: prologue %d, loop %d (scatter %d), tail %d, tiles/wave %d" % tuple(np.median(s[:, i]) for i in range(5)))
print("per tile: loop %.0f cycles, of which scatter %.0f; ideal MFMA per tile %d" % (np.median(s[:, 1] / s[:, 4]), np.median(s[:, 2] / s[:, 4]), 7 * 12 * 32))

