"""Scratch: BASELINE config 1's step (bench.sub_bench_c1) for several numbers of channelizer workgroups per launch."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
from iq_to_audio_amd import processing as PR
for blocks in [int(b) for b in sys.argv[1:]] or [256, 512, 768]:
    PR._ChannelKernel.launch_blocks = blocks
    PR._KERNEL_CACHE.clear()
    r = bench.sub_bench_c1(steps=200, warm=50)
    print(blocks, "workgroups:", json.dumps({k: r[k] for k in ("ms_per_step", "ms_per_step_direct_launches")}),
          "channelizer", r["roofline"]["kernel_ms"], "ms; parity", r["parity"]["rms_err_vs_oracle_fs_channel"], flush=True)
