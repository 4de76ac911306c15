"""iq-to-audio hot path, MI355X-native.

Drop-in for the DSP surface of the reference's ``iq_to_audio.processing`` and
``iq_to_audio.decoders`` modules; all arithmetic runs in hand-written gfx950 HIP kernels
behind the C ABI in ``include/iqa_hotpath.h``.  Importing the package does not touch the GPU
or the native library; the first stage call does (and raises if either is missing).

The directory is named ``iq-to-audio_amd``; import it as ``iq_to_audio_amd``.
"""
from __future__ import annotations

__version__ = "0.1.0"

import os as _os

# The batch path keeps three streams beside the caller's (egress, aux, and RCCL's own in a multi-GPU job).  The HIP
# runtime maps streams onto four hardware queues by default, so a fifth stream shares a queue with another one -- measured
# with RCCL in the process: the aux stream landed on the compute stream's queue, its launches ran after the resampler
# instead of beside it, +4.5 % per capture.  Read by the runtime when it initialises (the first HIP call), so setting
# it here works unless the application has already used the GPU; an explicit setting wins.  Sixteen: the batch path's named
# side streams (_dev.side_stream: aux, egress, tail, edge, upload, graph0..3) plus the caller's and the graph-capture stream
# are eleven -- with eight queues the four graph-replay streams of a long-lived process shared queues two by two and config 1
# ran at 55 us per capture instead of 39 (bench.py after its other sub-benches; 12 and 16 queues measured alike).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _native as native  # noqa: F401
from .decoders import Decoder, DecoderStats, create_decoder  # noqa: F401
from .processing import (  # noqa: F401
    ChannelBank,
    Channelizer,
    ComplexOscillator,
    Decimator,
    MultiChannelPipeline,
    OverlapSaveFIR,
    ProcessingCancelled,
    ProcessingConfig,
    ProcessingPipeline,
    ProcessingResult,
    choose_mix_sign,
    design_channel_filter,
    tune_chunk_size,
)
