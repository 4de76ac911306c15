"""ctypes binding of ``libiqa_hotpath.so`` (the C ABI declared in ``include/iqa_hotpath.h``).

There is deliberately **no CPU fallback**: if the HIP library is missing or fails to
load, :func:`lib` raises ``RuntimeError``.  Every function that reaches the device goes
through :func:`call`, which maps the library's status codes to the exceptions the
reference raises (``ValueError`` for bad configuration, ``RuntimeError`` otherwise).
"""
from __future__ import annotations

import ctypes
import subprocess
from ctypes import c_double, c_float, c_int32, c_int64, c_uint64, c_void_p
from pathlib import Path

import os

PKG_DIR = Path(__file__).resolve().parent
# (IQA_LIB: an experiment build of the same library -- profiles/ use it for A/B runs of build knobs)
LIB_PATH = Path(os.environ["IQA_LIB"]).resolve() if os.environ.get("IQA_LIB") else PKG_DIR / "libiqa_hotpath.so"
CSRC_DIR = PKG_DIR / "csrc"

IQA_OK, IQA_EINVAL, IQA_EHIP, IQA_ESTATE = 0, 1, 2, 3
FMT_S16, FMT_U8, FMT_F32 = 0, 1, 2
ORDER = {"iq": 0, "qi": 1, "iq_inv": 2, "qi_inv": 3}
ABI_VERSION = 1


class ChanParams(ctypes.Structure):
    """``iqa_chan_params`` (include/iqa_hotpath.h)."""

    _fields_ = [
        ("fmt", c_int32),
        ("ntaps", c_int32),
        ("decimation", c_int32),
        ("conj_sum", c_int32),
        ("rotate", c_int32),
        ("reserved", c_int32),
        ("rot_step", c_uint64),
        ("rot_base", c_uint64),
        ("out_scale_re", c_float),
        ("out_scale_im", c_float),
    ]


class MfmaParams(ctypes.Structure):
    """``iqa_mfma_params`` (include/iqa_hotpath.h)."""

    _fields_ = [("outputs_per_block", c_int32), ("reserved", c_int32), ("unit", c_double), ("c_re", c_double),
                ("c_im", c_double), ("debug_stamps", c_void_p), ("q_group", c_int32), ("k_first", c_int32),
                ("k_count", c_int32), ("finalize", c_int32), ("partial_in_dev", c_void_p), ("partial_out_dev", c_void_p)]


class MfmaLane(ctypes.Structure):
    """``iqa_mfma_lane`` (include/iqa_hotpath.h): one (channel, tap-row group) of a multi-lane launch."""

    _fields_ = [("afrag_dev", c_void_p), ("z_out_dev", c_void_p), ("partial_in_dev", c_void_p), ("partial_out_dev", c_void_p),
                ("unit", c_double), ("c_re", c_double), ("c_im", c_double), ("rot_step", c_uint64), ("rot_base", c_uint64),
                ("out_scale_re", c_float), ("out_scale_im", c_float), ("q_group", c_int32), ("finalize", c_int32),
                ("conj_sum", c_int32), ("rotate", c_int32), ("raw_partials", c_int32), ("reserved", c_int32)]


class DemodParams(ctypes.Structure):
    """``iqa_demod_params`` (include/iqa_hotpath.h)."""

    _fields_ = [("mode", c_int32), ("agc_enabled", c_int32), ("deemph_alpha", c_double), ("dc_radius", c_double),
                ("agc_target", c_double), ("agc_decay", c_double)]


DEMOD_MODE = {"nfm": 0, "fm": 0, "am": 1, "usb": 2, "ssb": 2, "lsb": 3}

_SIGNATURES = {
    "iqa_abi_version": (ctypes.c_int, []),
    "iqa_last_error": (ctypes.c_char_p, []),
    "iqa_taps_padded_len": (c_int64, [c_int32]),
    "iqa_channelize": (ctypes.c_int, [ctypes.POINTER(ChanParams), c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                      c_int64, c_int64, c_void_p, c_void_p]),
    "iqa_mfma_afrag_bytes": (c_int64, [c_int32]),
    "iqa_mfma_ring_bytes": (c_int64, [c_int32]),
    "iqa_mfma_ring_lds_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "iqa_mfma_ring_mode": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "iqa_channelize_mfma": (ctypes.c_int, [ctypes.POINTER(ChanParams), ctypes.POINTER(MfmaParams), c_void_p, c_void_p,
                                           c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "iqa_channelize_mfma_multi": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, ctypes.POINTER(MfmaLane), c_int32,
                                                 c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "iqa_channelize_mfma_pairs": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, ctypes.POINTER(MfmaLane), c_int32,
                                                 c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "iqa_mfma_ring_pairs": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "iqa_mfma_ring_lanes": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "iqa_mfma_combine": (ctypes.c_int, [ctypes.POINTER(ChanParams), ctypes.POINTER(c_void_p), c_int32, ctypes.POINTER(c_double),
                                        c_int64, c_int64, c_void_p, c_void_p]),
    "iqa_history_update": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "iqa_oscillator_mix": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_int64, c_double, c_double, c_void_p, c_void_p]),
    "iqa_decimate": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    "iqa_mean_power": (ctypes.c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "iqa_mean_power_batch": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "iqa_raw_level": (ctypes.c_int, [c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "iqa_quadrature": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "iqa_envelope": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "iqa_real_part": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "iqa_scan_workspace_bytes": (c_int64, [c_int64]),
    "iqa_deemphasis": (ctypes.c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_dc_block": (ctypes.c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_agc": (ctypes.c_int, [c_void_p, c_int64, c_double, c_double, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "iqa_demodulate": (ctypes.c_int, [ctypes.POINTER(DemodParams), c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_demodulate_from_reset": (ctypes.c_int, [ctypes.POINTER(DemodParams), c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_writer_clip": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "iqa_resample": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64, c_void_p,
                                    c_void_p, c_void_p]),
    "iqa_float_to_pcm16": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "iqa_trickle_copy": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "iqa_f32_to_s16_exact": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "iqa_f32_split_s16": (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_psd_frames": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                      c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iqa_pair_average_rows": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None


def build(force: bool = False) -> Path:
    """Compile the gfx950 library in-tree with hipcc (cross-compiles without a GPU)."""
    srcs = list(CSRC_DIR.glob("*.hip")) + list(CSRC_DIR.glob("*.h")) + [PKG_DIR.parent / "include" / "iqa_hotpath.h"]
    stale = not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(CSRC_DIR), "-s", "-j4"], check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"HIP hot-path library not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7; import it first so that our library binds to the
    # runtime instance that owns torch's device allocations and streams.
    import torch  # noqa: F401

    try:
        handle = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:
        raise RuntimeError(f"failed to load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if handle.iqa_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {handle.iqa_abi_version()} != binding {ABI_VERSION}")
    _lib = handle
    return handle


def call(name: str, *args) -> None:
    """Invoke an ``int``-returning entry point and raise the reference's exception types."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc == IQA_OK:
        return
    msg = (handle.iqa_last_error() or b"").decode("utf-8", "replace")
    if rc == IQA_EINVAL:
        raise ValueError(f"{name}: {msg}")
    raise RuntimeError(f"{name}: {msg}")


_torch_with_gpu = None


def require_gpu():
    """Device 0..N-1 visible through torch; raises RuntimeError otherwise (no fallback).  The positive answer is
    remembered: ``torch.cuda.is_available()`` costs ~8 us and this is called from every helper."""
    global _torch_with_gpu
    if _torch_with_gpu is not None:
        return _torch_with_gpu
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible to this process: the HIP hot path has no CPU fallback")
    _torch_with_gpu = torch
    return torch


def ptr(t) -> c_void_p:
    """Device pointer of a torch tensor (or NULL for None)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream_ptr() -> c_void_p:
    """The current torch stream of the current device as a raw ``hipStream_t`` (the direct binding: building a
    ``torch.cuda.Stream`` object for every native call costs more than some of the launches)."""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return c_void_p(raw(torch._C._cuda_getDevice()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
