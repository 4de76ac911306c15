"""Small device-memory helpers.  PyTorch is used for allocation, streams and copies only;
all arithmetic happens in the HIP library behind ``_native``."""
from __future__ import annotations

import numpy as np

from . import _native as N


def torch_mod():
    return N.require_gpu()


class on_stream:
    """``with on_stream(s, back):`` -- make ``s`` the current torch stream and ``back`` afterwards.  For callers that
    know which stream is current (``torch.cuda.stream`` looks it up first, ~10 us per block on this host)."""

    __slots__ = ("stream", "back")

    def __init__(self, stream, back):
        self.stream, self.back = stream, back

    def __enter__(self):
        torch_mod().cuda.set_stream(self.stream)
        return self.stream

    def __exit__(self, *exc):
        torch_mod().cuda.set_stream(self.back)
        return False


_SIDE_STREAMS: dict = {}


def side_stream(role: str, index: int | None = None):
    """The process-wide side stream for ``role`` on the current device ("aux", "egress", "tail", "graph0", ...), made on
    first request and shared by every runner.  Why shared rather than one per runner: the HIP runtime multiplexes streams
    onto a few hardware queues (``GPU_MAX_HW_QUEUES``, 4 by default), the least-used queue at a stream's first use -- a
    runner whose side stream lands on the queue of its compute stream loses the overlap it wanted, and which queue a NEW
    stream gets depends on every stream the process has used before (measured: config 4's unit 1.17 -> 1.33 ms, config 3
    13.6 -> 15.3 ms per capture after an unrelated runner had taken four more streams).  A fixed,
    small set of streams keeps the mapping the same for every runner of the process.  Sharing costs nothing in
    correctness (stream order only adds dependencies) and runners are used one at a time."""
    torch = torch_mod()
    key = (torch.cuda.current_device() if index is None else int(index), role)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key[0])
    return s


def current_raw_stream() -> int:
    """``hipStream_t`` of the current torch stream, as an integer."""
    torch = torch_mod()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return int(raw(torch._C._cuda_getDevice()))
    return int(torch.cuda.current_stream().cuda_stream)


def device(index: int | None = None):
    torch = torch_mod()
    return torch.device("cuda", torch.cuda.current_device() if index is None else index)


def is_tensor(x) -> bool:
    try:
        import torch

        return isinstance(x, torch.Tensor)
    except Exception:
        return False


def to_device(x, dtype: str):
    """numpy array / host tensor / device tensor -> contiguous device tensor of ``dtype``
    ('complex64', 'float32', 'int16', 'uint8', 'int64', 'float64')."""
    torch = torch_mod()
    td = getattr(torch, dtype)
    if is_tensor(x):
        t = x
        if t.dtype != td:
            t = t.to(td)
        if not t.is_cuda:
            t = t.to(device(), non_blocking=False)
        return t.contiguous()
    arr = np.ascontiguousarray(np.asarray(x), dtype=np.dtype(dtype))
    if arr.size == 0:
        return torch.empty(0, dtype=td, device=device())
    return _upload(torch, arr)


# Small host arrays are staged through a pool of persistent pinned buffers and copied asynchronously:
# a pageable `.to(device)` synchronises the stream (the host could not run ahead of the GPU), and
# `Tensor.pin_memory()` costs ~2.6 ms per call on this stack.  A slot is reused once the event recorded
# after its last copy has completed.
_PIN_LIMIT = 1 << 20
_pin_pool: dict[int, list] = {}


def _upload(torch, arr: np.ndarray):
    nbytes = arr.nbytes
    if nbytes > _PIN_LIMIT:
        if not arr.flags.writeable:  # (shared read-only plans: torch warns about tensors over read-only memory; this one is only copied from)
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return torch.from_numpy(arr).to(device())
        return torch.from_numpy(arr).to(device())
    cls = max(256, 1 << (nbytes - 1).bit_length())
    slots = _pin_pool.setdefault(cls, [])
    slot = None
    for cand in slots:
        if cand[1] is None or cand[1].query():
            slot = cand
            break
    if slot is None:
        slot = [torch.empty(cls, dtype=torch.uint8).pin_memory(), None]
        slots.append(slot)
    staged = slot[0][:nbytes]
    staged.numpy()[:] = arr.reshape(-1).view(np.uint8)
    dev = torch.empty(nbytes, dtype=torch.uint8, device=device())
    dev.copy_(staged, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slot[1] = ev
    return dev.view(getattr(torch, arr.dtype.name)).reshape(arr.shape)


def like_input(result, template):
    """Return ``result`` (device tensor) in the container type of ``template``:
    numpy in -> numpy out (as the reference's stages do), tensor in -> device tensor out."""
    if is_tensor(template):
        return result
    return result.cpu().numpy()


def empty(n: int, dtype: str):
    torch = torch_mod()
    return torch.empty(int(n), dtype=getattr(torch, dtype), device=device())


def zeros(n: int, dtype: str):
    torch = torch_mod()
    return torch.zeros(int(n), dtype=getattr(torch, dtype), device=device())


def from_numpy(arr: np.ndarray):
    torch = torch_mod()
    arr = np.ascontiguousarray(arr)
    if arr.size == 0:
        return torch.empty(0, dtype=torch.from_numpy(arr).dtype, device=device())
    return _upload(torch, arr)
