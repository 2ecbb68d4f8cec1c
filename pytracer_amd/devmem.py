"""Device memory and streams through the C-ABI (``pt_device_alloc`` / ``pt_stream_create``, ABI 1.5): what the resident frame of
the ``render`` command and the frames of a :class:`pytracer_amd.pipeline.FramePipeline` live in when the caller brings no GPU
framework of its own.  The reference keeps its frame in a Python list (hdrimages.py:70); here it stays in HBM between the
render kernel and the post-processing kernels, and this is the buffer that takes.  Nothing here imports torch."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib


class Stream:
    """A non-blocking HIP stream on ``device`` (``pt_stream_create``); ``handle`` is what ``pt_render_device`` takes."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _lib.check(_lib.lib().pt_stream_create(int(device), C.byref(h)))
        self.device, self.handle = int(device), h.value

    @property
    def cuda_stream(self) -> int:  # (the attribute torch's streams carry the same handle under)
        return self.handle

    def synchronize(self) -> None:
        if self.handle is not None:
            _lib.check(_lib.lib().pt_stream_sync(self.device, C.c_void_p(self.handle)))

    def close(self) -> None:
        if self.handle is not None:
            h, self.handle = self.handle, None
            _lib.check(_lib.lib().pt_stream_destroy(self.device, C.c_void_p(h)))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class DeviceBuffer:
    """``shape`` values of ``dtype`` in the HBM of ``device``: ``data_ptr()`` for the C-ABI, ``numpy()`` to copy it to the host."""

    is_device_buffer = True

    def __init__(self, shape: Tuple[int, ...], dtype=np.float64, device: int = 0):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(_lib.lib().pt_device_alloc(self.device, self.nbytes, C.byref(p)))
        self._ptr: Optional[int] = p.value

    def data_ptr(self) -> int:
        if self._ptr is None and self.nbytes:
            raise RuntimeError("DeviceBuffer used after free()")
        return self._ptr or 0

    def numpy(self, stream: Optional[Stream] = None) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(_lib.lib().pt_device_download(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.data_ptr()), self.nbytes,
                                                 C.c_void_p(stream.handle) if stream is not None else None))
        return out

    def free(self) -> None:
        if self._ptr is not None:
            p, self._ptr = self._ptr, None
            _lib.check(_lib.lib().pt_device_free(self.device, C.c_void_p(p)))

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass
