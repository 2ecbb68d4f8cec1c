"""HdrImage post-processing on the device: what pytracer's ``main.py:203-213``
does after ``fire_all_rays`` — ``write_pfm``, ``average_luminosity``, ``normalize_image``,
``clamp_image``, ``write_ldr_image`` (hdrimages.py:96-171) — on a frame that sits in HBM.

``DeviceImage`` wraps ``[H, W, 3]`` pixels (fp32 or fp64, row 0 on top) — a frame left in HBM by ``pt_render_device``
(a :class:`pytracer_amd.devmem.DeviceBuffer`, which needs no torch, or a CUDA torch tensor) or a host numpy array — and
mirrors the reference method names; every operation is
a HIP kernel behind the C-ABI (``pt_image_*``).  PNG encoding itself stays on the host (Pillow), as in
the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib, abi

LITTLE_ENDIAN, BIG_ENDIAN = 1, 2  # hdrimages.py:26-30 (Endianness values)


class DeviceImage:
    """``[H, W, 3]`` fp32/fp64 pixels: a ``DeviceBuffer`` or a CUDA torch tensor (frame resident in HBM) or a numpy array
    (host; the C-ABI stages it through the device).  Either way every operation below is a HIP kernel."""

    def __init__(self, pixels, device: int = 0):
        self.is_buffer = bool(getattr(pixels, "is_device_buffer", False))
        self.is_torch = hasattr(pixels, "data_ptr") and not self.is_buffer
        if self.is_buffer:
            if len(pixels.shape) != 3 or pixels.shape[2] != 3 or pixels.dtype not in (np.float32, np.float64):
                raise ValueError("expected a [H, W, 3] float32/float64 DeviceBuffer")
            f32 = pixels.dtype == np.float32
            self.device = pixels.device
        elif self.is_torch:
            if pixels.dim() != 3 or pixels.shape[2] != 3 or not pixels.is_cuda or not pixels.is_contiguous():
                raise ValueError("expected a contiguous [H, W, 3] CUDA tensor")
            f32 = str(pixels.dtype) == "torch.float32"
            if not f32 and str(pixels.dtype) != "torch.float64":
                raise ValueError("expected float32 or float64 pixels")
            self.device = pixels.device.index or 0
        else:
            pixels = np.ascontiguousarray(pixels)
            if pixels.ndim != 3 or pixels.shape[2] != 3 or pixels.dtype not in (np.float32, np.float64):
                raise ValueError("expected a [H, W, 3] float32/float64 array")
            f32 = pixels.dtype == np.float32
            self.device = device
        self.t = pixels
        self.height, self.width = int(pixels.shape[0]), int(pixels.shape[1])
        self.fmt = abi.OUT_F32 if f32 else abi.OUT_F64

    @classmethod
    def from_numpy(cls, arr, device: int = 0) -> "DeviceImage":
        return cls(np.array(arr, order="C"), device)

    def _ptr(self):
        return C.c_void_p(self.t.data_ptr() if (self.is_torch or self.is_buffer) else self.t.ctypes.data)

    def numpy(self) -> np.ndarray:
        if self.is_buffer:
            return self.t.numpy()
        return self.t.cpu().numpy() if self.is_torch else self.t

    # -- hdrimages.py:96-118 ---------------------------------------------------------------------------
    def pfm_payload(self, endianness: int = LITTLE_ENDIAN) -> bytes:
        out = np.empty(self.width * self.height * 12, dtype=np.uint8)
        _lib.check(_lib.lib().pt_image_pack_pfm(self.device, self._ptr(), self.fmt, self.width, self.height,
                                                int(endianness == BIG_ENDIAN), out.ctypes.data_as(C.c_void_p), None))
        return out.tobytes()

    def write_pfm(self, stream, endianness: int = LITTLE_ENDIAN) -> None:
        endianness_str = "-1.0" if endianness == LITTLE_ENDIAN else "1.0"
        stream.write(f"PF\n{self.width} {self.height}\n{endianness_str}\n".encode("ascii"))
        stream.write(self.pfm_payload(endianness))

    # -- hdrimages.py:120-146 --------------------------------------------------------------------------
    def average_luminosity(self, delta: float = 1e-10) -> float:
        out = C.c_double(0.0)
        _lib.check(_lib.lib().pt_image_average_luminosity(self.device, self._ptr(), self.fmt, self.width, self.height,
                                                          float(delta), C.byref(out), None))
        return float(out.value)

    def _tonemap(self, scale: float, clamp: bool, gamma: float, rgb8: Optional[np.ndarray], write_back: bool):
        _lib.check(_lib.lib().pt_image_tonemap(self.device, self._ptr(), self.fmt, self.width, self.height,
                                               float(scale), int(clamp), float(gamma),
                                               rgb8.ctypes.data_as(C.c_void_p) if rgb8 is not None else None,
                                               int(write_back), None))

    def normalize_image(self, factor: float, luminosity: Optional[float] = None) -> None:
        if not luminosity:
            luminosity = self.average_luminosity()
        self._tonemap(factor / luminosity, False, 1.0, None, True)

    def clamp_image(self) -> None:
        self._tonemap(1.0, True, 1.0, None, True)

    # -- hdrimages.py:148-171 ----------------------------------------------------------------------------
    def ldr_bytes(self, gamma: float = 1.0) -> np.ndarray:
        """``[H, W, 3]`` uint8: int(255 * pow(c, 1/gamma)) per channel (no change to the image)."""
        rgb8 = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._tonemap(1.0, False, gamma, rgb8, False)
        return rgb8

    def write_ldr_image(self, stream, format: str, gamma: float = 1.0) -> None:
        from PIL import Image  # host-side encoder, as in the reference

        Image.fromarray(self.ldr_bytes(gamma), mode="RGB").save(stream, format=format)
