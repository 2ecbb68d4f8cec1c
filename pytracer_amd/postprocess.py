"""HdrImage post-processing on the device: what pytracer's ``main.py:203-213``
does after ``fire_all_rays`` — ``write_pfm``, ``average_luminosity``, ``normalize_image``,
``clamp_image``, ``write_ldr_image`` (hdrimages.py:96-171) — on a frame that sits in HBM.

``DeviceImage`` wraps a ``[H, W, 3]`` torch tensor (fp32 or fp64, row 0 on top) on the GPU and mirrors
the reference method names; torch only provides the memory, every operation is a HIP kernel behind the
C-ABI (``pt_image_*``).  PNG encoding itself stays on the host (Pillow), as in the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, abi

LITTLE_ENDIAN, BIG_ENDIAN = 1, 2  # hdrimages.py:26-30 (Endianness values)


class DeviceImage:
    def __init__(self, tensor: torch.Tensor):
        if tensor.dim() != 3 or tensor.shape[2] != 3 or not tensor.is_cuda or not tensor.is_contiguous():
            raise ValueError("expected a contiguous [H, W, 3] CUDA tensor")
        if tensor.dtype not in (torch.float32, torch.float64):
            raise ValueError("expected float32 or float64 pixels")
        self.t = tensor
        self.height, self.width = int(tensor.shape[0]), int(tensor.shape[1])
        self.fmt = abi.OUT_F32 if tensor.dtype == torch.float32 else abi.OUT_F64
        self.device = tensor.device.index or 0

    @classmethod
    def from_numpy(cls, arr, device: int = 0) -> "DeviceImage":
        a = np.ascontiguousarray(arr)
        return cls(torch.from_numpy(a).to(f"cuda:{device}").contiguous())

    def numpy(self) -> np.ndarray:
        return self.t.cpu().numpy()

    # -- hdrimages.py:96-118 ---------------------------------------------------------------------------
    def pfm_payload(self, endianness: int = LITTLE_ENDIAN) -> bytes:
        out = torch.empty(self.width * self.height * 12, dtype=torch.uint8, device=self.t.device)
        _lib.check(_lib.lib().pt_image_pack_pfm(self.device, C.c_void_p(self.t.data_ptr()), self.fmt, self.width,
                                                self.height, int(endianness == BIG_ENDIAN),
                                                C.c_void_p(out.data_ptr()), None))
        return out.cpu().numpy().tobytes()

    def write_pfm(self, stream, endianness: int = LITTLE_ENDIAN) -> None:
        endianness_str = "-1.0" if endianness == LITTLE_ENDIAN else "1.0"
        stream.write(f"PF\n{self.width} {self.height}\n{endianness_str}\n".encode("ascii"))
        stream.write(self.pfm_payload(endianness))

    # -- hdrimages.py:120-146 --------------------------------------------------------------------------
    def average_luminosity(self, delta: float = 1e-10) -> float:
        out = C.c_double(0.0)
        _lib.check(_lib.lib().pt_image_average_luminosity(self.device, C.c_void_p(self.t.data_ptr()), self.fmt,
                                                          self.width, self.height, float(delta), C.byref(out), None))
        return float(out.value)

    def _tonemap(self, scale: float, clamp: bool, gamma: float, rgb8: Optional[torch.Tensor], write_back: bool):
        _lib.check(_lib.lib().pt_image_tonemap(self.device, C.c_void_p(self.t.data_ptr()), self.fmt, self.width,
                                               self.height, float(scale), int(clamp), float(gamma),
                                               C.c_void_p(rgb8.data_ptr()) if rgb8 is not None else None,
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
        rgb8 = torch.empty((self.height, self.width, 3), dtype=torch.uint8, device=self.t.device)
        self._tonemap(1.0, False, gamma, rgb8, False)
        return rgb8.cpu().numpy()

    def write_ldr_image(self, stream, format: str, gamma: float = 1.0) -> None:
        from PIL import Image  # host-side encoder, as in the reference

        Image.fromarray(self.ldr_bytes(gamma), mode="RGB").save(stream, format=format)
