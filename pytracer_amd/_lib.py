"""Loader for libptrace.so (the HIP product library).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

# PTRACE_LIB lets kernel A/B experiments load an alternative build of the same C-ABI library
_LIB_PATH = os.environ.get("PTRACE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptrace.so")
_lib = None

# every symbol include/ptrace.h declares
EXPORTS = ("pt_device_count", "pt_scene_upload", "pt_scene_clone", "pt_scene_free", "pt_rows_for_rank", "pt_output_bytes",
           "pt_render", "pt_render_device", "pt_get_stats", "pt_set_count_rays", "pt_sync", "pt_last_error",
           "pt_version", "pt_profile_begin", "pt_profile_end", "pt_set_timing", "pt_image_pack_pfm",
           "pt_image_average_luminosity", "pt_image_tonemap", "pt_host_alloc", "pt_host_free", "pt_set_dome_shortcut", "pt_device_info",
           "pt_image_sparse_fixed_bytes", "pt_image_sparse_encode", "pt_image_sparse_decode", "pt_image_sparse_decode_many", "pt_device_kernargs",
           "pt_device_alloc", "pt_device_free", "pt_device_download", "pt_stream_create", "pt_stream_sync", "pt_stream_destroy")


# every symbol include/ptrace_debug.h declares for ordinary builds (diagnostics: not part of the boundary)
DEBUG_EXPORTS = ("pt_debug_probe", "pt_debug_cull_probe", "pt_debug_hit_probe", "pt_debug_lanes_probe",
                 "pt_debug_camera_probe", "pt_debug_scatter_probe", "pt_debug_read_queue", "pt_debug_plan", "pt_debug_plan_scene",
                 "pt_debug_set_tuning", "pt_debug_get_tuning", "pt_debug_handed_over")


# include/ptrace.h: pt_version() = major << 16 | minor; abi.Stats mirrors the 56-byte pt_stats of minor >= 2, the tracer's
# default alignment needs the PT_PCG_SEQ of minor >= 3, device.device_kernargs() the entry point of minor 4
# ... devmem.DeviceBuffer / Stream the entry points of minor 5
ABI_MAJOR, ABI_MINOR_NEEDED = 1, 5


class PtraceError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{abi.ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


def lib_path() -> str:
    return _LIB_PATH


_share_with_torch = True


def standalone() -> None:
    """Call BEFORE the library is first loaded, from a process that will not use torch at all (the ``render`` command): the
    library then binds to the system's HIP runtime (/opt/rocm) and torch's bundled copy is not preloaded.  Such a process must
    not import torch afterwards (two HIP runtimes in one process: the second finds no GPU)."""
    global _share_with_torch
    if _lib is None:
        _share_with_torch = False


def _share_hip_runtime_with_torch() -> None:
    """One process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own ``libamdhip64.so`` (soname
    ``libamdhip64.so.7``, the soname libptrace.so needs too).  If libptrace.so is loaded first, the dynamic
    loader binds it to /opt/rocm's copy and a later ``import torch`` brings a second runtime that finds no
    GPU ("No HIP GPUs are available").  Preloading torch's copy — without importing torch — makes both
    share it whichever comes first."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """The loaded C-ABI library; raises if it has not been built (``python -m pytracer_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -m pytracer_amd.build` (needs hipcc); there is no CPU fallback.")
        if _share_with_torch:
            _share_hip_runtime_with_torch()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.pt_device_count.restype = C.c_int
        L.pt_device_info.restype = C.c_int
        L.pt_device_info.argtypes = [C.c_int, P(C.c_int), P(C.c_int)]
        L.pt_scene_upload.restype = C.c_int
        L.pt_scene_upload.argtypes = [P(abi.SceneDesc), C.c_int, P(C.c_void_p)]
        L.pt_scene_clone.restype = C.c_int
        L.pt_scene_clone.argtypes = [C.c_void_p, P(C.c_void_p)]
        L.pt_scene_free.restype = None
        L.pt_scene_free.argtypes = [C.c_void_p]
        L.pt_rows_for_rank.restype = C.c_int
        L.pt_rows_for_rank.argtypes = [P(abi.Params)]
        L.pt_output_bytes.restype = C.c_size_t
        L.pt_output_bytes.argtypes = [P(abi.Params)]
        L.pt_render.restype = C.c_int
        L.pt_render.argtypes = [C.c_void_p, P(abi.Camera), P(abi.Params), C.c_void_p, C.c_size_t]
        L.pt_render_device.restype = C.c_int
        L.pt_render_device.argtypes = [C.c_void_p, P(abi.Camera), P(abi.Params), C.c_void_p, C.c_size_t,
                                       C.c_void_p]
        L.pt_get_stats.restype = C.c_int
        L.pt_get_stats.argtypes = [C.c_void_p, P(abi.Stats)]
        L.pt_set_dome_shortcut.restype = C.c_int
        L.pt_set_dome_shortcut.argtypes = [C.c_void_p, C.c_int]
        L.pt_set_count_rays.restype = C.c_int
        L.pt_set_count_rays.argtypes = [C.c_void_p, C.c_int]
        L.pt_sync.restype = C.c_int
        L.pt_sync.argtypes = [C.c_void_p]
        L.pt_last_error.restype = C.c_int
        L.pt_last_error.argtypes = [C.c_char_p, C.c_size_t]
        L.pt_version.restype = C.c_int
        ver = int(L.pt_version())
        if ver >> 16 != ABI_MAJOR or (ver & 0xFFFF) < ABI_MINOR_NEEDED:
            raise ImportError(f"{_LIB_PATH} implements ABI {ver >> 16}.{ver & 0xFFFF}; this package needs {ABI_MAJOR}.>={ABI_MINOR_NEEDED} "
                              "(pt_stats layout, PT_PCG_SEQ): rebuild with `python -m pytracer_amd.build --force`")
        L.pt_device_kernargs.restype = C.c_int
        L.pt_set_timing.restype = C.c_int
        L.pt_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.pt_profile_begin.restype = C.c_int
        L.pt_profile_begin.argtypes = [C.c_void_p, C.c_int]
        L.pt_profile_end.restype = C.c_int
        L.pt_profile_end.argtypes = [C.c_void_p, P(C.c_double), P(C.c_int)]
        L.pt_image_pack_pfm.restype = C.c_int
        L.pt_image_pack_pfm.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pt_image_average_luminosity.restype = C.c_int
        L.pt_image_average_luminosity.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                                  P(C.c_double), C.c_void_p]
        L.pt_image_sparse_fixed_bytes.restype = C.c_longlong
        L.pt_image_sparse_fixed_bytes.argtypes = [C.c_longlong, C.c_int]
        L.pt_image_sparse_encode.restype = C.c_int
        L.pt_image_sparse_encode.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_image_sparse_decode.restype = C.c_int
        L.pt_image_sparse_decode.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pt_image_sparse_decode_many.restype = C.c_int
        L.pt_image_sparse_decode_many.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pt_image_tonemap.restype = C.c_int
        L.pt_image_tonemap.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                       C.c_void_p, C.c_int, C.c_void_p]
        L.pt_host_alloc.restype = C.c_int
        L.pt_host_alloc.argtypes = [C.c_size_t, P(C.c_void_p)]
        L.pt_host_free.restype = C.c_int
        L.pt_host_free.argtypes = [C.c_void_p]
        L.pt_device_alloc.restype = C.c_int
        L.pt_device_alloc.argtypes = [C.c_int, C.c_size_t, P(C.c_void_p)]
        L.pt_device_free.restype = C.c_int
        L.pt_device_free.argtypes = [C.c_int, C.c_void_p]
        L.pt_device_download.restype = C.c_int
        L.pt_device_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.pt_stream_create.restype = C.c_int
        L.pt_stream_create.argtypes = [C.c_int, P(C.c_void_p)]
        L.pt_stream_sync.restype = C.c_int
        L.pt_stream_sync.argtypes = [C.c_int, C.c_void_p]
        L.pt_stream_destroy.restype = C.c_int
        L.pt_stream_destroy.argtypes = [C.c_int, C.c_void_p]
        # diagnostics (include/ptrace_debug.h): bound when the build carries them -- a renderer never needs one
        if hasattr(L, "pt_debug_cull_probe"):
            L.pt_debug_cull_probe.restype = C.c_int
            L.pt_debug_cull_probe.argtypes = [C.c_void_p, P(abi.Camera)] + [C.c_int] * 8 + [C.c_void_p]
        if hasattr(L, "pt_debug_probe"):
            L.pt_debug_probe.restype = C.c_int
            L.pt_debug_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "pt_debug_hit_probe"):
            L.pt_debug_hit_probe.restype = C.c_int
            L.pt_debug_hit_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "pt_debug_lanes_probe"):
            L.pt_debug_lanes_probe.restype = C.c_int
            L.pt_debug_lanes_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "pt_debug_camera_probe"):
            L.pt_debug_camera_probe.restype = C.c_int
            L.pt_debug_camera_probe.argtypes = [P(abi.Camera), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "pt_debug_scatter_probe"):
            L.pt_debug_scatter_probe.restype = C.c_int
            L.pt_debug_scatter_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        if hasattr(L, "pt_debug_plan"):
            L.pt_debug_plan.restype = C.c_int
            L.pt_debug_plan.argtypes = [P(abi.SceneDesc), P(abi.Camera), P(abi.Params), C.c_int, C.c_int, P(abi.PlanInfo)]
            L.pt_debug_plan_scene.restype = C.c_int
            L.pt_debug_plan_scene.argtypes = [C.c_void_p, P(abi.Camera), P(abi.Params), P(abi.PlanInfo)]
            L.pt_debug_set_tuning.restype = C.c_int
            L.pt_debug_set_tuning.argtypes = [C.c_char_p, C.c_longlong]
            L.pt_debug_get_tuning.restype = C.c_int
            L.pt_debug_get_tuning.argtypes = [C.c_char_p, P(C.c_longlong)]
        if hasattr(L, "pt_debug_handed_over"):
            L.pt_debug_handed_over.restype = C.c_int
            L.pt_debug_handed_over.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "pt_debug_read_queue"):
            L.pt_debug_read_queue.restype = C.c_int
            L.pt_debug_read_queue.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().pt_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        raise PtraceError(rc, last_error())
