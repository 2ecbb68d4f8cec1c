"""Several frames in flight on one GPU.

The reference renders an animation as one process per frame (``main.py:76-214`` called once per camera angle); here a
frame is one kernel launch of ~10 us whose waves are bound by dependent fp64 latency, and a launch has a fixed cost of
~2-3 us (dispatch, ramp, drain: ``tools/micro/ramp.hip``) that nothing inside the frame can hide.  Consecutive frames
are independent, so the next frame can: ``FramePipeline`` keeps ``n_in_flight`` handles on ONE uploaded scene
(``pt_scene_clone``: each with its own per-camera tables, queues and counters -- one handle must not render two frames at
once) and as many HIP streams, and frame ``i`` goes to slot ``i % n_in_flight``.  The frames are bit-identical to the ones a single stream renders.
"""
from typing import List, Optional

from . import abi
from .device import DeviceScene
from .devmem import Stream


class FramePipeline:
    def __init__(self, flat: abi.FlatScene, n_in_flight: int = 2, device: int = 0):
        if n_in_flight < 1:
            raise ValueError("n_in_flight must be >= 1")
        self.device = int(device)
        first = DeviceScene(flat, device=device)  # uploaded once; the other slots are further handles on the same tables
        self.scenes: List[DeviceScene] = [first] + [first.clone() for _ in range(n_in_flight - 1)]
        self.streams = [Stream(self.device) for _ in range(n_in_flight)]  # (pt_stream_create: no GPU framework needed)
        self._busy: List[Optional[object]] = [None] * n_in_flight  # keeps a slot's output alive while it renders
        self._next = 0

    @property
    def n_in_flight(self) -> int:
        return len(self.scenes)

    def submit(self, cam: abi.Camera, params: abi.Params, out) -> int:
        """Enqueue one frame into ``out`` -- device memory of ``pt_output_bytes`` bytes: a
        :class:`pytracer_amd.devmem.DeviceBuffer` or a contiguous CUDA torch tensor -- on the next slot's stream; returns
        the slot.  Work already queued on that slot runs first (stream order); nothing here waits on the host."""
        if getattr(out, "is_device_buffer", False):
            nbytes = out.nbytes
        elif getattr(out, "is_cuda", False) and out.is_contiguous():
            nbytes = out.numel() * out.element_size()
        else:
            raise ValueError("out must be a DeviceBuffer or a contiguous CUDA tensor")
        slot = self._next
        self._next = (slot + 1) % len(self.scenes)
        self.scenes[slot].render_into(cam, params, out.data_ptr(), nbytes, self.streams[slot].handle)
        self._busy[slot] = out
        return slot

    def wait(self, slot: Optional[int] = None) -> None:
        """Block until the frames of ``slot`` (default: of every slot) are done."""
        for s in range(len(self.scenes)) if slot is None else (slot,):
            self.streams[s].synchronize()
            self._busy[s] = None

    def set_count_rays(self, enable: bool) -> None:
        for ds in self.scenes:
            ds.set_count_rays(enable)

    def set_timing(self, enable: bool) -> None:
        for ds in self.scenes:
            ds.set_timing(enable)

    def set_dome_shortcut(self, enable: bool) -> None:
        for ds in self.scenes:
            ds.set_dome_shortcut(enable)

    def close(self) -> None:
        self.wait()
        for ds in self.scenes:
            ds.close()
        self.scenes = []
        for st in self.streams:
            st.close()
        self.streams = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
