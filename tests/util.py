"""Helpers shared by the tests: golden-fixture loading and tolerance checks."""
import os

import numpy as np

from pytracer_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def params_from(d) -> abi.Params:
    p = abi.Params()
    for fname, _ in abi.Params._fields_:
        v = d["par_" + fname]
        if v.ndim == 0:
            setattr(p, fname, int(v) if np.issubdtype(v.dtype, np.integer) else float(v))
        else:
            arr = getattr(p, fname)
            for i in range(len(v)):
                arr[i] = float(v[i])
    return p


def load_frame(name):
    """-> (FlatScene, Camera, Params, pixels[H, W, 3] computed by the reference)"""
    d = load(name)
    return abi.FlatScene.from_dict(d), abi.camera_from_dict(d), params_from(d), d["pixels"]


FRAME_FIXTURES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("g5_") and f.endswith(".npz"))


def rel_err(a, b):
    """|a-b| / max(|a|,|b|) per channel, exact zero matching zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.where(den > 0, np.abs(a - b) / den, 0.0)
    return e


def bits_equal(a, b) -> bool:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def bits_equal_rows(a, b) -> np.ndarray:
    """Per row: are all values of the row bit-identical?"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    return (a2.view(np.uint64) == b2.view(np.uint64)).all(axis=1)
