"""SURVEY.md §8f next-3 — HdrImage post-processing (write_pfm, average_luminosity, normalize, clamp, LDR).

CPU: the oracle against the reference's golden outputs (bit-exact) and its known-answer vectors.
GPU: the device kernels (through the C-ABI) against the same goldens."""
from io import BytesIO

import numpy as np
import pytest

from tests import util

# test_all.py:112-143: the reference's own PFM vectors for the 3x2 image below
LE_REFERENCE_BYTES = bytes([
    0x50, 0x46, 0x0A, 0x33, 0x20, 0x32, 0x0A, 0x2D, 0x31, 0x2E, 0x30, 0x0A, 0x00, 0x00, 0xC8, 0x42,
    0x00, 0x00, 0x48, 0x43, 0x00, 0x00, 0x96, 0x43, 0x00, 0x00, 0xC8, 0x43, 0x00, 0x00, 0xFA, 0x43,
    0x00, 0x00, 0x16, 0x44, 0x00, 0x00, 0x2F, 0x44, 0x00, 0x00, 0x48, 0x44, 0x00, 0x00, 0x61, 0x44,
    0x00, 0x00, 0x20, 0x41, 0x00, 0x00, 0xA0, 0x41, 0x00, 0x00, 0xF0, 0x41, 0x00, 0x00, 0x20, 0x42,
    0x00, 0x00, 0x48, 0x42, 0x00, 0x00, 0x70, 0x42, 0x00, 0x00, 0x8C, 0x42, 0x00, 0x00, 0xA0, 0x42,
    0x00, 0x00, 0xB4, 0x42])
BE_REFERENCE_BYTES = bytes([
    0x50, 0x46, 0x0A, 0x33, 0x20, 0x32, 0x0A, 0x31, 0x2E, 0x30, 0x0A, 0x42, 0xC8, 0x00, 0x00, 0x43,
    0x48, 0x00, 0x00, 0x43, 0x96, 0x00, 0x00, 0x43, 0xC8, 0x00, 0x00, 0x43, 0xFA, 0x00, 0x00, 0x44,
    0x16, 0x00, 0x00, 0x44, 0x2F, 0x00, 0x00, 0x44, 0x48, 0x00, 0x00, 0x44, 0x61, 0x00, 0x00, 0x41,
    0x20, 0x00, 0x00, 0x41, 0xA0, 0x00, 0x00, 0x41, 0xF0, 0x00, 0x00, 0x42, 0x20, 0x00, 0x00, 0x42,
    0x48, 0x00, 0x00, 0x42, 0x70, 0x00, 0x00, 0x42, 0x8C, 0x00, 0x00, 0x42, 0xA0, 0x00, 0x00, 0x42,
    0xB4, 0x00, 0x00])
# test_all.py:177-186: the image those bytes encode
REF_IMAGE = np.array([[[1.0e1, 2.0e1, 3.0e1], [4.0e1, 5.0e1, 6.0e1], [7.0e1, 8.0e1, 9.0e1]],
                      [[1.0e2, 2.0e2, 3.0e2], [4.0e2, 5.0e2, 6.0e2], [7.0e2, 8.0e2, 9.0e2]]])
LE_HEADER, BE_HEADER = b"PF\n3 2\n-1.0\n", b"PF\n3 2\n1.0\n"


def test_oracle_pfm_known_answer(oracle):
    assert LE_HEADER + oracle.pack_pfm(REF_IMAGE, False) == LE_REFERENCE_BYTES
    assert BE_HEADER + oracle.pack_pfm(REF_IMAGE, True) == BE_REFERENCE_BYTES


def test_oracle_luminosity_known_answers(oracle):
    img = np.array([[[0.5e1, 1.0e1, 1.5e1], [0.5e3, 1.0e3, 1.5e3]]])  # test_all.py:239-246
    assert oracle.average_luminosity(img, delta=0.0) == pytest.approx(100.0)
    toned, _ = oracle.tonemap(img, 1000.0 / 100.0, clamp=False)  # test_all.py:248-256
    assert np.allclose(toned, [[[0.5e2, 1.0e2, 1.5e2], [0.5e4, 1.0e4, 1.5e4]]])
    clamped, _ = oracle.tonemap(img, 1.0, clamp=True)
    assert np.all((clamped >= 0) & (clamped <= 1))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_postprocess_golden(oracle, tag):
    g = util.load("g10_postprocess")
    px = g[f"{tag}_pixels"]
    h, w = px.shape[:2]
    hdr = f"PF\n{w} {h}\n".encode()
    assert hdr + b"-1.0\n" + oracle.pack_pfm(px, False) == g[f"{tag}_pfm_le"].tobytes()
    assert hdr + b"1.0\n" + oracle.pack_pfm(px, True) == g[f"{tag}_pfm_be"].tobytes()
    lum = oracle.average_luminosity(px)
    assert lum == float(g[f"{tag}_lum"])
    assert oracle.average_luminosity(px, 1e-3) == float(g[f"{tag}_lum_delta0"])
    toned, ldr = oracle.tonemap(px, 1.0 / lum, clamp=True, gamma=1.0)
    assert util.bits_equal(toned, g[f"{tag}_toned"])
    assert np.array_equal(ldr, g[f"{tag}_ldr_g10"])
    _, ldr22 = oracle.tonemap(g[f"{tag}_toned"], 1.0, clamp=False, gamma=2.2)
    assert np.array_equal(ldr22, g[f"{tag}_ldr_g22"])


# ---- device ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_pfm_known_answer():
    from pytracer_amd.postprocess import BIG_ENDIAN, LITTLE_ENDIAN, DeviceImage

    for dtype in (np.float64, np.float32):
        img = DeviceImage.from_numpy(REF_IMAGE.astype(dtype))
        buf = BytesIO()
        img.write_pfm(buf, LITTLE_ENDIAN)
        assert buf.getvalue() == LE_REFERENCE_BYTES
        buf = BytesIO()
        img.write_pfm(buf, BIG_ENDIAN)
        assert buf.getvalue() == BE_REFERENCE_BYTES


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_device_postprocess_golden(tag):
    from pytracer_amd.postprocess import BIG_ENDIAN, DeviceImage

    g = util.load("g10_postprocess")
    px = g[f"{tag}_pixels"]
    img = DeviceImage.from_numpy(px)
    buf = BytesIO()
    img.write_pfm(buf)
    assert buf.getvalue() == g[f"{tag}_pfm_le"].tobytes()  # byte-exact
    buf = BytesIO()
    img.write_pfm(buf, BIG_ENDIAN)
    assert buf.getvalue() == g[f"{tag}_pfm_be"].tobytes()
    # the reference sums log10 sequentially; the device sums pairwise: equal to rounding of the sum
    lum = img.average_luminosity()
    assert lum == pytest.approx(float(g[f"{tag}_lum"]), rel=1e-12)
    img.normalize_image(factor=1.0)
    img.clamp_image()
    assert util.rel_err(img.numpy(), g[f"{tag}_toned"]).max() <= 1e-12
    for gamma, key in ((1.0, "g10"), (2.2, "g22")):
        ldr = img.ldr_bytes(gamma).astype(np.int32)
        diff = np.abs(ldr - g[f"{tag}_ldr_{key}"])
        # int() truncation is discontinuous: allow a unit step on a handful of values (pow is not bit-equal)
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


@pytest.mark.gpu
def test_device_resident_tensor_path():
    """A frame left in HBM by pt_render_device (a torch tensor) goes through the same kernels in place."""
    import torch

    from pytracer_amd.postprocess import DeviceImage

    g = util.load("g10_postprocess")
    t = torch.from_numpy(g["a_pixels"]).cuda().contiguous()
    img = DeviceImage(t)
    buf = BytesIO()
    img.write_pfm(buf)
    assert buf.getvalue() == g["a_pfm_le"].tobytes()
    img.normalize_image(factor=1.0)
    img.clamp_image()
    assert util.rel_err(t.cpu().numpy(), g["a_toned"]).max() <= 1e-12  # modified in HBM, in place


@pytest.mark.gpu
def test_hdrimage_standin_methods():
    from pytracer_amd import hostmodel as hm

    g = util.load("g10_postprocess")
    px = g["a_pixels"]
    img = hm.HdrImage(px.shape[1], px.shape[0])
    img.set_array(px)
    buf = BytesIO()
    img.write_pfm(buf)
    assert buf.getvalue() == g["a_pfm_le"].tobytes()
    assert img.average_luminosity() == pytest.approx(float(g["a_lum"]), rel=1e-12)
    img.normalize_image(factor=1.0)
    img.clamp_image()
    assert util.rel_err(img.array, g["a_toned"]).max() <= 1e-12
    png = BytesIO()
    img.write_ldr_image(png, "PNG")
    assert png.getvalue()[:8] == b"\x89PNG\r\n\x1a\n"
