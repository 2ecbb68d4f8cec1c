"""The sparse form of a shard on the device (csrc/pt_post.h: pt_image_sparse_encode / _decode) against the torch
restatement of the same format in pytracer_amd/dist.py (which the gloo tests run on the CPU): same bytes, and back."""
import numpy as np
import pytest
import torch

from pytracer_amd import abi, dist as ptdist

pytestmark = pytest.mark.gpu


def _shard(rows, W, dtype, kind, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.empty((rows, W, 3), dtype=dtype)
    x[...] = torch.tensor([0.25, -0.0, float("nan")], dtype=dtype)  # one colour, with a sign bit and a NaN in it
    if kind == "patches":
        for _ in range(1 + rows * W // 2000):
            r0, c0 = int(torch.randint(0, rows, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
            x[r0:r0 + 2, c0:c0 + 50] = torch.rand((min(2, rows - r0), min(50, W - c0), 3), generator=g, dtype=dtype)
        x[rows // 2, W // 2, 1] = 0.0  # (0.0 where the sky has -0.0)
    elif kind == "noise":
        x = torch.rand((rows, W, 3), generator=g, dtype=dtype)
    return x


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("rows,W,kind", [
    (1, 1, "sky"), (1, 127, "patches"), (1, 128, "sky"), (3, 129, "patches"), (270, 3840, "patches"), (64, 1000, "noise"),
    (17, 333, "sky"), (2, 100000, "patches"),
])
def test_device_codec_is_the_torch_codec(dtype, rows, W, kind):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    bits = torch.int32 if dtype == torch.float32 else torch.int64
    x = _shard(rows, W, dtype, kind, rows * 7 + W)
    f_cpu, p_cpu = ptdist.encode_sparse(x)
    xd = x.cuda()
    f_dev, p_dev = ptdist.encode_sparse(xd)
    torch.cuda.synchronize()
    assert f_dev.is_cuda and torch.equal(f_dev.cpu(), f_cpu)
    assert p_dev.shape == p_cpu.shape and torch.equal(p_dev.cpu().view(bits), p_cpu.view(bits))
    back = ptdist.decode_sparse(f_dev, p_dev, rows * W, dtype)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu().view(bits), x.reshape(-1, 3).contiguous().view(bits))
    # ... and a shard encoded on the CPU decodes on the device
    back2 = ptdist.decode_sparse(f_cpu.cuda(), p_cpu.cuda(), rows * W, dtype)
    assert torch.equal(back2.cpu().view(bits), x.reshape(-1, 3).contiguous().view(bits))
    # ... straight into a frame: rank 1 of 3, blocks of 2 rows (the same placement as dist.place_shard on the CPU)
    world, rb, r = 3, 2, 1
    H = 0
    while len(ptdist.shard_rows(H, rb, world, r)) < rows:
        H += 1
    if len(ptdist.shard_rows(H, rb, world, r)) == rows:
        frame_dev = torch.full((H, W, 3), 7.0, dtype=dtype, device="cuda")
        ptdist.decode_sparse(f_dev, p_dev, rows * W, dtype, frame=frame_dev, row_block=rb, world=world, rank=r)
        frame_cpu = torch.full((H, W, 3), 7.0, dtype=dtype)
        ptdist.decode_sparse(f_cpu, p_cpu, rows * W, dtype, frame=frame_cpu, row_block=rb, world=world, rank=r)
        assert torch.equal(frame_dev.cpu().view(bits), frame_cpu.view(bits))
        assert torch.equal(frame_cpu[ptdist.shard_rows(H, rb, world, r)].view(bits), x.view(bits))


def test_a_rendered_shard_goes_through():
    """Rank 3 of 8's rows of a 4K path-traced frame: mostly sky; the sparse form is a fraction of the shard and lossless."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from pytracer_amd import device, flatten, scenes

    W, H = 1920, 1080
    flat = flatten.flatten_world(scenes.synthetic_world(256, wide=True))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, rr_limit=3, path_state=45,
                          path_seq=54, pcg_mode=abi.PCG_SAMPLE, n_ranks=8, rank=3, row_block=8, out_format=abi.OUT_F32)
    rows = len(ptdist.shard_rows(H, 8, 8, 3))
    shard = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    with device.DeviceScene(flat) as ds:
        ds.render_into(cam, par, shard.data_ptr(), shard.numel() * 4, None)
    fixed, payload = ptdist.encode_sparse(shard)
    back = ptdist.decode_sparse(fixed, payload, rows * W, torch.float32).view(rows, W, 3)
    assert torch.equal(back.view(torch.int32), shard.view(torch.int32))
    sent = fixed.numel() + payload.numel() * 4
    assert sent < shard.numel() * 4 // 4
    assert np.isfinite(back.cpu().numpy()).all()


@pytest.mark.parametrize("world,H,W,rb,dtype", [(8, 2160, 3840, 8, torch.float32), (3, 50, 300, 8, torch.float64), (5, 27, 129, 4, torch.float32)])
def test_all_remote_shards_in_one_launch(world, H, W, rb, dtype):
    """What rank 0 does per frame: the sparse shards of ranks 1 .. world-1 decoded straight into the frame by ONE launch."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    bits = torch.int32 if dtype == torch.float32 else torch.int64
    full = _shard(H, W, dtype, "patches", 11).cuda()
    frame = torch.zeros_like(full)
    ranks = [r for r in range(1, world) if ptdist.shard_rows(H, rb, world, r)]
    fixed, payload = [], []
    for r in ranks:
        f, p = ptdist.encode_sparse(full[ptdist.shard_rows(H, rb, world, r)].contiguous())
        fixed.append(f)
        payload.append(p.clone())  # (encode_sparse reuses its payload buffer from call to call)
    ptdist.decode_sparse_many(fixed, payload, ranks, frame, rb, world)
    ptdist.place_shard(frame, full[ptdist.shard_rows(H, rb, world, 0)].contiguous(), H, rb, world, 0)
    torch.cuda.synchronize()
    assert torch.equal(frame.view(bits), full.view(bits))
