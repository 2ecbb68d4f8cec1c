#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray-trace/shade path.

Workload (BASELINE.json configs[1], "C2"): 1280x720, 32 spheres + 1 checkered plane, FlatRenderer,
pixel-centre rays (S=0), synthetic scene of SURVEY.md §8(d).  One *step* = one frame through the C-ABI
(`pt_render_device`: hoist prep + render kernel) with the scene already resident in HBM and the output
left in HBM (fp32 RGB, the reference's PFM precision: 12 B/pixel).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling — the frame grows with
N (same scene and view, area x N), rows are cut in interleaved 8-row blocks and every rank renders its
blocks into its own HBM.  The path has no exchange step, so the timed region has no collective: as at
N = 1 the output stays resident in HBM (there on one GPU, here sharded over N).  Assembling the HdrImage
on rank 0 (one RCCL gather per frame on a side stream, double-buffered behind the next frame's render)
is timed in a second loop and reported as `with_gather`, together with a bit-for-bit check of the
assembled frame against a single-rank render.

Prints ONE JSON line (rank 0).  `value` = rays handed to a world query by all ranks / wall time.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402
from pytracer_amd.dist import ShardedFrameLoop  # noqa: E402

PEAK_FP64_VECTOR_TFLOPS = 78.6  # MI355X vector fp64 (vendor spec; an FMA counts 2), SURVEY.md §8(d)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOP_PER_SPHERE_TEST = 54       # 33 transform + 5 a + 6 b + 6 c + 4 delta
FLOP_PER_PLANE_TEST = 36


def frame_size(n_gpus: int):
    """Weak scaling: area x N at the 16:9 view of the 1280x720 base frame."""
    s = math.sqrt(n_gpus)
    w = int(round(1280 * s / 2)) * 2
    h = int(round(720 * s / 2)) * 2
    return w, h


def cpu_baseline(scene, cam_for, seconds_budget=20.0):
    """Time the CPU oracle (a C restatement of the reference path: kind "port") on this host's cores,
    on the same C2 frame.  Test infrastructure used only as a reported baseline."""
    from oracle import oracle as orc

    orc.build()
    par = abi.make_params(1280, 720, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
    cam = cam_for(1280, 720)
    # the GPU box gives this job a 16-core share of the host (more threads only oversubscribe it)
    threads = int(os.environ.get("PT_CPU_THREADS", min(orc.max_threads(), 16)))
    t0 = time.perf_counter()
    _, rays = orc.render(scene, cam, par, n_threads=threads, sqr_mode=orc.SQR_MUL)
    first = time.perf_counter() - t0
    reps = max(1, min(20, int(seconds_budget / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.render(scene, cam, par, n_threads=threads, sqr_mode=orc.SQR_MUL)
    dt = (time.perf_counter() - t0) / reps
    # one core, bounded: a 1/8 crop of the rows of the same frame
    par1 = abi.copy_params(par, n_ranks=8, rank=3, row_block=8)
    t0 = time.perf_counter()
    _, rays1 = orc.render(scene, cam, par1, n_threads=1, sqr_mode=orc.SQR_MUL)
    dt1 = time.perf_counter() - t0
    orc.set_sqr_mode(orc.SQR_POW)
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
        "sample": f"full 1280x720 C2 frame x{reps} on {threads} threads (OpenMP rows), C oracle, x*x arithmetic",
        "ms_per_frame": dt * 1e3,
        "one_core_Mray_s": rays1 / dt1 / 1e6,
        "one_core_sample": "rows of rank 3/8 (90 rows) of the same frame, 1 thread",
    }


def extra_rows(device: int):
    """Secondary rows (not the headline): the other 1280x720 configurations of BASELINE.json on one GPU,
    kernel time from the library's hipEvents, median of a few frames."""
    rows = {}
    cases = {
        "C3_pathtracer_1280x720_32sph_D3_spp16_N1": (32, False, False, 1280, 720, 7, dict(
            renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3,
            path_state=45, path_seq=54)),
        "C3_cli_default_N10_spp1": (32, False, False, 1280, 720, 3, dict(
            renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10, max_depth=3, rr_limit=3,
            path_state=45, path_seq=54)),
        "C5_flat_1280x720_10k_spheres": (10000, False, True, 1280, 720, 5, dict(renderer=abi.RENDERER_FLAT)),
        "C4_crop_pathtracer_960x540_256sph_D5_spp16": (256, False, True, 960, 540, 3, dict(
            renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=5, rr_limit=3,
            path_state=45, path_seq=54)),
        "C4_pathtracer_3840x2160_256sph_D5_spp64_one_gpu": (256, False, True, 3840, 2160, 3, dict(
            renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1, max_depth=5, rr_limit=3,
            path_state=45, path_seq=54)),
        "C4_share_of_rank_3_of_8": (256, False, True, 3840, 2160, 3, dict(
            renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1, max_depth=5, rr_limit=3,
            path_state=45, path_seq=54, n_ranks=8, rank=3, row_block=8)),
    }
    for name, (ns, plane, wide, W, H, reps, kw) in cases.items():
        flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
        cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
        par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
        ds = DeviceScene(flat, device=device)
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")  # (a rank's share uses the top of it)
        ms = []
        for r in range(reps + 1):
            ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
            st = ds.stats()
            if r > 0:
                ms.append(st.kernel_ms)
        t = float(np.median(ms)) * 1e-3
        n_sph = int((flat.kind == abi.SHAPE_SPHERE).sum())
        n_pl = flat.n_shapes - n_sph
        rows[name] = {"Mray_s": st.n_rays / t / 1e6, "ms_per_frame": t * 1e3, "rays_per_frame": int(st.n_rays),
                      "ray_shape_tests_per_s": st.n_rays * flat.n_shapes / t,
                      "algorithmic_TFLOP_s": st.n_rays * (n_sph * FLOP_PER_SPHERE_TEST + n_pl * FLOP_PER_PLANE_TEST) / t / 1e12}
        ds.close()
    return rows


def boundary_rows(flat, cam_for, device: int):
    """the same C2 frame seen from the drop-in boundary -- ms per frame at the C-ABI with
    caller-owned HOST buffers (`pt_render`: kernel + D2H), scene upload (flatten + H2D), and the Python
    fill of a reference-style HdrImage (a list of W*H Color objects).  None of these is `value`."""
    from pytracer_amd.tracer import _fill_image

    W, H = 1280, 720
    cam = cam_for(W, H)
    t0 = time.perf_counter()
    ds = DeviceScene(flat, device=device)
    upload_ms = (time.perf_counter() - t0) * 1e3
    rows = {"scene_upload_ms": upload_ms}
    for name, fmt in (("pt_render_host_f64_ms", abi.OUT_F64), ("pt_render_host_f32_ms", abi.OUT_F32)):
        par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=fmt)
        ds.render(cam, par)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = ds.render(cam, par)
            ts.append((time.perf_counter() - t0) * 1e3)
        rows[name] = float(np.median(ts))

    class RefColor:  # the reference's Color: three attributes
        __slots__ = ("r", "g", "b")

        def __init__(self, r=0.0, g=0.0, b=0.0):
            self.r, self.g, self.b = r, g, b

    class RefImage:
        def __init__(self, w, h):
            self.width, self.height = w, h
            self.pixels = [RefColor() for _ in range(w * h)]

    img = RefImage(W, H)
    t0 = time.perf_counter()
    _fill_image(img, out.astype(np.float64))
    rows["python_hdrimage_fill_ms"] = (time.perf_counter() - t0) * 1e3
    ds.close()
    return rows


def measured_traffic():
    """HBM bytes per launch of the headline kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
    collected separately, see profiles/): bench.py cannot read hardware counters itself."""
    path = os.path.join(ROOT, "profiles", "traffic_c2.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world_size != n and world_size > 1:
        n = world_size
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # (PT_DIST_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 path on a one-GPU box)
    backend = os.environ.get("PT_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)

    W, H = frame_size(n)
    world = scenes.synthetic_world(32, with_plane=True)
    flat = flatten.flatten_world(world)
    cam_for = lambda w, h: flatten.flatten_camera(scenes.synthetic_camera(w, h))  # noqa: E731
    cam = cam_for(W, H)
    ds = DeviceScene(flat, device=local_rank)
    par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
    loop = ShardedFrameLoop(ds, cam, par, row_block=8)
    rows = loop.rows
    stream = loop.stream
    ev_pairs = []

    TIME_EVERY = 8  # bracket every 8th launch of the timed region with a hipEvent pair

    def step(i, timed, gather=False):
        if timed:
            ds.set_timing(i % TIME_EVERY == 0)
        loop.step(i, gather=gather)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ds.set_count_rays(True)
    for i in range(args.warmup):
        step(i, False)
    fence()
    ds.sync()
    # rays per frame are counted (in-kernel counter) during warm-up; the workload is deterministic, so
    # the timed steps run without the counter: one step == exactly one render-kernel launch
    rays_per_step_local = int(ds.stats().n_rays) if args.warmup > 0 else rows * W
    ds.set_count_rays(False)
    ds.set_timing(False)
    loop.step(0, gather=False)  # one uncounted frame so the timed region starts from the steady state
    fence()
    # every TIME_EVERY-th timed launch is bracketed by its own hipEvent pair on the launch stream (inside
    # the library, directly around the render kernel): their mean is the kernel's average launch duration;
    # the other launches carry no event so that frames run back to back
    ds.profile_begin(args.steps // TIME_EVERY + 2)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_total_ms, kernel_launches = ds.profile_end()
    ds.set_timing(True)

    # the headline numbers first: max time over ranks, rays of all ranks
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_per_step_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_per_step = int(r.item())
    else:
        rays_per_step = rays_per_step_local

    # N > 1: the same loop with the image assembled on rank 0 every frame (RCCL gather, overlapped), and the
    # frame assembled on rank 0 must be bit-identical to the same frame rendered by one rank.  A failure
    # here is reported in the line, it does not take the headline measurement with it.
    gather_elapsed = None
    gather_check = None
    gather_error = None
    gather_steps = max(2, min(args.steps, 100))
    if dist is not None:
        try:
            for i in range(4):
                step(i, False, gather=True)
            fence()
            t0 = time.perf_counter()
            for i in range(gather_steps):
                step(i, False, gather=True)
            fence()
            gather_elapsed = time.perf_counter() - t0
            if rank == 0:
                full = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
                ds.render_into(cam, abi.copy_params(par, n_ranks=1, rank=0), full.data_ptr(), full.numel() * 4, None)
                gather_check = "ok" if torch.equal(full, loop.image()) else "MISMATCH"
            t = torch.tensor([gather_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_elapsed = float(t.item())
        except Exception as e:  # noqa: BLE001  (RCCL / driver errors surface as RuntimeError subclasses)
            gather_error = f"{type(e).__name__}: {e}"[:300]

    avg_kernel_s = kernel_total_ms / max(kernel_launches, 1) * 1e-3

    if rank == 0:
        n_sph = int((flat.kind == abi.SHAPE_SPHERE).sum())
        n_pl = int((flat.kind == abi.SHAPE_PLANE).sum())
        rays_local = rows * W
        flops = rays_local * (n_sph * FLOP_PER_SPHERE_TEST + n_pl * FLOP_PER_PLANE_TEST)
        n_wg = ds.stats().grid
        alg_bytes = rays_local * 12 + n_wg * flat.n_shapes * 104
        tflops = flops / avg_kernel_s / 1e12
        result = {
            "metric": "Mray/s (primary+shadow) at 1280x720 per GPU, C2: 32 spheres + 1 plane, FlatRenderer",
            "value": rays_per_step * args.steps / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C2 flat {W}x{H}, 32 spheres + 1 plane, S=0, fp32 RGB output",
                       "width": W, "height": H, "n_shapes": flat.n_shapes, "renderer": "FlatRenderer",
                       "partition": f"interleaved 8-row blocks over {n} rank(s), output resident in each rank's HBM"},
            "ray_shape_tests_per_s": rays_per_step * flat.n_shapes * args.steps / elapsed,
            "roofline": {
                "bound": "valu_fp64",
                "achieved": tflops, "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / PEAK_FP64_VECTOR_TFLOPS,
                "traffic": None,
                "kernel": "pt_tile_kernel<FLAT> (8x8 tiles, culled shape lists, hoisted scale+translate tests)",
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_flop_per_launch": flops,
                "note": "no dense contraction: MFMA unused; fp64 VALU issue/latency binds (SURVEY.md 8d). Rays of tiles whose only "
                        "possible hit is a dome around the camera are resolved without being traced (DESIGN.md 4, item 8) and are "
                        "counted like the others: they are part of the frame's workload. `achieved` is "
                        "ALGORITHMIC flop (54 per ray-sphere, 36 per ray-plane test, every ray x every shape) / kernel "
                        "time; the kernel executes fewer (tile culling, hoisted origin, scale+translate fast path) with "
                        "bit-identical results, so frac can exceed what brute force allows (peak/2 without FMA)",
                "hbm": {"achieved": alg_bytes / avg_kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": alg_bytes / avg_kernel_s / 1e9 / PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes},
            },
        }
        if gather_error is not None:
            result["with_gather"] = {"error": gather_error}
        elif gather_check is not None:
            result["gather_check"] = gather_check
            result["with_gather"] = {
                "value": rays_per_step * gather_steps / gather_elapsed / 1e6, "unit": "Mray/s",
                "ms_per_step": gather_elapsed / gather_steps * 1e3, "steps": gather_steps,
                "note": "same frames with the HdrImage assembled on rank 0 every frame (RCCL gather of "
                        f"{loop.nbytes / 1e6:.1f} MB per rank, overlapped with the next render); a Flat frame renders faster "
                        "than one xGMI link moves its shard, so this rate is link-bound (DESIGN.md 5)"}
        tr = measured_traffic()
        if tr is not None and n == 1:
            result["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = tr["source"]
        if n == 1 and not args.no_extras:
            result["extra"] = extra_rows(local_rank)
            result["boundary"] = boundary_rows(flat, cam_for, local_rank)
        if n == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(flat, cam_for)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ds.close()


if __name__ == "__main__":
    main()
