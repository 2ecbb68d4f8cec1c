#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X ray-trace/shade path (BASELINE.json metric: Mray/s, ms/frame).

    python bench.py [--gpus N] [--steps K] [--warmup W]

The LAST line on stdout is ONE compact JSON object (a fixed set of keys, < 8 KB: tests/test_bench_helpers.py); everything else
the run measured goes to bench_detail.json next to this script and, as one line, to stderr.

N = 1 -- workload = BASELINE.json configs[1] ("C2"): 1280x720, 32 spheres + 1 checkered plane, FlatRenderer,
pixel-centre rays (S=0), synthetic scene of SURVEY.md 8(d).  One *step* = one frame through the C-ABI
(`pt_render_device`) with the scene resident in HBM and the output left in HBM (fp32 RGB, the reference's PFM
precision: 12 B/pixel).  The headline loop runs with the dome shortcut OFF: every primary ray is generated and handed to a
world query, so `value` counts only queried rays; the library's default frame (shortcut on) is the side row
`frame_with_dome_shortcut`, and `ms_per_frame_at_c_abi` is kernel + D2H through `pt_render`.

N > 1 (one rank per GPU, RCCL; under `torch.distributed.run`, or by itself: `python bench.py --gpus N` starts its
own N ranks before touching the GPU and relays rank 0's line) -- workload = configs[3] ("C4"): ONE 3840x2160 frame,
256 spheres, PathTracer depth 5, 64 samples per pixel, STRONG-scaled: rows are cut in interleaved 8-row blocks,
every rank renders its blocks, and the frame is assembled on rank 0 by one batched RCCL point-to-point group per
frame, INSIDE the timed region: `value` = rays that went through a world query x K / wall time.  The whole-shard gather (one
transfer per remote rank) is measured first and completely -- loop, check against rank 0 alone, the same loop on one GPU, the
oracle check -- then the sparse gather; the headline is the faster, and the whole-shard row if the sparse one fails or overruns
its deadline (`gather.fallback_reason`).  Every phase runs under a deadline (PT_BENCH_PHASE_S); a phase that overruns ends the
job with the line measured so far (or an `error` line when nothing is complete yet).  The estimated wall time is printed to
stderr up front.
"""
import argparse
import datetime
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory (profiles/r04_dev_kernarg.txt): this process is the benchmark's own, so it asks for them --
# before its first HIP call, i.e. before torch is imported; the line reports what the run had (`hip_force_dev_kernarg`)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene, device_info  # noqa: E402
from pytracer_amd.dist import ShardedFrameLoop  # noqa: E402

PEAK_FP64_VECTOR_TFLOPS = 78.6  # MI355X vector fp64, vendor figure: an FMA counts 2
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOP_PER_SPHERE_TEST = 54       # 33 transform + 5 a + 6 b + 6 c + 4 delta
FLOP_PER_PLANE_TEST = 36
# SIMD-cycles one wave64 instruction costs its SIMD, MEASURED on this part with every CU busy and 2-8 waves per SIMD
# (tools/micro/issue.hip -> profiles/r03_issue_rates.{txt,json}; the loop overhead of the benchmark, ~5 %, removed):
# fp64 add/mul/fma 4; fp32 and 32-bit integer add/mul/fma/logic/move 2 (4 for ONE wave alone on a SIMD); packed fp32,
# 64-bit integer, 32-bit integer multiply, conversions to/from fp64, compares into an SGPR pair, selects on an SGPR
# mask, lane reads 4; fp32 transcendentals 8; fp64 rcp/rsq/sqrt 16.  SALU: 4 per SIMD, issued beside the VALU.
# rocprofv3 counts VALU instructions by class (SQ_INSTS_VALU_*); what no class counter covers ("other": moves,
# compares, selects, lane reads, bit operations) is priced at 3, between the 2 of a move and the 4 of a compare,
# and the fractions under 2 and under 4 are printed beside it.
VALU_ISSUE_CYCLES = {"SQ_INSTS_VALU_ADD_F64": 4, "SQ_INSTS_VALU_MUL_F64": 4, "SQ_INSTS_VALU_FMA_F64": 4, "SQ_INSTS_VALU_TRANS_F64": 16,
                     "SQ_INSTS_VALU_ADD_F32": 2, "SQ_INSTS_VALU_MUL_F32": 2, "SQ_INSTS_VALU_FMA_F32": 2, "SQ_INSTS_VALU_TRANS_F32": 8,
                     "SQ_INSTS_VALU_CVT": 4, "SQ_INSTS_VALU_INT32": 2, "SQ_INSTS_VALU_INT64": 4}
VALU_OTHER_CYCLES = (2, 3, 4)   # (low, priced, high) for instructions outside the class counters
SALU_ISSUE_CYCLES = 4
# What a scalar instruction costs the VECTOR issue of its SIMD: not nothing.  Measured with the same
# microbenchmark under rocprofv3 (profiles/r03_issue_pmc_probe.txt, 4 waves per SIMD): v_mul_f64 alone 4.88 SIMD-cycles per
# instruction, v_mul_f64 + s_add_u32 1:1 6.29 -> +1.41 per scalar instruction; v_add_u32 2.58 -> 6.15 with s_add_u32 1:1
# (the scalar unit's 4 cycles per SIMD become the bound there); v_add_f32 + s_add_u32 2:1 2.77 per VALU -> +0.24, 4:1 -> +1.56.
# Priced: 1.4 SIMD-cycles of lost vector issue per scalar instruction (the fp64 figure: the kernels here are fp64 chains).
SALU_COISSUE_PENALTY_CYCLES = 1.4
FP64_CLASSES = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")


def priced_issue(counters, simd_cycles):
    """VALU / fp64-pipe / SALU issue utilisation of a launch from its per-class instruction counts.
    -> dict, or None when the class counters are missing."""
    valu = counters.get("SQ_INSTS_VALU")
    if not valu or any(k not in counters for k in VALU_ISSUE_CYCLES):
        return None
    classed = sum(counters[k] for k in VALU_ISSUE_CYCLES)
    other = max(0.0, valu - classed)
    base = sum(counters[k] * c for k, c in VALU_ISSUE_CYCLES.items())
    lo, mid, hi = (base + other * c for c in VALU_OTHER_CYCLES)
    f64 = sum(counters[k] * VALU_ISSUE_CYCLES[k] for k in FP64_CLASSES)
    salu = counters.get("SQ_INSTS_SALU", 0.0)
    row = {"valu_issue_cycles": mid, "frac": mid / simd_cycles, "frac_other_at_2": lo / simd_cycles, "frac_other_at_4": hi / simd_cycles,
           "fp64_pipe_frac": f64 / simd_cycles,
           "salu_issue_frac": salu * SALU_ISSUE_CYCLES / simd_cycles,
           # vector issue INCLUDING what the scalar instructions take from it (they do not issue entirely beside the VALU)
           "frac_with_salu_coissue": (mid + salu * SALU_COISSUE_PENALTY_CYCLES) / simd_cycles,
           "mean_cycles_per_valu_instruction": mid / valu,
           "instructions_by_class": {**{k.replace("SQ_INSTS_VALU_", "").lower(): counters[k] for k in VALU_ISSUE_CYCLES}, "other": other},
           "unclassed_share": other / valu,
           "fp64_instruction_share": sum(counters[k] for k in FP64_CLASSES) / valu}
    # counter-derived cross-checks, no price table involved: SQ_ACTIVE_INST_VALU counts quad-cycles of VALU execution (1 per
    # ordinary instruction whether it issues in 2 or 4 cycles, 2 / 4 for 8- / 16-cycle transcendentals:
    # profiles/r03_issue_pmc_probe.txt), so x 4 / SIMD-cycles is the utilisation with every fp32 / int32 instruction
    # charged 4 cycles -- an UPPER bracket of `frac`; SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of their lifetime the waves wait
    if "SQ_ACTIVE_INST_VALU" in counters:
        row["frac_counter_active_inst_valu_x4"] = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
    if counters.get("SQ_WAVE_CYCLES"):
        for name, key in (("SQ_WAIT_ANY", "wave_cycles_waiting_any_frac"), ("SQ_WAIT_INST_ANY", "wave_cycles_waiting_inst_frac")):
            if name in counters:
                row[key] = counters[name] / counters["SQ_WAVE_CYCLES"]
    return row

C4 = dict(n_spheres=256, wide=True, W=3840, H=2160,
          kw=dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1, max_depth=5, rr_limit=3,
                  path_state=45, path_seq=54))
PCG_NAMES = {abi.PCG_PIXEL: "PT_PCG_PIXEL", abi.PCG_SAMPLE: "PT_PCG_SAMPLE"}


def cam_for(w, h):
    return flatten.flatten_camera(scenes.synthetic_camera(w, h))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def usable_cores():
    """-> (cores this job may really use, where that number came from): the affinity mask, cut by the cgroup's CPU quota
    when there is one ("cgroup quota"); a GPU box hands a one-GPU job a 16-core share of a much larger host, so with
    neither a quota nor a narrowed affinity mask to read, 16 is ASSUMED and the line says so."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    source = "affinity mask" if n < (os.cpu_count() or n) else "all cores of the host"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    return max(1, min(n, int(int(txt[0]) / int(txt[1]) + 0.5))), "cgroup quota"
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    return max(1, min(n, int(q / per + 0.5))), "cgroup quota"
        except (OSError, ValueError, IndexError):
            pass
    if n > 16:
        return 16, "assumed (no cgroup quota readable, affinity mask wider than the box's 16-core share)"
    return n, source


def cpu_baseline(scene, seconds_budget=20.0):
    """Time the CPU oracle (a C restatement of the reference path: kind "port") on this host's cores, on the
    same C2 frame.  Test infrastructure used only as a reported baseline."""
    from oracle import oracle as orc

    orc.build()
    par = abi.make_params(1280, 720, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
    cam = cam_for(1280, 720)
    usable, usable_source = usable_cores()
    threads = int(os.environ.get("PT_CPU_THREADS", max(1, min(orc.max_threads(), usable))))
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
    # the same loop in the interpreter: oracle/pyloop.py, pure Python, one thread, a quarter-size frame
    from oracle import pyloop

    par_py = abi.make_params(320, 180, abi.RENDERER_FLAT)
    t0 = time.perf_counter()
    _, rays_py = pyloop.render(scene, cam_for(320, 180), par_py)
    dt_py = time.perf_counter() - t0
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
        "interpreted": {"value": rays_py / dt_py / 1e6, "unit": "Mray/s", "cores": 1, "kind": "port",
                        "sample": "the C2 scene at 320x180 (57 600 rays), oracle/pyloop.py: the per-pixel loop in pure Python "
                                  "on flattened arrays, bit-identical to the C oracle (tests/test_oracle_golden.py)"},
        "sample": f"full 1280x720 C2 frame x{reps} on {threads} threads (OpenMP rows), C oracle, x*x arithmetic",
        "cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": usable, "usable_cores_source": usable_source,
        "ms_per_frame": dt * 1e3,
        "one_core_Mray_s": rays1 / dt1 / 1e6,
        "one_core_sample": "rows of rank 3/8 (90 rows) of the same frame, 1 thread",
        "reference_itself": "pure Python, one core, measured in the build container: 4.2e3 rays/s on this scene shape (BASELINE.md)",
    }


def parity_check(flat, cam, par, frame):
    """The frame the timed loop left in HBM against the CPU oracle's frame of the same scene, camera and parameters
    (oracle/pt_oracle.c in the device's x*x arithmetic, fp32 output = the rounded fp64 value on both sides): C2 is Flat
    over uniform and checkered-plane pigments, no libm transcendental involved, so the bar is BIT-IDENTICAL.  The oracle
    is the checker here, never the thing measured (20 ms on the box's cores)."""
    from oracle import oracle as orc

    orc.build()
    want, n_rays = orc.render(flat, cam, par, n_threads=max(1, min(orc.max_threads(), usable_cores()[0])), sqr_mode=orc.SQR_MUL)
    orc.set_sqr_mode(orc.SQR_POW)
    got = frame.detach().cpu().numpy()
    same = got.shape == want.shape and got.dtype == want.dtype and got.tobytes() == want.tobytes()
    row = {"bit_identical": bool(same), "pixels": int(want.shape[0] * want.shape[1]), "oracle_rays": int(n_rays),
           "against": "oracle/pt_oracle.c (C restatement of the reference path, pinned to the reference's own outputs by "
                      "tests/golden), x*x arithmetic, same fp32 output format",
           "frame": "the buffer the LAST step of the timed loop rendered into, downloaded after the timed region"}
    if not same and got.shape == want.shape:
        row["pixels_differing"] = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    return row


ORACLE_TESTS_PER_CORE_S = 9.0e7  # ray-shape tests per second and core of oracle/pt_oracle.c (C2: 2.9 Mray/s x 33 shapes on an EPYC 9575F core)


def oracle_check(flat, cam, par, frame, rays):
    """The assembled frame of a sharded run against the CPU oracle's frame of the same scene, camera and parameters (x*x
    arithmetic, fp32 output): bit-identical pixels counted, outliers beyond 1e-5 relative per channel counted.  The oracle is
    the checker, outside every timed region.  Skipped with a reason when it would take longer than PT_BENCH_ORACLE_S (default
    200 s) on this host's usable cores -- the other ranks wait behind a barrier meanwhile."""
    from oracle import oracle as orc

    orc.build()
    # every core this JOB may use: the other ranks wait behind a barrier meanwhile, and a launcher's OMP_NUM_THREADS (torchrun
    # sets 1 per rank) is a default for the ranks' own libraries, not a limit on this check (the oracle sets its thread count)
    cores = max(1, usable_cores()[0])
    est = rays * flat.n_shapes / (ORACLE_TESTS_PER_CORE_S * cores)
    limit = float(os.environ.get("PT_BENCH_ORACLE_S", "200"))
    if est > limit:
        return {"checked": False, "reason": f"the oracle would need about {est:.0f} s on {cores} cores (limit {limit:.0f} s: PT_BENCH_ORACLE_S)"}
    t0 = time.perf_counter()
    want, n_rays = orc.render(flat, cam, abi.copy_params(par, n_ranks=1, rank=0), n_threads=cores, sqr_mode=orc.SQR_MUL)
    orc.set_sqr_mode(orc.SQR_POW)
    dt = time.perf_counter() - t0
    got = frame.detach().cpu().numpy()
    same_px = (got.view(np.uint32) == want.view(np.uint32)).all(axis=-1)
    den = np.maximum(np.abs(got), np.abs(want)).astype(np.float64)
    err = np.where(den > 0, np.abs(got.astype(np.float64) - want) / np.where(den > 0, den, 1.0), 0.0)
    return {"checked": True, "bit_identical": bool(same_px.all()), "pixels": int(same_px.size), "pixels_differing": int((~same_px).sum()),
            "pixels_beyond_1e-5": int((err > 1e-5).any(axis=-1).sum()), "max_rel_err": float(err.max()),
            "rays_match": int(n_rays) == int(rays), "oracle_seconds": dt, "cores": cores,
            "against": "oracle/pt_oracle.c (C restatement of the reference path, pinned to the reference's own outputs by tests/golden), "
                       "x*x arithmetic, fp32 output; computed on rank 0's host cores outside every timed region"}


def kernel_row(ds, cam, par, out, reps, flat):
    """Median kernel time (events in the dispatches) and ray statistics of `reps` frames."""
    ms = []
    for r in range(reps + 1):
        ds.render_into(cam, par, out.data_ptr(), out.numel() * out.element_size(), None)
        st = ds.stats()
        if r > 0:
            ms.append(st.kernel_ms)
    t = float(np.median(ms)) * 1e-3
    n_sph = int((flat.kind == abi.SHAPE_SPHERE).sum())
    n_pl = flat.n_shapes - n_sph
    traced = int(st.n_rays - st.n_rays_resolved)
    # (rates count only rays that went through a world query; the rays the dome shortcut settled are listed beside them)
    return {"traced_Mray_s": traced / t / 1e6, "ms_per_frame": t * 1e3, "rays_traced_per_frame": traced,
            "rays_settled_without_a_query_per_frame": int(st.n_rays_resolved),
            "ray_shape_tests_per_s": traced * flat.n_shapes / t,
            "algorithmic_equivalent_TFLOP_s": traced * (n_sph * FLOP_PER_SPHERE_TEST + n_pl * FLOP_PER_PLANE_TEST) / t / 1e12}


def in_flight_row(ds, cam, par, out, frames):
    """The same frames with TWO in flight: a second handle on the uploaded scene (pt_scene_clone) and a stream each, frames
    dealt alternately; wall clock between two device synchronisations -> ms per frame (a frame RATE: one frame still
    takes what ms_per_frame says).  Checked: both handles' frames are the frame `out` holds."""
    other = ds.clone()
    outs = [out, torch.empty_like(out)]
    handles, streams = [ds, other], [torch.cuda.Stream(), torch.cuda.Stream()]
    nbytes = out.numel() * out.element_size()
    for h in handles:
        h.set_count_rays(False)
        h.set_timing(False)
    try:
        for i in range(4):
            handles[i & 1].render_into(cam, par, outs[i & 1].data_ptr(), nbytes, streams[i & 1].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            handles[i & 1].render_into(cam, par, outs[i & 1].data_ptr(), nbytes, streams[i & 1].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = bool(torch.equal(outs[0], outs[1]))
    finally:
        ds.set_count_rays(True)
        ds.set_timing(True)
        other.close()
    return {"ms_per_frame": dt / frames * 1e3, "frames": frames, "frames_identical": same,
            "note": "a frame rate (wall clock / frames), two frames in flight on two handles of the scene"}


def extra_rows(device: int):
    """Secondary rows (not the headline): the other configurations of BASELINE.json on one GPU, kernel time from
    the library's events, median of a few frames.  The path-traced ones in both per-thread PCG modes."""
    rows = {}
    c3 = dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3,
              path_state=45, path_seq=54)
    cases = {
        "C3_pathtracer_1280x720_32sph_D3_spp16_N1": (32, False, False, 1280, 720, 7, c3),
        "C3_same_PT_PCG_SAMPLE": (32, False, False, 1280, 720, 7, dict(c3, pcg_mode=abi.PCG_SAMPLE)),
        "C3_cli_default_N10_spp1": (32, False, False, 1280, 720, 3, dict(c3, samples_per_side=1, num_of_rays=10)),
        # the reference CLI's defaults (main.py:95-102: N = 10, D = 3, one sample) on frames FULL of scattering pixels: the C2
        # scene with its ground plane, and the reference's own demo scene (main.py:40-93) at its 4:3 aspect
        "C2_scene_with_plane_cli_default_N10_spp1": (32, True, False, 1280, 720, 3, dict(c3, samples_per_side=1, num_of_rays=10)),
        "demo_scene_1280x960_cli_default_N10_spp1": ("demo", False, False, 1280, 960, 3, dict(c3, samples_per_side=1, num_of_rays=10)),
        "C5_flat_1280x720_10k_spheres": (10000, False, True, 1280, 720, 5, dict(renderer=abi.RENDERER_FLAT)),
        "C4_pathtracer_3840x2160_256sph_D5_spp64_one_gpu": (256, False, True, 3840, 2160, 3, C4["kw"]),
        "C4_same_PT_PCG_SAMPLE": (256, False, True, 3840, 2160, 3, dict(C4["kw"], pcg_mode=abi.PCG_SAMPLE)),
        "C4_share_of_rank_3_of_8": (256, False, True, 3840, 2160, 3, dict(C4["kw"], n_ranks=8, rank=3, row_block=8)),
        "C4_share_of_rank_3_of_8_PT_PCG_SAMPLE": (256, False, True, 3840, 2160, 3,
                                                  dict(C4["kw"], n_ranks=8, rank=3, row_block=8, pcg_mode=abi.PCG_SAMPLE)),
    }
    # primary + SHADOW rays (the metric's wording): PointLightRenderer over the C2 scene with two point lights
    cases["C2_pointlight_2_lights_1280x720"] = (32, True, False, 1280, 720, 7, dict(renderer=abi.RENDERER_POINTLIGHT, _lights=2))
    scene_cache = {}
    for name, (ns, plane, wide, W, H, reps, kw) in cases.items():
        kw = dict(kw)
        n_lights = kw.pop("_lights", 0)
        key = (ns, plane, wide, n_lights)
        if key not in scene_cache:
            for ds_old in scene_cache.values():
                ds_old[1].close()
            scene_cache.clear()
            demo_cam = None
            if ns == "demo":
                world, demo_cam = scenes.demo_world(clock=150.0)
            else:
                world = scenes.synthetic_world(ns, with_plane=plane, wide=wide)
            for l in range(n_lights):
                from pytracer_amd import hostmodel as hm

                world.add_light(hm.PointLight(hm.Vec(-3.0 + 4.0 * l, 6.0 - 9.0 * l, 8.0), hm.Color(1.0, 0.9, 0.8), 0.0))
            flat = flatten.flatten_world(world)
            scene_cache[key] = (flat, DeviceScene(flat, device=device), flatten.flatten_camera(demo_cam) if demo_cam is not None else None)
        flat, ds, own_cam = scene_cache[key]
        cam = own_cam if own_cam is not None else cam_for(W, H)
        par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")  # (a rank's share uses the top of it)
        rows[name] = kernel_row(ds, cam, par, out, reps, flat)
        rows[name]["two_frames_in_flight"] = in_flight_row(ds, cam, par, out, 40 if W * H > 2_000_000 else 120)
    for ds_old in scene_cache.values():
        ds_old[1].close()
    # the path tracer's second pass on C3: executed VALU instructions (PMC medians committed under profiles/) against
    # the SIMD-cycles of its launch -- the same fraction as roofline.frac, for the kernel VERDICT r1 named
    n_cu, clock_khz = device_info(device)
    for tag, fname in (("C3_pathtracer_1280x720_32sph_D3_spp16_N1", "pmc_c3_second_pass.json"),
                       ("C3_same_PT_PCG_SAMPLE", "pmc_c3_second_pass_sample.json"),
                       ("C4_same_PT_PCG_SAMPLE", "pmc_c4_second_pass_sample.json")):
        pmc = load_profile(fname)
        if pmc is not None and not fresh(pmc) and tag in rows:
            rows[tag]["second_pass_executed"] = {"frac": None, "reason": pmc["stale"]}
        if fresh(pmc) and tag in rows:
            valu, dur = pmc["counters"]["SQ_INSTS_VALU"], pmc["dur_us"] * 1e-6
            simd_cycles = n_cu * 4 * dur * clock_khz * 1e3
            pr = priced_issue(pmc["counters"], simd_cycles)
            rows[tag]["second_pass_executed"] = {
                "kernel": "pt_path_regions_kernel", "valu_wave_instructions_per_launch": valu, "kernel_us_under_pmc_collection": pmc["dur_us"],
                "valu_issue_utilisation_if_every_instruction_took_4_cycles": valu * 4 / simd_cycles,
                "valu_issue_utilisation": pr["frac"] if pr else None, "fp64_pipe_frac": pr["fp64_pipe_frac"] if pr else None,
                "frac_counter_active_inst_valu_x4": pr.get("frac_counter_active_inst_valu_x4") if pr else None,
                "unclassed_share": pr["unclassed_share"] if pr else None,
                # (tools/isa_mix.py on the same file: the disassembly priced per instruction, block counts bounded by the counters)
                "valu_issue_utilisation_bounds_from_disassembly": (
                    [b / simd_cycles for b in pmc["static_mix"]["valu_issue_cycles_bounds"]]
                    if pmc.get("static_mix", {}).get("valu_issue_cycles_bounds") and pmc["static_mix"].get("code_hash") == loaded_code_hash() else None),
                "code_hash": pmc.get("code_hash"), "kernel_resources": pmc.get("kernel_resources"),
                "source": pmc.get("source")}
    # C5: HBM bytes per frame of its two kernels (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, medians
    # committed under profiles/) over the kernel time measured here
    c5 = rows.get("C5_flat_1280x720_10k_spheres")
    tile, cell = load_profile("pmc_c5_tile.json"), load_profile("pmc_c5_cell.json")
    if c5 is not None and (tile is not None and not fresh(tile) or cell is not None and not fresh(cell)):
        c5["hbm"] = {"frac": None, "reason": (tile if not fresh(tile) else cell)["stale"]}
    if c5 is not None and fresh(tile) and fresh(cell) and "hbm_bytes_per_launch" in tile and "hbm_bytes_per_launch" in cell:
        nbytes = tile["hbm_bytes_per_launch"] + cell["hbm_bytes_per_launch"]
        gbs = nbytes / (c5["ms_per_frame"] * 1e-3) / 1e9
        c5["hbm"] = {"bytes_per_frame": nbytes, "achieved_GB_s": gbs, "peak_GB_s": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS,
                     "algorithmic_bytes_per_frame": 1280 * 720 * 12 + 10000 * (128 + 256 + 16),
                     "note": "pt_cell_kernel + pt_tile_kernel<FLAT, HIER>: pixels written once, the 10 000 shapes' bounds read per "
                             "32x32-pixel cell group, survivor lists written and re-read -- far from the HBM roof: the frame is "
                             "bound by the culling arithmetic", "code_hash": tile.get("code_hash"), "source": tile.get("source")}
    return rows


def boundary_rows(flat, device: int, rays_per_frame: int):
    """SURVEY.md 8(d) / BASELINE.md 4.3: the same C2 frame seen from the drop-in boundary -- ms per frame at the
    C-ABI = kernel + D2H (`pt_render` into a caller-owned host buffer), scene upload (flatten + H2D), and the
    Python fill of a reference-style HdrImage (a list of W*H Color objects).  None of these is `value`."""
    from pytracer_amd.tracer import _fill_image

    W, H = 1280, 720
    cam = cam_for(W, H)
    t0 = time.perf_counter()
    ds = DeviceScene(flat, device=device)
    upload_ms = (time.perf_counter() - t0) * 1e3
    rows = {"scene_upload_ms": upload_ms}
    ds.set_count_rays(False)  # (the frame's rays are known: the headline counted them)
    ds.set_dome_shortcut(False)  # (the headline's frame: every primary ray traced)
    for fmt, tag, nbytes in ((abi.OUT_F32, "f32", 4), (abi.OUT_F64, "f64", 8)):
        par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=fmt)
        for pinned in (True, False):
            ds.render(cam, par, pinned=pinned)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = ds.render(cam, par, pinned=pinned)
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = float(np.median(ts))
            rows[f"pt_render_host_{tag}_{'pinned' if pinned else 'pageable'}_ms"] = ms
            if pinned:
                st = ds.stats()
                rows[f"pt_render_host_{tag}_d2h_GB_s"] = W * H * 3 * nbytes / max(1e-9, (st.total_ms - st.kernel_ms) * 1e-3) / 1e9
    rows["value_at_c_abi"] = {
        "value": rays_per_frame / rows["pt_render_host_f32_pinned_ms"] / 1e3, "unit": "Mray/s",
        "ms_per_frame": rows["pt_render_host_f32_pinned_ms"],
        "note": "kernel + device-to-host copy of the fp32 frame (11.06 MB) into a page-locked caller buffer, one blocking "
                "pt_render call per frame: the copy runs at the host link's rate and is >90 % of the time",
    }

    # the `render` command's path since round 3 (main.py:197-213 on the device): the fp64 frame stays in HBM, the PFM
    # floats (12 B/pixel) and the tone-mapped bytes (3 B/pixel) are the only device-to-host traffic -- against the round-2
    # path: copy the fp64 frame (24 B/pixel) to the host, then upload it again for every post-processing step
    from pytracer_amd.postprocess import DeviceImage

    par64 = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F64)
    dev_frame = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")

    def cli_resident():
        ds.render_into(cam, par64, dev_frame.data_ptr(), dev_frame.numel() * 8, None)
        img = DeviceImage(dev_frame)
        pfm = img.pfm_payload()
        img.normalize_image(factor=1.0)
        img.clamp_image()
        return pfm, img.ldr_bytes()

    def cli_host_frame():
        img = DeviceImage.from_numpy(ds.render(cam, par64))
        pfm = img.pfm_payload()
        img.normalize_image(factor=1.0)
        img.clamp_image()
        return pfm, img.ldr_bytes()

    for fn, key in ((cli_resident, "cli_render_to_pfm_and_rgb8_resident_ms"), (cli_host_frame, "cli_render_to_pfm_and_rgb8_via_host_frame_ms")):
        fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        rows[key] = float(np.median(ts))
    rows["cli_note"] = ("render + write_pfm payload + normalize_image + clamp_image + LDR bytes (main.py:197-213) for the C2 frame: "
                        "`resident` leaves the fp64 frame in HBM (device-to-host: 11.06 MB of PFM floats + 2.76 MB of rgb8), "
                        "`via_host_frame` is the round-2 path (22.1 MB fp64 frame to the host, uploaded again per step)")

    class RefColor:  # the reference's Color: three attributes
        __slots__ = ("r", "g", "b")

        def __init__(self, r=0.0, g=0.0, b=0.0):
            self.r, self.g, self.b = r, g, b

    class RefImage:
        def __init__(self, w, h):
            self.width, self.height = w, h
            self.pixels = [RefColor() for _ in range(w * h)]

    t0 = time.perf_counter()
    img = RefImage(W, H)
    rows["python_hdrimage_constructor_ms"] = (time.perf_counter() - t0) * 1e3  # (what HdrImage(W, H) itself costs: hdrimages.py:70)
    frame64 = np.asarray(out, dtype=np.float64)
    t0 = time.perf_counter()
    _fill_image(img, frame64, lazy=True)
    rows["python_hdrimage_first_fill_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    px = img.pixels[W * 360 + 640]
    rows["python_hdrimage_first_pixel_read_us"] = (time.perf_counter() - t0) * 1e6
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        _fill_image(img, frame64, lazy=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    rows["python_hdrimage_lazy_fill_ms"] = float(np.median(ts))
    img2 = RefImage(W, H)
    t0 = time.perf_counter()
    _fill_image(img2, frame64)
    rows["python_hdrimage_fill_ms"] = (time.perf_counter() - t0) * 1e3
    rows["python_hdrimage_note"] = ("handing a frame to a reference-style HdrImage (a list of W*H Color objects, hdrimages.py:70): "
                                    "`fill` (the DEFAULT since round 5) fills the existing list in place with 921 600 new Color objects, as "
                                    "the reference's set_pixel loop does; `lazy_fill` (GpuImageTracer(lazy_pixels=True), opt-in) installs "
                                    "pytracer_amd.pixels.LazyPixels over the numpy frame (a Color is made when an index is first read and "
                                    "kept from then on); `first_fill` is the lazy one on a freshly constructed image and is dominated by "
                                    "FREEING the 921 600 black Color objects its constructor made (`constructor_ms`: not this library's)")
    assert (px.r, px.g, px.b) == tuple(frame64[360, 640].tolist())
    ds.close()
    return rows


_loaded_code_hash = None


def loaded_code_hash():
    """sha256 of the device code of the library this process loaded (pytracer_amd.build.code_hash of _lib.lib_path())."""
    global _loaded_code_hash
    if _loaded_code_hash is None:
        from pytracer_amd import _lib
        from pytracer_amd.build import code_hash

        _loaded_code_hash = code_hash(_lib.lib_path())
    return _loaded_code_hash


def load_profile(name, want_hash="loaded"):
    """A committed PMC summary (profiles/pmc_*.json, tools/pmc_summary.py) -- but only if it measured THIS binary: the file
    records the sha256 of the device code it was collected on, and a file of another build (or of a round before files
    carried a hash) comes back as {"stale": reason} so that nothing is priced from it."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pmc = json.load(f)
    if want_hash is None:
        return pmc
    have = loaded_code_hash() if want_hash == "loaded" else want_hash
    if pmc.get("code_hash") != have:
        return {"stale": f"profiles/{name} was collected on device code {str(pmc.get('code_hash'))[:16]}, this run loaded {have[:16]}: "
                         "not priced (re-run tools/prof_bench.sh on this build)"}
    return pmc


def fresh(pmc):
    return pmc is not None and "stale" not in pmc


def fence(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()


def timed_loop(ds, loop, steps, dist, gather, events=True):
    """K frames between two fences; -> (wall s, summed kernel ms, launches).  With `events` every launch is timed by
    an event pair carried in its own dispatch (they cost a few microseconds of pipelining per launch, so the
    headline loop runs without them and a second, identical loop supplies the per-launch figure)."""
    ds.set_count_rays(False)
    ds.set_timing(events)
    loop.step(0, gather=gather)  # one uncounted frame so the timed region starts from the steady state
    loop.finish()
    fence(dist)
    if events:
        ds.profile_begin(steps + 2)
    t0 = time.perf_counter()
    for i in range(steps):
        loop.step(i, gather=gather)
    loop.finish()  # (the gather runs one frame behind the render: the last frame is assembled here, inside the timed region)
    fence(dist)
    elapsed = time.perf_counter() - t0
    kernel_total_ms, launches = ds.profile_end() if events else (0.0, 0)
    ds.set_timing(True)
    return elapsed, kernel_total_ms, launches


def bracketed_loop(ds, loop, steps, repeats=1):
    """The K launches of the timed region between ONE pair of HIP events recorded on the stream the kernels are
    launched on (no host synchronisation, no per-launch events in between): -> average duration of a launch in
    seconds = what rocprofv3's kernel trace shows for the same kernel run back to back.  Median of `repeats` such loops."""
    ds.set_count_rays(False)
    ds.set_timing(False)
    per_launch = []
    for _ in range(max(1, repeats)):
        loop.step(0, gather=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(loop.stream)
        for i in range(steps):
            loop.step(i, gather=False)
        e1.record(loop.stream)
        e1.synchronize()
        per_launch.append(e0.elapsed_time(e1) * 1e-3 / steps)
    ds.set_timing(True)
    return float(np.median(per_launch))


def pre_roll(loop, min_ms=30.0, chunk=50):
    """Untimed launches of the headline kernel until `min_ms` of wall time have passed (chunks of `chunk` frames, a device
    synchronisation behind each): the GPU's clocks ramp over the first milliseconds of work, and a 20-launch timed loop of
    14-us frames (the driver's command) is over before they have.  -> (launches, ms)."""
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < min_ms:
        for i in range(chunk):
            loop.step(i, gather=False)
        n += chunk
        torch.cuda.synchronize()
    return n, (time.perf_counter() - t0) * 1e3


def in_flight_rows(flat, cam, par, steps, local_rank, rays_per_step, pmc, n_simd, clock_hz):
    """The same K frames with 2 and 4 of them in flight (pytracer_amd.pipeline.FramePipeline: one scene handle and one
    HIP stream per slot): what an animation gets.  NOT the headline -- `value` above is one frame after the other on one
    stream -- but the same kernel, the same frames (checked), and the figure that says how much of the chip one frame
    leaves idle."""
    from pytracer_amd.pipeline import FramePipeline

    rows = {"note": "K frames dealt round-robin to n scene handles / HIP streams, wall clock between two device "
                    "synchronisations; valu_issue_utilisation = the headline kernel's priced issue cycles per launch "
                    "(roofline.executed) / (SIMDs x clock x time per frame)"}
    H, W = par.height, par.width
    ref = None
    for n in (1, 2, 4):
        with FramePipeline(flat, n_in_flight=n, device=local_rank) as pipe:
            pipe.set_count_rays(False)
            pipe.set_timing(False)
            pipe.set_dome_shortcut(False)  # (the headline's frames: every primary ray traced)
            outs = [torch.empty((H, W, 3), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(n)]
            for i in range(2 * n):
                pipe.submit(cam, par, outs[i % n])
            pipe.wait()
            torch.cuda.synchronize()
            if n == 1:  # the reference frame only: one frame after the other is the headline itself
                ref = outs[0].clone()
                continue
            t0 = time.perf_counter()
            for i in range(steps):
                pipe.submit(cam, par, outs[i % n])
            pipe.wait()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = all(bool(torch.equal(ref, o)) for o in outs)
        row = {"value": rays_per_step * steps / dt / 1e6, "unit": "Mray/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
               "frames_identical_to_one_stream": same}
        if pmc is not None:
            pr = priced_issue(pmc["counters"], n_simd * (dt / steps) * clock_hz)
            if pr is not None:
                row["valu_issue_utilisation"] = pr["frac"]
                row["fp64_pipe_frac"] = pr["fp64_pipe_frac"]
        rows[str(n)] = row
    return rows


def run_single(args, local_rank):
    W, H = 1280, 720
    world = scenes.synthetic_world(32, with_plane=True)
    flat = flatten.flatten_world(world)
    cam = cam_for(W, H)
    ds = DeviceScene(flat, device=local_rank)
    par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
    loop = ShardedFrameLoop(ds, cam, par, row_block=8)

    # The headline loop runs with the dome shortcut OFF (pt_set_dome_shortcut(0)): every primary ray of the frame is generated and
    # handed to a world query, which is what SURVEY.md 8(d) counts ("all rays passed to a world query").  The product's default
    # (shortcut on: tiles that can only see the sphere around the camera are settled without rays, exact) is the side row
    # `frame_with_dome_shortcut`, whose rate counts only the rays that were traced.
    ds.set_dome_shortcut(False)
    ds.set_count_rays(True)
    for i in range(max(1, args.warmup)):
        loop.step(i, gather=False)
    fence(None)
    ds.sync()
    st = ds.stats()
    # rays per frame are counted (in-kernel counter) during warm-up; the workload is deterministic, so the timed
    # steps run without the counter: one step == exactly one render-kernel launch
    rays_per_step, resolved = int(st.n_rays), int(st.n_rays_resolved)
    if resolved != 0:
        raise RuntimeError(f"{resolved} rays were settled without a world query although the dome shortcut is off")
    n_wg = st.grid
    # the headline: K frames back to back -- the K-step loop REPEATED (20 launches of 15 us are a 0.3 ms timed region);
    # `value` / `ms_per_step` are the MEDIAN repeat, min and max in the detail file.  Before the first timed loop the
    # same launches run untimed for >= 30 ms (clocks), and the loop is repeated until >= 20 ms have been timed (at least
    # --repeats times): K stays what --steps says for every loop
    ds.set_count_rays(False)
    ds.set_timing(False)
    pre_n, pre_ms = pre_roll(loop, args.pre_roll_ms)
    est_loop_s = max(1e-6, pre_ms * 1e-3 / max(1, pre_n) * args.steps)
    repeats = max(1, args.repeats, min(400, int(np.ceil(args.min_timed_ms * 1e-3 / est_loop_s))))
    elapsed_all = [timed_loop(ds, loop, args.steps, None, False, events=False)[0] for _ in range(repeats)]
    elapsed = float(np.median(elapsed_all))
    # the same loop with TEN TIMES the steps (a few repeats): what a timed region costs beyond its frames -- the first launch's
    # way to the GPU and the last synchronisation's way back, ~16 us per region -- divides by K: at the driver's K = 20 it is
    # 0.8 us of every step, at K = 200 a tenth of that.  Reported so that the detail file explains its own K.
    k10 = 10 * args.steps
    el10 = float(np.median([timed_loop(ds, loop, k10, None, False, events=False)[0] for _ in range(max(3, min(repeats, 400 // max(1, args.steps))))]))
    parity = parity_check(flat, cam, par, loop.image())  # the frame those loops left in HBM, against the oracle's
    elapsed_ev, kernel_total_ms, launches = timed_loop(ds, loop, args.steps, None, False, events=True)  # the same, every launch with its own event pair
    per_launch_pair_s = kernel_total_ms / max(launches, 1) * 1e-3
    avg_kernel_s = bracketed_loop(ds, loop, args.steps, repeats)  # the same K launches between one event pair on their stream (median of the repeats)
    ms_per_step = elapsed / args.steps * 1e3

    # the product's default: the same frame with the dome shortcut on (a side row: 5 loops of K launches against the headline's
    # pre-roll and repeats, so that the bench command's launches of this kernel under rocprofv3 are essentially the headline's)
    dome_row = None
    if not args.headline_only:
        ds.set_dome_shortcut(True)
        ds.set_count_rays(True)
        for i in range(3):
            loop.step(i, gather=False)
        fence(None)
        ds.sync()
        st_on = ds.stats()
        traced_on = int(st_on.n_rays) - int(st_on.n_rays_resolved)
        n_dome_on = args.steps  # (the same K as the headline: a timed region's fixed cost divides by it)
        el_on = float(np.median([timed_loop(ds, loop, n_dome_on, None, False, events=False)[0] for _ in range(5)]))
        parity_on = parity_check(flat, cam, par, loop.image())
        dome_row = {"ms_per_frame": el_on / n_dome_on * 1e3, "rays_traced_per_frame": traced_on,
                    "rays_settled_without_a_query_per_frame": int(st_on.n_rays_resolved),
                    "traced_Mray_s": traced_on * n_dome_on / el_on / 1e6, "steps": n_dome_on, "bit_identical": parity_on["bit_identical"],
                    "note": "the library's default (pt_set_dome_shortcut(1)): tiles whose only possible hit is the sphere around the camera "
                            "are settled without generating rays (exact, DESIGN.md 4 item 8); only the traced rays are counted in the rate"}
        ds.set_dome_shortcut(False)
        ds.set_count_rays(False)

    n_cu, clock_khz = device_info(local_rank)
    n_simd = n_cu * 4
    clock_hz = clock_khz * 1e3
    n_sph = int((flat.kind == abi.SHAPE_SPHERE).sum())
    n_pl = int((flat.kind == abi.SHAPE_PLANE).sum())
    flops = rays_per_step * (n_sph * FLOP_PER_SPHERE_TEST + n_pl * FLOP_PER_PLANE_TEST)
    alg_bytes = rays_per_step * 12 + n_wg * flat.n_shapes * 104
    pmc = load_profile("pmc_c2.json")
    roofline = {
        "bound": "valu_issue",
        "achieved": None, "peak": None, "unit": "T lane-op/s (executed VALU lane-operations, any type)", "frac": None, "traffic": None,
        "kernel": "pt_tile4_kernel<FLAT> (16x16 tiles, four pixels per lane, culled shape lists, hoisted scale+translate tests), dome shortcut off",
        "avg_kernel_ms": avg_kernel_s * 1e3, "launches_timed": args.steps * repeats,
        "avg_kernel_ms_method": "K launches back to back between one HIP event pair recorded on their stream / K (rocprofv3's "
                                "kernel trace of the same command shows the same average for launches run back to back)",
        "per_launch_event_pair_ms": per_launch_pair_s * 1e3,
        "per_launch_event_pair_note": "every launch with its own event pair in the dispatch: includes the events' own pipeline "
                                      "bubble (a few us per launch; rocprofv3 sees the kernels of this loop no longer than the others)",
        "ms_per_step_of_the_per_launch_event_loop": elapsed_ev / args.steps * 1e3,
        "note": "no dense contraction: MFMA unused; the path is bound by vector issue and dependent latency, not HBM (SURVEY.md 8d). "
                "`frac` = VALU issue utilisation = sum over instruction classes of (wave-instructions per launch, rocprofv3 PMC "
                "SQ_INSTS_VALU_*, profiles/pmc_c2.json) x (SIMD-cycles per instruction of that class, MEASURED by "
                "tools/micro/issue.hip, profiles/r03_issue_rates.txt: fp64 4, fp32/int32 2, fp64 transcendental 16, ...; "
                "instructions no class counter covers priced at 3, see frac_other_at_2 / _at_4) / (SIMDs x clock x the kernel's "
                "average duration measured here over every timed launch).  `achieved` = executed VALU lane-operations per "
                "second; `peak` = achieved / frac = the rate at which this instruction mix would issue with no SIMD ever idle.  "
                "fp64_pipe_frac counts the fp64 instructions alone (x 4, transcendentals x 16); salu_issue_frac the scalar "
                "instructions (x 4 per SIMD); frac_with_salu_coissue adds the vector issue a scalar instruction costs (1.4 "
                "cycles, measured: profiles/r03_issue_pmc_probe.txt); frac_counter_active_inst_valu_x4 = SQ_ACTIVE_INST_VALU x 4 / "
                "SIMD-cycles, a pure counter figure that charges every fp32 / int32 instruction 4 cycles (upper bracket); "
                "executed.code_hash = the device code the counters were collected on, compared with the library this run loaded.",
    }
    if avg_kernel_s * 1e3 > ms_per_step * 1.05:
        roofline["avg_kernel_ms"] = None
        roofline["note"] += " (kernel time discarded: it exceeded the step time)"
    elif pmc is not None and not fresh(pmc):
        roofline["reason_frac_is_null"] = pmc["stale"]
        roofline["note"] += " (NOT PRICED: " + pmc["stale"] + ")"
    elif pmc is not None:
        valu = pmc["counters"]["SQ_INSTS_VALU"]
        simd_cycles = n_simd * avg_kernel_s * clock_hz
        ach = valu * 64 / avg_kernel_s / 1e12
        pr = priced_issue(pmc["counters"], simd_cycles)
        roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
        roofline["achieved"] = ach
        roofline["executed"] = {"valu_wave_instructions_per_launch": valu,
                                "salu_wave_instructions_per_launch": pmc["counters"].get("SQ_INSTS_SALU"),
                                "kernel_us_under_pmc_collection": pmc.get("dur_us"), "simds": n_simd, "clock_GHz": clock_hz / 1e9,
                                "kernel_profiled": pmc.get("kernel"), "source": pmc.get("source"),
                                "code_hash": pmc.get("code_hash"), "code_hash_matches_loaded_library": pmc.get("code_hash") == loaded_code_hash(),
                                "kernel_resources": pmc.get("kernel_resources")}
        if pr is not None:
            roofline["frac"] = pr["frac"]
            roofline["peak"] = ach / pr["frac"]
            roofline.update({k: pr[k] for k in ("frac_other_at_2", "frac_other_at_4", "fp64_pipe_frac", "salu_issue_frac", "frac_with_salu_coissue",
                                                "frac_counter_active_inst_valu_x4", "wave_cycles_waiting_any_frac", "wave_cycles_waiting_inst_frac") if k in pr})
            roofline["executed"].update({k: pr[k] for k in ("mean_cycles_per_valu_instruction", "instructions_by_class", "unclassed_share", "fp64_instruction_share")})
            roofline["frac_method"] = "class counters, unclassed instructions at 3 cycles"
            mix = pmc.get("static_mix")
            if mix:  # tools/isa_mix.py: the kernel's disassembly priced instruction by instruction, block counts bounded by the counters
                roofline["executed"]["static_mix"] = {k: v for k, v in mix.items() if k not in ("static_valu_mix", "dynamic_mix_at_min")}
                if mix.get("valu_issue_cycles_bounds") and mix.get("code_hash") == loaded_code_hash():
                    # every VALU instruction of the disassembly at its measured price; how often each basic block runs bounded by
                    # the control-flow graph and all per-launch counters: frac = the middle of what they allow, the bounds beside it
                    lo_c, hi_c = mix["valu_issue_cycles_bounds"]
                    roofline["frac_bounds_from_disassembly"] = [lo_c / simd_cycles, hi_c / simd_cycles]
                    roofline["frac_class_counters_other_at_3"] = roofline["frac"]
                    roofline["frac"] = 0.5 * (lo_c + hi_c) / simd_cycles
                    roofline["peak"] = ach / roofline["frac"]
                    roofline["frac_method"] = ("tools/isa_mix.py: the kernel's disassembly, every VALU instruction at its MEASURED issue cost "
                                               "(tools/micro/issue.hip, issue2.hip), basic-block execution counts bounded by two linear "
                                               "programmes over the control-flow graph and the per-launch PMC counters; frac = midpoint")
                    salu = pmc["counters"].get("SQ_INSTS_SALU", 0.0)
                    roofline["frac_with_salu_coissue"] = (0.5 * (lo_c + hi_c) + salu * SALU_COISSUE_PENALTY_CYCLES) / simd_cycles
        else:
            roofline["note"] += " (profiles/pmc_c2.json lacks the per-class counters: frac not priced)"
    roofline["algorithmic_equivalent"] = {
        "flop_per_launch": flops, "TFLOP_s": flops / avg_kernel_s / 1e12,
        "note": "SURVEY.md 8(d) recipe: 54 flop per ray-sphere and 36 per ray-plane test, every ray x every shape, / kernel "
                "time.  NOT a hardware fraction: the kernel executes far less (tile culling, hoisted origin, scale+translate "
                "fast path, dome shortcut) with bit-identical results, so this figure may exceed the fp64 peak (78.6 TFLOP/s "
                "with FMA); it says how fast a brute-force loop would have to be to keep up",
    }
    roofline["hbm"] = {"achieved": alg_bytes / avg_kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": alg_bytes / avg_kernel_s / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": alg_bytes,
                       "note": "12 B per pixel written + the scene once per workgroup: not the bound"}
    result = {
        "metric": "Mray/s (primary+shadow; rays handed to a world query) at 1280x720 per GPU, C2: 32 spheres + 1 plane, FlatRenderer",
        "value": rays_per_step * args.steps / elapsed / 1e6,
        "unit": "Mray/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "repeats": {"n": repeats, "ms_per_step_median": ms_per_step, "ms_per_step_min": min(elapsed_all) / args.steps * 1e3,
                    "ms_per_step_max": max(elapsed_all) / args.steps * 1e3,
                    "ms_per_step_first_5": [e / args.steps * 1e3 for e in elapsed_all[:5]],
                    "ms_per_step_last_5": [e / args.steps * 1e3 for e in elapsed_all[-5:]],
                    "timed_ms_total": sum(elapsed_all) * 1e3,
                    "pre_roll": {"launches": pre_n, "ms": pre_ms, "note": "untimed launches of the same kernel before the first timed loop (clock ramp)"},
                    "note": "the K-step timed loop run `n` times back to back (until >= 20 ms are timed); value and ms_per_step are the median repeat"},
        "same_loop_at_10x_steps": {"steps": k10, "ms_per_step": el10 / k10 * 1e3,
                                   "fixed_cost_per_timed_region_us": (elapsed - el10 / 10.0) / 0.9 * 1e6,
                                   "note": "the identical timed loop with 10 x K steps: ms_per_step(K) - ms_per_step(10 K) is the timed region's "
                                           "fixed cost (first launch in, last synchronisation out) spread over K steps, not kernel time"},
        "hip_force_dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG"),
        "code_hash": loaded_code_hash(),
        "parity_check": parity,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"C2 flat {W}x{H}, 32 spheres + 1 plane, S=0, every primary ray traced (dome shortcut off), fp32 RGB resident in HBM",
                   "width": W, "height": H, "n_shapes": flat.n_shapes, "renderer": "FlatRenderer"},
        "ray_shape_tests_per_s": rays_per_step * flat.n_shapes * args.steps / elapsed,
        "rays_per_step": rays_per_step,
        "value_note": "every ray counted in `value` was generated and handed to a world query (dome shortcut off for the headline "
                      "loop: pt_set_dome_shortcut(0)); the library's default frame is frame_with_dome_shortcut",
        "frame_with_dome_shortcut": dome_row,
        "roofline": roofline,
    }
    if not parity["bit_identical"]:  # a fast frame that is not the reference's frame is not a result
        result["error"] = f"parity_check failed: the timed frame differs from the oracle's in {parity.get('pixels_differing', '?')} pixels"
        result["value_unchecked"], result["value"] = result["value"], None
    elif dome_row is not None and not dome_row["bit_identical"]:
        result["error"] = "parity_check failed: the frame rendered with the dome shortcut differs from the oracle's"
        result["value_unchecked"], result["value"] = result["value"], None
    # in-flight rows: the headline's frames (shortcut off) on 2 and 4 streams
    if not args.no_in_flight:  # measured by THIS run: a few hundred frames, behind the headline loops
        result["frames_in_flight"] = in_flight_rows(flat, cam, par, args.in_flight_steps, local_rank, rays_per_step, pmc if fresh(pmc) else None,
                                                    n_simd, clock_hz)
    ds.close()
    if not args.no_extras:
        result["extra"] = extra_rows(local_rank)
        result["boundary"] = boundary_rows(flat, local_rank, rays_per_step)
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(flat)
    emit(result, compact_single(result))


# ---------------------------------------------------------------------------------------------------------------------------
# The result line.  bench.py's LAST stdout line is a COMPACT object (target <= 4 KB, hard limit 8 KB: tests/test_bench_helpers.py)
# with a fixed set of keys; everything else the run measured goes to bench_detail.json next to this script and to stderr.
# (Rounds 1-5 printed everything on the one line; it grew to 21.7 KB and the driver could no longer read it.)
COMPACT_LIMIT = 8192
DETAIL_NAME = "bench_detail.json"


def _pick(d, keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


def _short(x, n=200):
    return x if x is None or len(str(x)) <= n else str(x)[: n - 3] + "..."


def compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac"))
    out["frac_bounds"] = r.get("frac_bounds_from_disassembly")
    out["traffic"] = r.get("traffic")
    out["algorithmic_bytes"] = (r.get("hbm") or {}).get("algorithmic_bytes_per_launch")
    out["hbm_frac"] = (r.get("hbm") or {}).get("frac")
    out["kernel"] = _short(r.get("kernel"), 120)
    out["avg_kernel_ms"] = r.get("avg_kernel_ms")
    out["code_hash_matches_loaded_library"] = (r.get("executed") or {}).get("code_hash_matches_loaded_library")
    if r.get("reason_frac_is_null"):
        out["reason_frac_is_null"] = _short(r["reason_frac_is_null"], 240)
    return out


def compact_cpu_baseline(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "ms_per_frame", "one_core_Mray_s", "cpu_model"))
    out["sample"] = _short(c.get("sample"), 160)
    out["interpreted_Mray_s"] = (c.get("interpreted") or {}).get("value")
    return out


def compact_single(full):
    """The N = 1 line: BASELINE.json's metric on C2 plus `roofline` and `cpu_baseline`, nothing that is not a number, a short
    name or a flag."""
    dome = full.get("frame_with_dome_shortcut")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["config"] = _pick(full.get("config"), ("workload", "width", "height", "n_shapes", "renderer"))
    out["rays_per_step"] = full.get("rays_per_step")
    out["ray_shape_tests_per_s"] = full.get("ray_shape_tests_per_s")
    out["parity_check"] = _pick(full.get("parity_check"), ("bit_identical",))
    out["roofline"] = compact_roofline(full.get("roofline"))
    out["cpu_baseline"] = compact_cpu_baseline(full.get("cpu_baseline"))
    out["ms_per_frame_at_c_abi"] = ((full.get("boundary") or {}).get("value_at_c_abi") or {}).get("ms_per_frame")
    out["frame_with_dome_shortcut"] = _pick(dome, ("ms_per_frame", "rays_traced_per_frame", "traced_Mray_s", "bit_identical"))
    out["code_hash"] = (full.get("code_hash") or "")[:16] or None
    if full.get("error"):
        out["error"] = _short(full["error"], 400)
    return out


def compact_multi(full):
    """The N > 1 line: the C4 frame strong-scaled over the ranks; rates count only rays that went through a world query."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data", "ranks_seen", "backend")}
    out["config"] = _pick(full.get("config"), ("workload", "width", "height", "n_shapes", "renderer", "pcg_mode", "frames_in_flight_per_rank"))
    out["rays_traced_per_frame"] = full.get("rays_traced_per_frame")
    out["ms_per_frame_without_gather"] = (full.get("without_gather") or {}).get("ms_per_step")
    g = full.get("gather") or {}
    out["gather"] = {"used": g.get("used"), "ms_per_frame": g.get("probe_ms_per_frame"), "bytes_per_frame_whole": g.get("gather_bytes_per_frame"),
                     "bytes_per_frame_sent": full.get("gather_bytes_per_frame_sent"), "fallback_reason": _short(g.get("fallback_reason"), 240)}
    out["gather_check"] = full.get("gather_check")
    oc = full.get("oracle_check")
    out["parity_check"] = _pick(oc, ("checked", "bit_identical", "pixels_beyond_1e-5", "rays_match")) if isinstance(oc, dict) else None
    n1 = full.get("n1_same_workload")
    out["n1_same_workload"] = _pick(n1, ("value", "ms_per_step")) if isinstance(n1, dict) else None
    out["speedup"] = full.get("speedup")
    # (the N = 1 line measures C2, 15 us per frame, nothing to shard: dividing this line's value by that one's is not a scaling
    #  efficiency -- `speedup` against n1_same_workload is)
    out["comparable_with_the_n1_line"] = False
    out["rank_share_imbalance"] = full.get("rank_share_imbalance")
    a = full.get("at_1280x720")
    out["at_1280x720"] = _pick(a, ("value", "ms_per_step", "speedup")) if isinstance(a, dict) else None
    r = full.get("c2_replicas")
    out["c2_replicas"] = _pick(r, ("value", "ms_per_step")) if isinstance(r, dict) else None
    out["code_hash"] = (full.get("code_hash") or "")[:16] or None
    if full.get("error"):
        out["error"] = _short(full["error"], 400)
    return out


def detail_path():
    """Where the detail file goes: next to this script; if that is not writable, under gpurun_out/ or the temp directory."""
    import tempfile

    for d in (ROOT, os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            probe = os.path.join(d, ".bench_detail_probe")
            with open(probe, "w"):
                pass
            os.remove(probe)
            return os.path.join(d, DETAIL_NAME)
        except OSError:
            continue
    return None


def compact_dumps(compact):
    """-> the JSON text of the compact line, bounded: floats to 6 significant digits; should it still exceed COMPACT_LIMIT (it
    cannot with the fixed key set, but a line that does not parse loses the round), optional sub-objects are dropped one by one."""
    def rnd(x):
        if isinstance(x, float):
            return float(f"{x:.6g}") if x == x and abs(x) != float("inf") else None
        if isinstance(x, dict):
            return {k: rnd(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [rnd(v) for v in x]
        return x

    obj = rnd(compact)
    text = json.dumps(obj, separators=(",", ":"))
    for k in ("at_1280x720", "c2_replicas", "n1_same_workload", "frame_with_dome_shortcut", "gather", "config"):
        if len(text) < COMPACT_LIMIT:
            break
        obj.pop(k, None)
        obj["truncated"] = True
        text = json.dumps(obj, separators=(",", ":"))
    return text


def write_detail(full):
    """-> where the detail file was written (relative to this script when next to it), or a reason why not."""
    path = detail_path()
    if not path:
        return "not written: no writable directory"
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
    except OSError as e:
        return f"not written: {e}"
    return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path


def emit(full, compact):
    """Detail -> bench_detail.json + stderr; the compact line -> stdout, LAST."""
    compact["detail_file"] = write_detail(full)
    try:
        print("[bench detail] " + json.dumps(full), file=sys.stderr, flush=True)
    except (OSError, ValueError):
        pass
    sys.stdout.flush()
    print(compact_dumps(compact), flush=True)


class SceneGroup:
    """n handles of the same scene on one GPU (a frame in flight each, pytracer_amd/dist.py: ShardedFrameLoop); the
    switches go to all of them, counters and times are summed, `stats()` is the first handle's last frame."""

    def __init__(self, flat, n, device):
        first = DeviceScene(flat, device=device)
        self.scenes = [first] + [first.clone() for _ in range(max(1, n) - 1)]

    def set_count_rays(self, on):
        for d in self.scenes:
            d.set_count_rays(on)

    def set_timing(self, on):
        for d in self.scenes:
            d.set_timing(on)

    def set_dome_shortcut(self, on):
        for d in self.scenes:
            d.set_dome_shortcut(on)

    def sync(self):
        for d in self.scenes:
            d.sync()

    def profile_begin(self, capacity):
        for d in self.scenes:
            d.profile_begin(capacity)

    def profile_end(self):
        total, launches = 0.0, 0
        for d in self.scenes:
            t, n = d.profile_end()
            total += t
            launches += n
        return total, launches

    def stats(self):
        return self.scenes[0].stats()

    def close(self):
        for d in self.scenes:
            d.close()


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr (rank 0 of a multi-rank run): which phase the job is in and since when -- a run of several minutes
    that prints nothing looks hung from outside, and a real hang should say where."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


class Watchdog:
    """A per-phase deadline on every rank (a thread; the main thread may sit in a collective that never returns -- torch releases
    the GIL there).  When a phase overruns: rank 0 prints the FALLBACK line if one has been stashed (the whole-shard gather's
    measurement, complete and checked, marked with what failed) or an `error` line, and every rank leaves with os._exit -- status
    0 with a fallback (all ranks know whether one exists: it is stashed behind an all-reduce), 1 without.  No rank re-execs."""

    def __init__(self, rank):
        import threading

        self.rank = rank
        self.lock = threading.Lock()
        self.phase, self.deadline = None, None
        self.fallback = None       # rank 0: (full, compact) of the measurement that is already complete
        self.have_fallback = False  # every rank
        self.error_stub = {"metric": "Mray/s", "value": None, "unit": "Mray/s"}
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def arm(self, phase, seconds):
        with self.lock:
            self.phase, self.deadline = phase, time.perf_counter() + seconds

    def disarm(self):
        with self.lock:
            self.phase, self.deadline = None, None

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                phase, deadline = self.phase, self.deadline
            if deadline is None or time.perf_counter() < deadline:
                continue
            self.bail(f"phase '{phase}' exceeded its deadline")

    def bail(self, why):
        """End the job from this rank NOW (any thread): rank 0 prints the line measured so far, every rank leaves with the same
        status -- 0 when a complete measurement exists, 1 otherwise."""
        with self.lock:
            if getattr(self, "bailing", False):
                time.sleep(30.0)  # (another thread of this rank is already on its way out)
                os._exit(0 if self.have_fallback else 1)
            self.bailing = True
        if self.rank == 0:
            print(f"[bench] {why}: leaving", file=sys.stderr, flush=True)
            if self.fallback is not None:
                full = dict(self.fallback[0])
                full["gather"] = dict(full.get("gather") or {})
                if full["gather"].get("used") != "sparse":
                    full["gather"].update(used="whole", fallback_reason=f"{why} (the job ended there)")
                full["phases_not_run"] = why
                emit(full, compact_multi(full))
            else:
                stub = dict(self.error_stub, error=why)
                emit(stub, dict(stub))
        else:
            time.sleep(3.0)  # (rank 0 prints first)
        os._exit(0 if self.have_fallback else 1)


PHASE_DEADLINE_S = float(os.environ.get("PT_BENCH_PHASE_S", "150"))


class Agreement:
    """Ranks agree after every phase on whether all of them got through it: a rank that raised still enters this
    all-reduce, so the others learn of it here instead of waiting in the next collective until the watchdog ends
    the job -- and rank 0 can still print its line (with `error`).  Every phase runs under the Watchdog's deadline."""

    def __init__(self, dist, watchdog=None):
        self.dist = dist
        self.error = None
        self.soft = {}
        self.watchdog = watchdog
        self.log = []  # (phase, seconds)

    def attempt(self, phase, fn, deadline_s=None):
        """Like run(), but a failure is remembered as `soft[phase]` only: the job goes on (used for a choice that has a
        fallback)."""
        saved, self.error = self.error, None
        ok = self.run(phase, fn, deadline_s)
        if not ok:
            self.soft[phase] = self.error
        self.error = saved
        return ok

    def run(self, phase, fn, deadline_s=None):
        """Run `fn()` on this rank; -> True iff EVERY rank completed it (collective)."""
        progress(phase)
        t0 = time.perf_counter()
        if self.watchdog is not None:
            self.watchdog.arm(phase, deadline_s or PHASE_DEADLINE_S)
        if self.error is None:
            try:
                fn()
            except Exception as e:  # noqa: BLE001  (RCCL / driver errors surface as RuntimeError subclasses)
                self.error = f"{phase}: {type(e).__name__}: {e}"[:400]
        try:
            flag = torch.tensor([0 if self.error is None else 1], dtype=torch.int32, device="cuda")
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if int(flag.item()) and self.error is None:
                self.error = f"{phase}: another rank failed"
        except Exception as e:  # noqa: BLE001
            self.error = self.error or f"{phase}: agreement failed: {type(e).__name__}: {e}"[:400]
            if self.watchdog is not None:  # the ranks can no longer talk (a peer has left): nothing collective can follow
                self.watchdog.bail(f"phase '{phase}': the ranks' agreement failed ({type(e).__name__}: a peer has left?)")
        if self.watchdog is not None:
            self.watchdog.disarm()
        self.log.append((phase, time.perf_counter() - t0))
        return self.error is None


C3 = dict(n_spheres=32, wide=False, W=1280, H=720,
          kw=dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3,
                  path_state=45, path_seq=54))


def rank_spread(dist, values, world_size):
    """One number per rank and key -> {key: {min, median, max, per_rank}} (all-gathered; collective)."""
    keys = sorted(values)
    mine = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device="cuda")
    every = [torch.empty_like(mine) for _ in range(world_size)]
    dist.all_gather(every, mine)
    table = torch.stack(every).cpu().numpy()  # [rank, key]
    return {k: {"min": float(table[:, i].min()), "median": float(np.median(table[:, i])), "max": float(table[:, i].max()),
                "per_rank": [float(v) for v in table[:, i]]} for i, k in enumerate(keys)}


def choose_gather(whole_ms, sparse_ms, sparse_error):
    """Which gather the headline uses, from what was measured: -> (used, fallback_reason).  The whole-shard gather (one padded
    transfer per remote rank: the plainest use of RCCL) is measured FIRST and is what the line falls back on; the sparse one is
    used only if it completed on every rank, assembled the same frame and was faster."""
    if sparse_error:
        return "whole", f"sparse failed: {sparse_error}"[:300]
    if sparse_ms is None:
        return "whole", "sparse not measured"
    if whole_ms is None:
        return "sparse", None
    return ("sparse", None) if sparse_ms <= whole_ms else ("whole", None)


def sharded_workload(args, cfg, mode, ds, cam, rank, world_size, dist, agree, gather_sparse, steps, full=True):
    """One frame of `cfg` strong-scaled over the ranks under PCG mode `mode`, shards gathered sparse or whole: warm-up and ray
    count, the timed loop with the gather; with `full` also the loop without the gather, the per-phase probe, the gather check
    against rank 0 alone, the same loop on ONE GPU by the same clock and the oracle check.  Every rate counts only rays that
    went through a world query.  -> row dict on rank 0 (None elsewhere, or when a phase failed)."""
    W, H = cfg["W"], cfg["H"]
    rows = {}
    par = abi.make_params(W, H, out_format=abi.OUT_F32, pcg_mode=mode, **cfg["kw"])
    loop = [None]
    tag = f"{cfg['name']} {PCG_NAMES[mode]} {'sparse' if gather_sparse else 'whole'} gather"
    counts = {}

    def warm():
        loop[0] = ShardedFrameLoop(ds.scenes, cam, par, row_block=8, sparse=gather_sparse)
        ds.set_count_rays(True)
        ds.set_timing(True)
        for i in range(max(2, args.warmup)):
            loop[0].step(i, gather=True)
        loop[0].finish()
        fence(dist)
        ds.sync()
        st = ds.stats()
        r = torch.tensor([int(st.n_rays), int(st.n_rays_resolved)], dtype=torch.int64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        counts["rays"], counts["resolved"] = int(r[0].item()), int(r[1].item())

    def close():
        if loop[0] is not None:
            if agree.watchdog is not None:  # (draining a loop whose gather failed on one rank may wait for that rank)
                agree.watchdog.arm(f"closing the loop of {tag}", 60.0)
            try:
                loop[0].close()
            except Exception:  # noqa: BLE001
                pass
            if agree.watchdog is not None:
                agree.watchdog.disarm()

    if not agree.run(f"warm-up {tag}", warm):
        close()
        return None
    for gather in ((True, False) if full else (True,)):
        def timed():
            inject_failure(f"timed {'sparse' if gather_sparse else 'whole'}")
            elapsed, _, _ = timed_loop(ds, loop[0], steps, dist, gather, events=False)
            _, kernel_ms, launches = timed_loop(ds, loop[0], max(2, steps // 2), dist, gather, events=True)
            t = torch.tensor([elapsed, kernel_ms / max(1, launches)], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rows[("t", gather)] = (float(t[0].item()), float(t[1].item()))

        if not agree.run(f"timed loop {tag} gather={gather}", timed):
            close()
            return None

    def phases():  # where a frame's time goes on every rank, phases one after the other (a measurement mode)
        ds.set_count_rays(False)
        ds.set_timing(False)
        mine = loop[0].phase_probe(frames=4)
        ds.set_timing(True)
        rows["phases"] = rank_spread(dist, mine, world_size)

    if full and not agree.run(f"phase probe {tag}", phases):
        close()
        return None

    def check_and_solo():
        # the frame assembled on rank 0 must be bit-identical to the same frame rendered by one rank ...
        loop[0].step(0, gather=True)
        loop[0].finish()
        fence(dist)
        if rank == 0:
            rows["gather_bytes"] = loop[0].gather_bytes
            solo = ShardedFrameLoop(ds.scenes[:1], cam, par, row_block=8, solo=True)
            if full:
                # ... and the SAME workload on ONE GPU, by the same wall clock as `value` (rank 0 alone, the others wait)
                el1, _, _ = timed_loop(ds, solo, steps, None, False, events=False)
                if len(ds.scenes) > 1:  # ... with as many frames in flight as the ranks have, if that is faster on one GPU
                    many = ShardedFrameLoop(ds.scenes, cam, par, row_block=8, solo=True)
                    el_many, _, _ = timed_loop(ds, many, steps, None, False, events=False)
                    rows["n1_in_flight"] = (el1, el_many)
                    el1 = min(el1, el_many)
                _, k1, n1 = timed_loop(ds, solo, max(2, steps // 2), None, False, events=True)
                rows["n1"] = (el1, k1 / max(1, n1))
            else:
                solo.step(0, gather=False)
                solo.finish()
            rows["check"] = "ok" if torch.equal(solo.image(), loop[0].image()) else "MISMATCH"
            rows["image"] = loop[0].image()
        fence(dist)
        if rank == 0 and rows["check"] != "ok":  # (raised behind the fence: every rank has left its collectives)
            raise RuntimeError(f"{tag}: the gathered frame differs from the frame rank 0 renders alone")

    if not agree.run(f"gather check{' + one-GPU loop' if full else ''} {tag}", check_and_solo):
        close()
        return None

    def oracle():
        # ... and against the ORACLE's frame: the CPU restatement of the reference path on this rank's host cores, while the
        # GPUs idle -- bounded: skipped (and said so) when it would take the host longer than PT_BENCH_ORACLE_S seconds
        if rank == 0:
            rows["oracle"] = oracle_check(cfg["flat"], cam, par, rows["image"], counts["rays"])
            oc = rows["oracle"]
            if oc.get("checked") and (oc["pixels_beyond_1e-5"] > 1 or not oc["rays_match"]):
                raise RuntimeError(f"{tag}: the gathered frame differs from the ORACLE's in {oc['pixels_beyond_1e-5']} pixels beyond 1e-5 "
                                   f"(rays match: {oc['rays_match']})")

    ok = True
    if full:
        ok = agree.run(f"oracle check {tag} (the CPU oracle renders the whole frame on rank 0's host cores; the other ranks wait)", oracle,
                       deadline_s=ORACLE_LIMIT_S * 1.5 + 60)
    close()
    rows.pop("image", None)
    if not ok or rank != 0:
        return None

    traced = counts["rays"] - counts["resolved"]

    def line(gather):
        el, k = rows[("t", gather)]
        return {"value": traced * steps / el / 1e6, "unit": "Mray/s", "ms_per_step": el / steps * 1e3,
                "avg_render_kernels_ms_max_over_ranks": k}

    out = dict(line(True), gather_check=rows.get("check"), oracle_check=rows.get("oracle"),
               gather_bytes_per_frame_sent=rows.get("gather_bytes"), rays_traced_per_frame=traced,
               rays_settled_without_a_query_per_frame=counts["resolved"], steps=steps)
    if ("t", False) in rows:
        out["without_gather"] = line(False)
    if "phases" in rows:
        ph = rows["phases"]
        out["phases_ms"] = dict(ph, note="per rank, mean of 4 frames, phases run ONE AFTER THE OTHER with a device "
                                         "synchronisation between them (pytracer_amd/dist.py: phase_probe) and a barrier behind "
                                         "the render: render = the rank's rows; encode = the sparse encode incl. its count "
                                         "read-back (remote ranks); transfer = sends (remote) / until every shard is in (rank 0: "
                                         "the slowest remote encode + the wire); decode = placement + one-launch decode (rank 0). "
                                         "The frame loops overlap these; min / median / max are over the ranks")
        r = ph["render_ms"]["per_rank"]
        out["rank_share_imbalance"] = max(r) / max(1e-12, sum(r) / len(r))
    if "n1" in rows:
        el1, k1 = rows["n1"]
        n1 = {"value": traced * steps / el1 / 1e6, "unit": "Mray/s", "ms_per_step": el1 / steps * 1e3,
              "avg_render_kernels_ms": k1,
              "note": "the same frame loop on rank 0 alone (no partition, no gather), same wall clock as `value`; the "
                      "faster of one frame after the other and as many frames in flight as the ranks have"}
        if "n1_in_flight" in rows:
            a1, am = rows["n1_in_flight"]
            n1["ms_per_step_one_after_the_other"] = a1 / steps * 1e3
            n1["ms_per_step_frames_in_flight"] = am / steps * 1e3
        out["n1_same_workload"] = n1
        out["speedup"] = out["value"] / n1["value"]
        out["parallel_efficiency"] = out["speedup"] / world_size
    return out


ORACLE_LIMIT_S = float(os.environ.get("PT_BENCH_ORACLE_S", "150"))


def inject_failure(where):
    """PT_BENCH_FAIL=<where>:raise|hang -- the rehearsal of the fallback paths (profiles/r06_bench_gloo2_*): `raise` throws on
    rank 1 inside that phase, `hang` parks rank 1 there until the watchdog ends the job."""
    spec = os.environ.get("PT_BENCH_FAIL", "")
    if not spec or int(os.environ.get("RANK", "0")) != 1:
        return
    target, _, how = spec.partition(":")
    if target != where:
        return
    if how == "hang":
        time.sleep(10 * PHASE_DEADLINE_S + 3600)
    raise RuntimeError(f"injected failure in '{where}' (PT_BENCH_FAIL)")


def estimate_wall_s(world_size, cores):
    """Rough wall time of the N > 1 run, printed up front: the GPU phases are seconds; rank 0's oracle check of the 4K frame is the
    long part and is bounded by PT_BENCH_ORACLE_S (skipped, and said so, beyond it)."""
    c4_tests = 5.33e8 * 257  # rays of the C4 frame (every ray: the oracle traces them all) x shapes
    oracle_s = c4_tests / (ORACLE_TESTS_PER_CORE_S * max(1, cores))
    oracle_s = oracle_s if oracle_s <= ORACLE_LIMIT_S else 0.0
    gpu_s = 45.0 + 3.0 * world_size  # start-up, RCCL communicators, the loops (measured under gloo at 2 ranks: ~25 s without the oracle)
    return gpu_s + oracle_s + 3.0, oracle_s


def run_multi(args, rank, local_rank, world_size, dist, backend):
    from pytracer_amd import dist as ptdist

    W, H = C4["W"], C4["H"]
    flat = flatten.flatten_world(scenes.synthetic_world(C4["n_spheres"], wide=C4["wide"]))
    cam = cam_for(W, H)
    est, est_oracle = estimate_wall_s(world_size, usable_cores()[0])
    progress(f"estimated wall time of this run: about {est:.0f} s ({est_oracle:.0f} s of it rank 0's oracle check on {usable_cores()[0]} host cores; "
             f"per-phase deadline {PHASE_DEADLINE_S:.0f} s)")
    # frames in flight per rank: a rank's share of the frame is short and latency-bound (0.11 ms for an eighth of C4 against
    # 0.55 ms for the whole), and only another frame fills what it leaves idle (tools/share_in_flight.py: 0.108 -> 0.077 ms
    # per frame with two).  The one-GPU reference loop (n1_same_workload) runs with the same number in flight.
    n_in_flight = max(1, int(os.environ.get("PT_FRAMES_IN_FLIGHT", "2")))
    ds = SceneGroup(flat, n_in_flight, local_rank)
    dog = Watchdog(rank)
    dog.error_stub.update(n_gpus=world_size, steps=args.steps, warmup=args.warmup, backend=backend)
    agree = Agreement(dist, dog)
    seen = [0]

    def count_ranks():
        t = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        seen[0] = int(t.item())
        if seen[0] != world_size:
            raise RuntimeError(f"{seen[0]} ranks answered, {world_size} expected")

    agree.run("rank count", count_ranks)

    plan = ptdist.gather_plan(H, W, 8, world_size, itemsize=4, transport=ptdist.P2P)
    HEAD, SIDE = abi.PCG_SAMPLE, abi.PCG_PIXEL
    result = {
        "metric": "Mray/s (primary+shadow; rays handed to a world query), C4: ONE 3840x2160 frame, 256 spheres, PathTracer depth 5, "
                  f"64 spp, strong-scaled over {world_size} MI355X, RCCL gather of the HdrImage inside the timed region",
        "value": None, "unit": "Mray/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "ranks_seen": seen[0], "backend": backend, "code_hash": loaded_code_hash(),
        "config": {"workload": "C4 path tracer 3840x2160, 256 spheres, N=1, D=5, rr=3, S=8 (64 spp), fp32 RGB assembled on rank 0",
                   "width": W, "height": H, "n_shapes": flat.n_shapes, "renderer": "PathTracer",
                   "pcg_mode": "PT_PCG_SAMPLE", "frames_in_flight_per_rank": n_in_flight,
                   "partition": f"interleaved 8-row blocks over {world_size} ranks; a frame's shards reach rank 0 in batched RCCL send/recv groups "
                                "(whole: ONE transfer per remote rank; sparse: a fixed-size part and the runs that are not one colour), "
                                "then a strided placement copy per rank; the gather of a frame runs behind the next frame's render"},
        "workload_note": "NOT the N=1 line's workload (`python bench.py` measures C2, 1280x720 Flat, 15 us per frame: nothing to shard); the "
                         "one-GPU figure for THIS workload is n1_same_workload (rank 0 alone, same clock), speedup is against it; the "
                         "1280x720 figure on N ranks is at_1280x720; pcg_mode: one generator per sample ('PCG random state per-thread')",
        "gather": dict(plan, used=None, probe_ms_per_frame={}, fallback_reason=None,
                       note="the whole-shard gather (one transfer per remote rank, gather_bytes_per_frame) is measured FIRST, complete "
                            "with its checks; then the sparse one (pytracer_amd/dist.py: runs of 128 pixels that are one colour to the bit "
                            "travel as one pixel, lossless; two messages per remote rank); the headline is the faster of the two, and the "
                            "whole-shard row if the sparse one failed or overran its deadline (fallback_reason)"),
        "value_note": "every rate counts only rays that went through a world query (SURVEY.md 8(d)); the rays of sky tiles that the dome "
                      "shortcut settles without generating them (exact: DESIGN.md 4 items 6/8) are rays_settled_without_a_query_per_frame",
    }

    # ---- 1. the whole-shard gather: the plainest use of RCCL, measured first and completely (this is the fallback row) ----
    cfg4 = dict(C4, name="C4", flat=flat)
    whole = sharded_workload(args, cfg4, HEAD, ds, cam, rank, world_size, dist, agree, False, args.steps, full=True)
    if agree.error is None:
        dog.have_fallback = True
        if rank == 0:
            result.update(whole)
            result["gather"].update(used="whole", fallback_reason="sparse not measured yet")
            result["gather"]["probe_ms_per_frame"]["whole"] = whole["ms_per_step"]
            result["whole_gather"] = _pick(whole, ("value", "ms_per_step", "gather_bytes_per_frame_sent"))
            dog.fallback = (dict(result), None)
            write_detail(dict(result, partial="whole-shard gather measured; sparse gather and side rows follow"))

    # ---- 2. the sparse gather of the same frame: used if it completes everywhere, assembles the same frame and is faster ----
    if agree.error is None and ptdist.sparse_default():
        inner_error = [None]
        saved = agree.error
        sp = sharded_workload(args, cfg4, HEAD, ds, cam, rank, world_size, dist, agree, True, args.steps, full=False)
        if agree.error is not None:  # (a failure of the sparse path is soft: the whole-shard row stands)
            inner_error[0], agree.error = agree.error, saved
            agree.soft["sparse gather"] = inner_error[0]
        if rank == 0:
            sparse_ms = sp["ms_per_step"] if sp else None
            used, why = choose_gather(whole["ms_per_step"], sparse_ms, inner_error[0])
            result["gather"].update(used=used, fallback_reason=why)
            if sparse_ms is not None:
                result["gather"]["probe_ms_per_frame"]["sparse"] = sparse_ms
                result["sparse_gather"] = _pick(sp, ("value", "ms_per_step", "gather_bytes_per_frame_sent", "gather_check"))
            if used == "sparse":  # the same frame (checked against rank 0 alone, which the oracle checked): its loop is the headline
                for k in ("value", "ms_per_step", "avg_render_kernels_ms_max_over_ranks", "gather_bytes_per_frame_sent", "gather_check"):
                    result[k] = sp[k]
                n1 = result.get("n1_same_workload")
                if n1:
                    result["speedup"] = result["value"] / n1["value"]
                    result["parallel_efficiency"] = result["speedup"] / world_size
            dog.fallback = (dict(result), None)
            write_detail(dict(result, partial="C4 measured with both gathers; side rows follow"))
    elif agree.error is None and rank == 0:
        result["gather"].update(used="whole", fallback_reason="PT_GATHER_SPARSE=0")
    use_sparse = [bool(result["gather"].get("used") == "sparse")]
    if agree.error is None:  # every rank uses what rank 0 chose
        def share_choice():
            t = torch.tensor([1 if use_sparse[0] else 0], dtype=torch.int32, device="cuda")
            dist.broadcast(t, src=0)
            use_sparse[0] = bool(int(t.item()))

        agree.run("gather choice", share_choice)

    # ---- 3. side rows, each soft: the same frame under PT_PCG_PIXEL, C3 at 1280x720, C2 replicas ----
    def soft_rows(name, cfg, mode, dsx, camx, steps):
        if agree.error is not None:
            return None
        saved = agree.error
        row = sharded_workload(args, cfg, mode, dsx, camx, rank, world_size, dist, agree, use_sparse[0], steps, full=(name != "pcg_pixel"))
        if agree.error is not None:
            agree.soft[name], agree.error = agree.error, saved
            return {"error": agree.soft[name]} if rank == 0 else None
        return row

    pix = soft_rows("pcg_pixel", cfg4, SIDE, ds, cam, args.steps)
    if rank == 0 and pix is not None:
        result["pcg_pixel"] = dict(pix, note="the same frame with one generator per PIXEL (SURVEY.md 8c Mode PIXEL): a pixel's 64 samples "
                                             "consume ONE stream in order; checked against rank 0 alone (tests/test_gpu_fullsize.py checks "
                                             "this alignment against the oracle on the whole frame)")
    ds.close()

    # BASELINE.json's metric is quoted "at 1280x720, 1/2/4/8 MI355X": C3 -- 1280x720, 32 spheres, PathTracer D = 3, spp 16, per-thread
    # PCG -- through the same sharded loop.  A frame of 0.12 ms cut in N: the launch and the gather are most of what is left.
    flat3 = flatten.flatten_world(scenes.synthetic_world(C3["n_spheres"], wide=C3["wide"]))
    ds3 = SceneGroup(flat3, n_in_flight, local_rank)
    c3 = soft_rows("at_1280x720", dict(C3, name="C3", flat=flat3), HEAD, ds3, cam_for(C3["W"], C3["H"]), 5 * args.steps)
    ds3.close()
    if rank == 0 and c3 is not None:
        result["at_1280x720"] = dict(c3, workload="C3 path tracer 1280x720, 32 spheres, N=1, D=3, rr=3, S=4 (16 spp), PT_PCG_SAMPLE, fp32 RGB "
                                                  f"assembled on rank 0: ONE frame strong-scaled over {world_size} ranks (interleaved 8-row "
                                                  "blocks), same ShardedFrameLoop, gather inside the timed region")

    # ... and the N = 1 line's own workload, C2 (1280x720 Flat: one launch of 15 us, nothing to shard), as N independent
    # replicas without any collective: the figure the N = 1 line's `value` scales to if every GPU renders its own frames
    replicas = {}

    def c2_replicas():
        flat2 = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
        ds2 = DeviceScene(flat2, device=local_rank)
        try:
            par2 = abi.make_params(1280, 720, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
            lp = ShardedFrameLoop(ds2, cam_for(1280, 720), par2, row_block=8, solo=True)
            ds2.set_dome_shortcut(False)  # (the N = 1 line's frames: every primary ray traced)
            ds2.set_count_rays(True)
            for i in range(max(2, args.warmup)):
                lp.step(i, gather=False)
            lp.finish()
            ds2.sync()
            rays2 = int(ds2.stats().n_rays)
            k2 = 10 * args.steps
            el, _, _ = timed_loop(ds2, lp, k2, dist, False, events=False)
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            replicas.update(value=world_size * rays2 * k2 / float(t.item()) / 1e6, unit="Mray/s", ms_per_step=float(t.item()) / k2 * 1e3,
                            steps=k2, note=f"{world_size} independent replicas of the N = 1 line's workload (C2, 1280x720 Flat, dome shortcut "
                                           "off, one frame after the other per GPU), no collective: frames of all ranks / the slowest rank's time")
        finally:
            ds2.close()

    if agree.error is None:
        agree.attempt("C2 replicas", c2_replicas)

    dog.disarm()
    if rank == 0:
        if replicas:
            result["c2_replicas"] = replicas
        elif "C2 replicas" in agree.soft:
            result["c2_replicas"] = {"error": agree.soft["C2 replicas"]}
        result["ranks_seen"] = seen[0]
        result["phase_seconds"] = [[p, round(t, 2)] for p, t in agree.log]
        if agree.soft:
            result["soft_failures"] = agree.soft
        if agree.error is not None:
            result["error"] = agree.error
        emit(result, compact_multi(result))
    return 0 if agree.error is None else 1


def visible_gpus():
    """GPUs this job can see, counted WITHOUT the HIP runtime (torch.cuda.device_count() may fall through to
    hipGetDeviceCount and bring the runtime up in the parent): the KFD topology's nodes with SIMDs, cut by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  -> int, or None when the topology cannot be read."""
    import glob

    n = 0
    try:
        for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(path):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except (OSError, ValueError, IndexError):
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves
    (`python -m torch.distributed.run`, one per GPU) as a CHILD process (never an exec), relay rank 0's JSON line and
    exit with the job's status.  This parent does not touch the GPU: the devices are counted from the KFD topology in
    sysfs; where that cannot be read the count is left to the ranks (a rank without a device reports it)."""
    import socket
    import subprocess

    n_dev = visible_gpus()
    if n_dev is None:
        n_dev = args.gpus  # unknown here: the child ranks find out
    env = dict(os.environ)
    if n_dev < args.gpus and env.get("PT_DIST_BACKEND", "nccl") == "nccl":
        print(json.dumps({"metric": "Mray/s", "value": None, "unit": "Mray/s", "n_gpus": args.gpus,
                          "error": f"--gpus {args.gpus} but {n_dev} GPU(s) visible (PT_DIST_BACKEND=gloo rehearses the "
                                   "N-rank path on fewer GPUs, ranks sharing a card)"}), flush=True)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got_line = False
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            got_line = True
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = proc.wait()
    if not got_line:
        print(json.dumps({"metric": "Mray/s", "value": None, "unit": "Mray/s", "n_gpus": args.gpus,
                          "error": f"the ranks exited with status {rc} without printing a result line"}), flush=True)
    return rc if rc != 0 or got_line else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="N=1: how often the K-step timed loop is repeated (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--pre-roll-ms", type=float, default=30.0, help="N=1: untimed launches before the first timed loop, in ms of wall time")
    ap.add_argument("--min-timed-ms", type=float, default=20.0, help="N=1: the K-step loop is repeated until this much has been timed")
    ap.add_argument("--no-in-flight", action="store_true", help="N=1: skip the rows with 2 and 4 frames in flight")
    ap.add_argument("--headline-only", action="store_true",
                    help="N=1: skip the side row with the dome shortcut on (PMC collection: every launch of the kernel is the headline's)")
    ap.add_argument("--in-flight-steps", type=int, default=100,
                    help="N=1: frames per in-flight row (few against the headline's launches: rocprofv3's average of the headline "
                         "kernel over the whole command stays that of launches run back to back)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_size and "WORLD_SIZE" in os.environ:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world_size} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.steps is None:
        args.steps = 200 if world_size == 1 else 20
    if args.warmup is None:
        args.warmup = 20 if world_size == 1 else 3
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # (PT_DIST_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the N > 1 path on a one-GPU box)
    backend = os.environ.get("PT_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world_size == 1:
        run_single(args, local_rank)
        return 0
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    timeout = datetime.timedelta(seconds=int(os.environ.get("PT_DIST_TIMEOUT_S", "900")))  # (rank 0's oracle check: the others wait)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world_size, timeout=timeout,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world_size, timeout=timeout)
    rc = run_multi(args, rank, local_rank, world_size, dist, backend)
    try:
        if rc == 0:
            dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
