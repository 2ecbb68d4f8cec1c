"""Command-line driver with the flags of pytracer's ``render`` command.

    python -m pytracer_amd render [options] SCENE

Mirrors ``python -m pytracer render`` (main.py:76-214): same options, same defaults, same outputs
(a PFM file and a tone-mapped PNG), with the per-pixel loop running on the MI355X.  The scene-file
language is parsed by pytracer's own parser when pytracer is importable (the parser is out of scope
for this path and stays the reference's); ``SCENE`` may also name a built-in scene —
``builtin:demo`` (the scene of examples/demo.txt), ``builtin:c2`` ... ``builtin:c5`` —
so the driver also runs where pytracer is not installed.
"""
from __future__ import annotations

import sys
from math import isqrt
from time import perf_counter
from typing import Dict, Iterable

import click

from . import hostmodel as hm
from . import scenes
from .tracer import GpuImageTracer

RENDERERS = ["onoff", "flat", "pathtracing", "pointlight"]


class UsageError(Exception):
    """A command-line value the driver cannot use; reported on stderr, exit status 2."""


def parse_float_overrides(switches: Iterable[str]) -> Dict[str, float]:
    """Every ``-d NAME:VALUE`` switch becomes one entry of the scene parser's variable table (the contract of
    main.py:48-70 and scene_file.py:640-675: float variables that override the scene file's own ``float`` lines).
    A switch without exactly one colon, with an empty name, or whose value is not a float is refused."""
    table: Dict[str, float] = {}
    for switch in switches:
        name, colon, text = switch.partition(":")
        if not colon or not name or ":" in text:
            raise UsageError(f"-d expects NAME:VALUE, got {switch!r}")
        try:
            table[name] = float(text)
        except ValueError:
            raise UsageError(f"-d {name}: {text!r} is not a floating-point number") from None
    return table


BUILTIN_SCENES = {  # name -> (spheres, ground plane, "wide" recipe) of SURVEY.md 8(d)
    "c2": (32, True, False), "c3": (32, False, False), "c4": (256, False, True), "c5": (10000, False, True)}


def _load_scene(name: str, variables: Dict[str, float], width: int, height: int):
    """-> (world, camera, classes): ``classes`` supplies the renderer / PCG types matching the world.

    ``builtin:NAME`` builds one of this repository's scene recipes from parameter-holder classes; anything
    else is a scene file in pytracer's language and goes through pytracer's own parser (out of scope for this
    path: it stays the reference's, scene_file.py:640-697), whose objects the flattener reads directly."""
    if name.startswith("builtin:"):
        which = name[len("builtin:"):]
        if which == "demo":
            return (*scenes.demo_world(clock=variables.get("clock", 150.0)), hm)
        if which in BUILTIN_SCENES:
            n, plane, wide = BUILTIN_SCENES[which]
            return scenes.synthetic_world(n, with_plane=plane, wide=wide), scenes.synthetic_camera(width, height), hm
        raise UsageError(f"no built-in scene {which!r} (have: demo, {', '.join(sorted(BUILTIN_SCENES))})")
    try:
        from pytracer import render as ref_render
        from pytracer.pcg import PCG as RefPCG
        from pytracer.scene_file import GrammarError, InputStream, parse_scene
    except ImportError:
        raise UsageError(f"{name}: reading a scene file needs pytracer's parser, and pytracer is not importable "
                         "here; the built-in scenes (builtin:demo, builtin:c2, ...) need nothing") from None
    try:
        with open(name, "rt") as f:
            scene = parse_scene(input_file=InputStream(stream=f, file_name=name), variables=variables)
    except OSError as e:
        raise UsageError(f"{name}: {e.strerror}") from None
    except GrammarError as e:
        at = e.location
        raise UsageError(f"{at.file_name}, line {at.line_num}, column {at.col_num}: {e.message}") from None

    class Ref:  # the reference's own classes, so the flattener sees exactly what main.py would build
        OnOffRenderer, FlatRenderer = ref_render.OnOffRenderer, ref_render.FlatRenderer
        PathTracer, PointLightRenderer = ref_render.PathTracer, ref_render.PointLightRenderer
        PCG = RefPCG

    return scene.world, scene.camera, Ref


class RenderJob:
    """Everything `render` needs before a GPU is touched: scene objects and the renderer parameter holder."""

    def __init__(self, world, camera, renderer, samples_per_side):
        self.world, self.camera, self.renderer, self.samples_per_side = world, camera, renderer, samples_per_side


def plan_render(width, height, algorithm, num_of_rays, max_depth, init_state, init_seq, samples_per_pixel,
                declare_float, input_scene_name) -> RenderJob:
    """Options -> scene + renderer (main.py:144-191 in effect): the sample count must be a perfect square; the
    four algorithms map onto the four renderer classes, only the path tracer takes the ray/depth/seed options."""
    side = isqrt(samples_per_pixel) if samples_per_pixel >= 0 else -1
    if side * side != samples_per_pixel:
        raise UsageError(f"--samples-per-pixel {samples_per_pixel}: must be a perfect square (1, 4, 9, 16, ...)")
    world, camera, K = _load_scene(input_scene_name, parse_float_overrides(declare_float), width, height)
    make = {
        "onoff": lambda: K.OnOffRenderer(world=world),
        "flat": lambda: K.FlatRenderer(world=world),
        "pathtracing": lambda: K.PathTracer(world=world, pcg=K.PCG(init_state=init_state, init_seq=init_seq),
                                            num_of_rays=num_of_rays, max_depth=max_depth),
        "pointlight": lambda: K.PointLightRenderer(world=world),
    }
    return RenderJob(world, camera, make[algorithm](), side)


@click.group()
def cli():
    pass


@click.command("render")
@click.option("--width", type=int, default=640, help="Width of the image to render")
@click.option("--height", type=int, default=480, help="Height of the image to render")
@click.option("--algorithm", type=click.Choice(RENDERERS), default="pathtracing")
@click.option("--pfm-output", type=str, default="output.pfm", help="Name of the PFM file to create")
@click.option("--png-output", type=str, default="output.png", help="Name of the PNG file to create")
@click.option("--num-of-rays", type=int, default=10,
              help="Number of rays departing from each surface point (only with --algorithm=pathtracing).")
@click.option("--max-depth", type=int, default=3, help="Maximum allowed ray depth (only with --algorithm=pathtracing).")
@click.option("--init-state", type=int, default=45, help="Initial seed for the random number generator.")
@click.option("--init-seq", type=int, default=54, help="Identifier of the random sequence.")
@click.option("--samples-per-pixel", type=int, default=1, help="Samples per pixel (a perfect square, e.g. 16).")
@click.option("--declare-float", "-d", type=str, multiple=True, help="Declare a variable: --declare-float=VAR:VALUE")
@click.option("--device", type=int, default=0, help="GPU to render on")
@click.option("--pcg-mode", type=click.Choice(["auto", "seq", "pixel", "sample"]), default="auto",
              help="Random streams.  auto (default): onoff / flat / pointlight draw their jitter from the reference's own "
                   "sequential stream PCG(42, 54) -- the frame `python -m pytracer render` writes with the same flags, bit "
                   "for bit -- and pathtracing uses one generator per sample (its scattering stream is serial in the "
                   "reference and cannot be reproduced in parallel; per-sample and per-pixel generators are both pinned by "
                   "reference-held frames and coincide at the default --samples-per-pixel 1).  seq: the reference's stream "
                   "(refused for pathtracing).  pixel: one generator per pixel, consumed in program order (slower with "
                   "--samples-per-pixel > 1: a pixel's samples then depend on each other).  sample: one per sample.")
@click.option("--host-postprocess", is_flag=True, default=False,
              help="Copy the fp64 frame to the host first and post-process there (the round-2 behaviour; same bytes).")
@click.argument("input_scene_name", type=str)
def render(width, height, algorithm, pfm_output, png_output, num_of_rays, max_depth, init_state, init_seq,
           samples_per_pixel, declare_float, device, pcg_mode, host_postprocess, input_scene_name):
    try:
        if pcg_mode == "seq" and algorithm == "pathtracing":
            raise UsageError("--pcg-mode seq with --algorithm pathtracing: the reference's scattering stream is one generator "
                             "consumed in the order the paths of all pixels end in (render.py:118,128) -- serial by "
                             "construction; use sample (what auto, the default, picks for pathtracing) or pixel")
        job = plan_render(width, height, algorithm, num_of_rays, max_depth, init_state, init_seq, samples_per_pixel,
                          declare_float, input_scene_name)
    except UsageError as e:
        click.echo(f"pytracer_amd render: {e}", err=True)
        sys.exit(2)
    from . import _lib, prefer_device_kernargs

    prefer_device_kernargs()  # (this process is ours: kernel arguments in device memory, before its first HIP call)
    _lib.standalone()  # (... and it never imports torch: the frame's HBM and streams come from the C-ABI, pt_device_alloc)
    click.echo(f"{width}x{height} px, {algorithm}, {job.samples_per_side ** 2} sample(s) per pixel, "
               f"{len(job.world.shapes)} shape(s), GPU {device}")
    image = hm.HdrImage(width, height)
    tracer = GpuImageTracer(image=image, camera=job.camera, samples_per_side=job.samples_per_side, device=device,
                            pcg_mode=pcg_mode, resident=not host_postprocess)
    started = perf_counter()
    tracer.fire_all_rays(job.renderer, callback=lambda col, row: click.echo(f"  row {row + 1} of {height}\r", nl=False))
    wall = perf_counter() - started
    st = tracer.last_stats
    click.echo(f"frame done: {wall:.3f} s wall, {st.kernel_ms:.3f} ms in kernels, {st.n_rays} rays")
    # what main.py:203-213 leaves behind: the PFM of the raw frame, then a PNG of the tone-mapped one.  The frame stays
    # in HBM (fp64, like the reference's HdrImage): the PFM floats and the tone-mapped bytes are all that crosses the link
    frame = image if host_postprocess else tracer.device_image
    with open(pfm_output, "wb") as f:
        frame.write_pfm(f)
    frame.normalize_image(factor=1.0)
    frame.clamp_image()
    with open(png_output, "wb") as f:
        frame.write_ldr_image(f, "PNG")
    click.echo(f"wrote {pfm_output} and {png_output}")
    tracer.close()


cli.add_command(render)
