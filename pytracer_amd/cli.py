"""Command-line driver with the flags of pytracer's ``render`` command.

    python -m pytracer_amd render [options] SCENE

Mirrors ``python -m pytracer render`` (main.py:76-214): same options, same defaults, same outputs
(a PFM file and a tone-mapped PNG), with the per-pixel loop running on the MI355X.  The scene-file
language is parsed by pytracer's own parser when pytracer is importable (the parser is out of scope
for this path and stays the reference's); ``SCENE`` may also name a built-in scene —
``builtin:demo`` (the scene of examples/demo.txt), ``builtin:c2``, ``builtin:c3``, ``builtin:c5`` —
so the driver also runs where pytracer is not installed.
"""
from __future__ import annotations

import sys
from math import sqrt
from time import perf_counter
from typing import Dict, List

import click

from . import hostmodel as hm
from . import scenes
from .tracer import GpuImageTracer

RENDERERS = ["onoff", "flat", "pathtracing", "pointlight"]


def build_variable_table(definitions: List[str]) -> Dict[str, float]:
    """``-d NAME:VALUE`` switches -> {name: value} (main.py:48-70)."""
    variables = {}
    for declaration in definitions:
        parts = declaration.split(":")
        if len(parts) != 2:
            print(f"error, the definition «{declaration}» does not follow the pattern NAME:VALUE")
            sys.exit(1)
        name, value = parts
        try:
            variables[name] = float(value)
        except ValueError:
            print(f"invalid floating-point value «{value}» in definition «{declaration}»")
            sys.exit(1)
    return variables


def _load_scene(name: str, variables: Dict[str, float], width: int, height: int):
    """-> (world, camera, classes): ``classes`` supplies the renderer / PCG types matching the world."""
    if name.startswith("builtin:"):
        which = name.split(":", 1)[1]
        if which == "demo":
            world, camera = scenes.demo_world(clock=variables.get("clock", 150.0))
        elif which in ("c2", "c3", "c5"):
            n, plane, wide = {"c2": (32, True, False), "c3": (32, False, False), "c5": (10000, False, True)}[which]
            world, camera = scenes.synthetic_world(n, with_plane=plane, wide=wide), scenes.synthetic_camera(width, height)
        else:
            print(f"unknown built-in scene «{which}» (demo, c2, c3, c5)")
            sys.exit(1)
        return world, camera, hm
    try:
        from pytracer import render as ref_render  # the reference's renderer classes (parameter holders here)
        from pytracer.pcg import PCG as RefPCG
        from pytracer.scene_file import GrammarError, InputStream, parse_scene
    except ImportError:
        print("pytracer is not importable: its scene-file parser is needed to read scene files "
              "(or use builtin:demo, builtin:c2, builtin:c3, builtin:c5)")
        sys.exit(1)
    with open(name, "rt") as f:
        try:
            scene = parse_scene(input_file=InputStream(stream=f, file_name=name), variables=variables)
        except GrammarError as e:
            loc = e.location
            print(f"{loc.file_name}:{loc.line_num}:{loc.col_num}: {e.message}")
            sys.exit(1)

    class Ref:  # the reference's own classes, so the flattener sees exactly what main.py would build
        OnOffRenderer, FlatRenderer = ref_render.OnOffRenderer, ref_render.FlatRenderer
        PathTracer, PointLightRenderer = ref_render.PathTracer, ref_render.PointLightRenderer
        PCG = RefPCG

    return scene.world, scene.camera, Ref


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
@click.argument("input_scene_name", type=str)
def render(width, height, algorithm, pfm_output, png_output, num_of_rays, max_depth, init_state, init_seq,
           samples_per_pixel, declare_float, device, input_scene_name):
    samples_per_side = int(sqrt(samples_per_pixel))
    if samples_per_side ** 2 != samples_per_pixel:
        print(f"Error, the number of samples per pixel ({samples_per_pixel}) must be a perfect square")
        return
    variables = build_variable_table(list(declare_float))
    world, camera, K = _load_scene(input_scene_name, variables, width, height)

    image = hm.HdrImage(width, height)
    print(f"Generating a {width}×{height} image")
    tracer = GpuImageTracer(image=image, camera=camera, samples_per_side=samples_per_side, device=device)
    if algorithm == "onoff":
        print("Using on/off renderer")
        renderer = K.OnOffRenderer(world=world)
    elif algorithm == "flat":
        print("Using flat renderer")
        renderer = K.FlatRenderer(world=world)
    elif algorithm == "pathtracing":
        print("Using a path tracer")
        renderer = K.PathTracer(world=world, pcg=K.PCG(init_state=init_state, init_seq=init_seq),
                                num_of_rays=num_of_rays, max_depth=max_depth)
    else:
        print("Using a point-light tracer")
        renderer = K.PointLightRenderer(world=world)

    def print_progress(row, col):
        print(f"Rendering row {row + 1}/{image.height}\r", end="")

    start = perf_counter()
    tracer.fire_all_rays(renderer, callback=print_progress)
    elapsed = perf_counter() - start
    st = tracer.last_stats
    print(f"Rendering completed in {elapsed:.3f} s (kernel {st.kernel_ms:.3f} ms, {st.n_rays} rays)")

    with open(pfm_output, "wb") as outf:  # main.py:203-204
        image.write_pfm(outf)
    print(f"HDR demo image written to {pfm_output}")
    image.normalize_image(factor=1.0)  # main.py:208-209
    image.clamp_image()
    with open(png_output, "wb") as outf:  # main.py:212-213
        image.write_ldr_image(outf, "PNG")
    print(f"PNG demo image written to {png_output}")
    tracer.close()


cli.add_command(render)
