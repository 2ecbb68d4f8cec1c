"""Which kernels a (scene, camera, parameters) triple gets -- asserted on the CPU.

``launch()`` (csrc/ptrace.hip) is plan + enqueue since round 5: the plan is a pure host function (csrc/pt_plan.h) of the
scene's facts, the camera, ``pt_params`` and the tuning table, exported as ``pt_debug_plan`` (include/ptrace_debug.h).  Round
4's 630-line launch() could only be observed on a GPU (``pt_stats.kernel``).  Here: every BASELINE.json
configuration, the kernel families' borders (shapes 3 / 4, 256 / 257, 1023 / 1024 spheres; the LDS budget of the frame
stacks; jitter; orthogonal cameras; partitions), and every variant the GPU suite forces through a switch.  No GPU, no oracle:
the plan touches no device.  The kernels' names are the templates' with their variant spelled out (pt_plan_kernel_name).
"""
import pytest

from pytracer_amd import abi, device, flatten, scenes
from pytracer_amd import hostmodel as hm

C3 = dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3, path_state=45, path_seq=54)
CLI = dict(C3, samples_per_side=1, num_of_rays=10)  # main.py:95-102: pathtracing, N = 10, D = 3, one sample per pixel
_worlds = {}


def world(n, plane=False, wide=False):
    key = (n, plane, wide)
    if key not in _worlds:
        _worlds[key] = flatten.flatten_world(scenes.synthetic_world(n, with_plane=plane, wide=wide))
    return _worlds[key]


def cam(w, h):
    return flatten.flatten_camera(scenes.synthetic_camera(w, h))


def ortho_cam(w, h):
    return flatten.flatten_camera(hm.OrthogonalCamera(w / h, hm.translation(hm.Vec(-1.0, 0.0, 1.5)) * hm.scaling(hm.Vec(1.0, 3.0, 1.7))))


def plan(flat, w, h, camera=None, n_cu=256, dome=True, **kw):
    return device.plan(flat, camera or cam(w, h), abi.make_params(w, h, **kw), n_cu=n_cu, dome_shortcut=dome)


@pytest.fixture()
def tuning():
    """Set switches for one test, restore them afterwards (the table is process-wide)."""
    saved = {}

    def set_(name, value):
        saved.setdefault(name, device.get_tuning(name))
        device.set_tuning(name, value)

    yield set_
    for name, value in saved.items():
        device.set_tuning(name, value)


# ---- BASELINE.json's configurations -----------------------------------------------------------------------------------------
def test_c2_is_one_launch_of_the_16x16_tile_kernel():
    p = plan(world(32, plane=True), 1280, 720, renderer=abi.RENDERER_FLAT)
    assert p.kernels == ["pt_tile4_kernel<FLAT, LDS>"] and p.kernel == abi.KERNEL_TILE4
    assert (p.grid4_x, p.grid4_y, p.grid, p.npx) == (40, 23, 920, 4)  # 80 x 45 tiles of 16 x 16, 2 x 2 per workgroup
    assert p.lds_main == 33 * (128 + 256) and p.tile4_lds == 1 and p.hoist == 1 and p.frame_stack is None
    q = plan(world(32, plane=True), 1280, 720, renderer=abi.RENDERER_ONOFF)
    assert q.kernels == ["pt_tile4_kernel<ONOFF, noLDS>"] and q.lds_main == 0


def test_c3_two_passes_frame_stack_and_scene_in_lds():
    p = plan(world(32), 1280, 720, **C3)
    assert p.kernels == ["pt_tile_kernel<PATHTRACER>", "pt_path_regions_kernel<LDS, SCENE, LEAN>"] and p.kernel == abi.KERNEL_PATH_REGIONS
    assert (p.grid_first, p.grid, p.wg_per_cu) == (2048, 512, 2) and p.frame_stack == "LDS" and p.frame_doubles == 6
    assert p.lds_first == 32 and p.workspace_bytes == 0 and p.nregions == 160 * 90
    # masks 32 B + frames 3 x 6 x 256 x 8 + 32 scale+translate records + the scene's 32 x 384 B at a 256-B boundary
    assert p.lds_main == 39168 + 32 * 384
    for mode, rounds in ((abi.PCG_PIXEL, 16), (abi.PCG_SAMPLE, -2)):
        assert plan(world(32), 1280, 720, **dict(C3, pcg_mode=mode)).min_rounds == rounds


def test_cli_defaults_enqueue_the_tree_kernel_and_the_one_queue_alternative():
    p = plan(world(32), 1280, 720, **CLI)
    assert p.kernels == ["pt_tile_kernel<PATHTRACER>", "pt_path_tree_kernel<LEAN, SCENE>", "pt_path_flagged_kernel<LEAN, LDS>"]
    assert p.kernel == abi.KERNEL_PATH_TREE and p.frame_doubles == 20 and p.frame_stack == "LDS" and p.alt_frame_stack == "LDS"
    # node records per WAVE and depth (not per lane) + the 32 scale+translate records, then the leaf rounds' jump table
    # (2 x 4 x 64 pairs of 64-bit words), then -- at a 256-byte boundary -- the shapes' records for shading
    assert p.lds_main == (32 + 3 * 20 * 4 * 8 + 32 * 64 + 8192 + 255) // 256 * 256 + 32 * 384
    assert p.lds_alt == 3 * 20 * 256 * 8 + 32 * 64 and p.grid_alt == 256 and p.workspace_bytes == 0  # 120 KB of frames: one workgroup per CU
    # C3's 7.8 k flagged pixels at 640x360: tree; its 29 k at 1280x720 and up: one queue -- which hands its
    # heavy pixels to the tree kernel behind it: a budget derived on the device from the flagged pixels (-1), 50 rays after
    # the pixel queue ran dry, or 16 lanes left in a wave
    assert 8_000 < p.q_min_flagged < 29_000
    assert p.alt_budget == -1
    assert all(device.get_tuning(k) == -1 for k in ("q_budget", "q_tail_budget", "q_few_lanes"))  # (the plan's defaults)
    deep = plan(world(32), 1280, 720, **dict(CLI, num_of_rays=3, max_depth=5))
    assert deep.alt_kernel == "pt_path_flagged_kernel<LEAN, SPLIT>" and deep.alt_frame_stack == "SPLIT" and deep.grid_alt == 512
    assert deep.workspace_bytes == 5 * 20 * 256 * 8 * 512


def test_c4_blocks_of_strips_and_the_ball_hierarchy():
    p = plan(world(256, wide=True), 3840, 2160, **dict(C3, samples_per_side=8, max_depth=5))
    assert p.kernels == ["pt_tile_kernel<PATHTRACER, BLOCKS>", "pt_path_regions_kernel<LDS, NOGRID>"]
    assert p.block_h == 4 and p.ball_levels == 1 and p.has_grid == 0 and p.frame_stack == "LDS" and p.grid == 512
    assert plan(world(256, wide=True), 3840, 2160, dome=False, **dict(C3, samples_per_side=8, max_depth=5)).first_kernel == "pt_tile_kernel<PATHTRACER>"
    share = plan(world(256, wide=True), 3840, 2160, **dict(C3, samples_per_side=8, max_depth=5, n_ranks=8, rank=3, row_block=8))
    assert share.rows == 272 and share.npix == 272 * 3840 and share.main_kernel == p.main_kernel


def test_c5_cell_lists_and_the_grid():
    p = plan(world(10000, wide=True), 1280, 720, renderer=abi.RENDERER_FLAT)
    assert p.kernels == ["pt_cell_kernel", "pt_tile_kernel<FLAT, HIER>"] and p.hier == 1 and p.has_grid == 1 and p.grid == 2048
    q = plan(world(10000, wide=True), 1280, 720, **dict(C3, samples_per_side=2))
    assert q.kernels == ["pt_cell_kernel", "pt_tile_kernel<PATHTRACER, HIER>", "pt_path_regions_kernel<LDS>"]


def test_c1_demo_scene():
    w, camera = scenes.demo_world(clock=150.0)
    flat = flatten.flatten_world(w)
    p = device.plan(flat, flatten.flatten_camera(camera), abi.make_params(160, 120, abi.RENDERER_ONOFF))
    assert p.kernels == ["pt_simple_kernel<ONOFF, HOIST>"]  # three shapes: below the tile kernels' four
    q = device.plan(flat, flatten.flatten_camera(camera), abi.make_params(1280, 960, **CLI))
    assert q.kernels == ["pt_tile_kernel<PATHTRACER>", "pt_path_tree_kernel<LEAN, SCENE>", "pt_path_flagged_kernel<LEAN, LDS>"]


# ---- the families' borders --------------------------------------------------------------------------------------------------
def test_renderers_cameras_and_jitter():
    w = world(32, plane=True)
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT, samples_per_side=2).kernels == ["pt_tile_kernel<FLAT>"]
    assert plan(w, 1280, 720, renderer=abi.RENDERER_POINTLIGHT).kernels == ["pt_tile_kernel<POINTLIGHT>"]
    assert plan(w, 1280, 720, camera=ortho_cam(1280, 720), renderer=abi.RENDERER_FLAT).kernels == ["pt_tile_kernel<FLAT, ORTHO>"]
    o = plan(world(32), 1280, 720, camera=ortho_cam(1280, 720), **C3)
    assert o.kernels == ["pt_tile_kernel<PATHTRACER, ORTHO>", "pt_path_regions_kernel<LDS, SCENE, LEAN>"] and o.hoist == 0 and o.ortho == 1
    # a rank's share keeps the 16x16 tiles only when its row blocks are multiples of 16
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT, n_ranks=2, rank=1, row_block=8).kernels == ["pt_tile_kernel<FLAT>"]
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT, n_ranks=2, rank=1, row_block=16).main_kernel.startswith("pt_tile4_kernel")
    # nothing to render: no kernel at all
    assert plan(w, 64, 8, renderer=abi.RENDERER_FLAT, n_ranks=4, rank=3, row_block=8).kernels == []


def test_world_sizes():
    flat3, flat4 = world(3), world(4)
    assert plan(flat3, 320, 180, renderer=abi.RENDERER_FLAT).kernels == ["pt_simple_kernel<FLAT, HOIST>"]
    assert plan(flat4, 320, 180, renderer=abi.RENDERER_FLAT).main_kernel.startswith("pt_tile4_kernel<FLAT")
    assert plan(world(256), 320, 180, renderer=abi.RENDERER_FLAT).main_kernel == "pt_tile4_kernel<FLAT, noLDS>"  # 96 KB of records: not staged
    assert plan(world(257), 320, 180, renderer=abi.RENDERER_FLAT).kernels == ["pt_cell_kernel", "pt_tile_kernel<FLAT, HIER>"]
    assert plan(world(127), 320, 180, **C3).ball_levels == 0 and plan(world(129), 320, 180, **C3).ball_levels == 1
    empty = flatten.flatten_world(hm.World())
    assert plan(empty, 64, 36, renderer=abi.RENDERER_FLAT).kernels == ["pt_simple_kernel<FLAT, noHOIST>"]
    assert plan(empty, 64, 36, **C3).kernels == ["pt_path_kernel"]
    assert plan(world(32), 64, 36, **dict(C3, max_depth=-1)).kernels == ["memset"]


def test_frame_stack_leaves_the_lds_when_it_no_longer_fits():
    w = world(32)
    assert plan(w, 640, 360, **dict(C3, max_depth=13)).frame_stack == "LDS"   # 13 x 6 x 256 x 8 = 156 KB
    deep = plan(w, 640, 360, **dict(C3, max_depth=14))
    assert deep.frame_stack == "HBM" and deep.main_kernel == "pt_path_regions_kernel<HBM>"
    assert deep.workspace_bytes == 14 * 6 * 8 * deep.grid * 256
    # the tree kernel's per-wave node stack: LDS while D x 20 x 4 x 8 fits half the budget, else back to regions
    assert plan(w, 640, 360, **dict(CLI, max_depth=100)).main_kernel == "pt_path_tree_kernel<LEAN>"  # (64 KB of node records: no room left for the scene)
    assert plan(w, 640, 360, **dict(CLI, max_depth=40)).main_kernel == "pt_path_tree_kernel<LEAN, SCENE>"
    assert plan(w, 640, 360, **dict(CLI, max_depth=200)).main_kernel.startswith("pt_path_regions_kernel")
    # frames beyond 9 M pixels (a unit per pixel: 16 B each) never take the tree kernel; 4K frames do since round 5 -- with the
    # one-queue kernel in front, which the device picks when the flagged pixels are many
    assert plan(w, 3840, 2160, **CLI).main_kernel.startswith("pt_path_tree_kernel") and plan(w, 3840, 2160, **CLI).alt_kernel != ""
    assert plan(w, 4096, 2304, **CLI).main_kernel.startswith("pt_path_regions_kernel")


def test_fewer_compute_units_fewer_workgroups():
    a, b = plan(world(32), 1280, 720, n_cu=256, **C3), plan(world(32), 1280, 720, n_cu=64, **C3)
    assert (a.grid, b.grid) == (512, 128) and (a.grid_first, b.grid_first) == (2048, 512)


# ---- the switches the GPU suite forces (tests/test_gpu_parity.py, profiles/r0*_gpu_tests_*_forced.log) ----------------------
def test_switches_select_the_documented_variants(tuning):
    w = world(32, plane=True)
    tuning("tile4", 0)
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT).kernels == ["pt_tile_kernel<FLAT>"]
    tuning("tile4", 1)
    tuning("PTRACE_TILE4_LDS", 0)  # (environment spelling accepted too)
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT).main_kernel == "pt_tile4_kernel<FLAT, noLDS>"
    tuning("tile4_lds", 1)
    tuning("cull", 0)
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT).kernels == ["pt_simple_kernel<FLAT, HOIST>"]
    assert plan(world(32), 1280, 720, **C3).kernels == ["pt_path_kernel"]  # (one lane per pixel, every shape, the frame stack in HBM)
    tuning("cull", 1)
    tuning("tree", 0)
    assert plan(world(32), 1280, 720, **CLI).kernels == ["pt_tile_kernel<PATHTRACER>", "pt_path_regions_kernel<LDS, NOGRID>"]  # (120 KB of frames: no room for the scene)
    tuning("tree", 1)
    tuning("qchoice", 0)
    assert plan(world(32), 1280, 720, **CLI).alt_kernel == "" and plan(world(32), 1280, 720, **CLI).q_min_flagged == -1
    tuning("qchoice", 2)
    assert plan(world(32), 1280, 720, **CLI).q_min_flagged == 0
    tuning("qchoice", 1)
    tuning("q_frames_home", 2)  # only the deepest slot in LDS (40 KB: two workgroups per CU), the others in HBM: measured slower at D <= 3
    two = plan(world(32), 1280, 720, **CLI)
    assert two.alt_kernel == "pt_path_flagged_kernel<LEAN, SPLIT>" and two.grid_alt == 512 and two.lds_alt == 20 * 256 * 8 + 32 * 64
    assert two.workspace_bytes == 3 * 20 * 256 * 8 * 512
    tuning("q_frames_home", 1)
    assert plan(world(32), 1280, 720, **dict(CLI, max_depth=5)).alt_kernel == "pt_path_flagged_kernel<LEAN, SPLIT>"  # (does not fit: split)
    for gone in ("q_lanes", "q_lds_frames", "tile4_npx", "tree_fuse", "tree_uniform_max"):  # variants deleted in round 6: no switch left
        with pytest.raises(Exception, match="unknown tuning switch"):
            device.set_tuning(gone, 0)
    tuning("q_frames_home", 0)  # (all of the stack in HBM is gone: not a choice any more, the plan decides)
    assert plan(world(32), 1280, 720, **CLI).alt_kernel == "pt_path_flagged_kernel<LEAN, LDS>"
    tuning("q_frames_home", -1)
    tuning("q_budget", 5)  # (the GPU suite's way of handing nearly every pixel over in the middle of its tree)
    assert plan(world(32), 1280, 720, **CLI).alt_budget == 5
    tuning("q_budget", 0)
    assert plan(world(32), 1280, 720, **CLI).alt_budget == 0
    tuning("q_budget", -1)
    tuning("q_frames_home", -1)
    tuning("lds_frames", 0)
    assert plan(world(32), 1280, 720, **C3).main_kernel == "pt_path_regions_kernel<HBM>"
    tuning("lds_frames", 1)
    tuning("hier_min", 16)
    assert plan(w, 1280, 720, renderer=abi.RENDERER_FLAT, samples_per_side=2).kernels == ["pt_cell_kernel", "pt_tile_kernel<FLAT, HIER>"]
    tuning("hier_min", 256)
    tuning("small_query", 0)
    assert plan(world(32), 1280, 720, **C3).main_kernel == "pt_path_regions_kernel<LDS, SCENE>"
    assert plan(world(32), 1280, 720, **CLI).main_kernel == "pt_path_tree_kernel"
    tuning("small_query", 1)
    tuning("tree_scene_lds", 0)
    tuning("tree_jump", 0)
    lean = plan(world(32), 1280, 720, **CLI)
    assert lean.main_kernel == "pt_path_tree_kernel<LEAN>" and lean.lds_main == 32 + 3 * 20 * 4 * 8 + 32 * 64
    with pytest.raises(Exception, match="unknown tuning switch"):
        device.set_tuning("no_such_switch", 1)


def test_the_crossover_estimate_is_clamped_to_its_fitted_range():
    """the tree / one-queue crossover was fitted on N = 2 ... 20, D <= 8, <= 300 shapes; outside that range the
    estimate takes the range's corner instead of extrapolating."""
    w = world(32)
    assert plan(w, 640, 360, **dict(CLI, num_of_rays=20)).q_min_flagged == plan(w, 640, 360, **dict(CLI, num_of_rays=200)).q_min_flagged
    assert plan(w, 640, 360, **dict(CLI, num_of_rays=2, max_depth=8)).q_min_flagged == plan(w, 640, 360, **dict(CLI, num_of_rays=2, max_depth=30)).q_min_flagged
