// pt_plan.h — what a frame WILL launch, decided by a pure host function (no HIP call, no device, no global but the tuning
// table): which kernels, their grids and dynamic LDS, where the path tracer's frame stack lives, every threshold.
//
//   pt_make_plan(facts, camera, params, tuning) -> PtPlan          ptrace.hip: launch() = plan + enqueue
//
// Round 4's launch() decided all of this inline, between the HIP calls, from ~45 getenv()s parsed into function-local
// statics: which kernel a (scene, params) pair got could only be observed on a GPU.  Now the decision
// is data: pt_debug_plan() (include/ptrace_debug.h) runs it for a scene DESCRIPTION on any machine, tests/test_plan.py pins
// it for every BASELINE.json configuration and for every edge the GPU tests force through a switch, and the switches live
// in ONE table (PtTuning: read from the environment once, settable through pt_debug_set_tuning -- also between scenes of
// one process, which the function-local statics could not).
//
// Nothing here changes a pixel: every kernel a plan can name renders the same image (tests/test_gpu_parity.py runs the
// forced variants against the oracle).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ptrace.h"
#include "../../include/ptrace_debug.h"

// ---- the switches ---------------------------------------------------------------------------------------------------------
// (name, environment variable, default).  Debug / measurement switches: none changes a result.
#define PT_TUNING_TABLE(X)                                                                                                  \
  X(cull, "PTRACE_CULL", 1)                       /* 0: no tile culling at all (pt_simple_kernel / pt_path_kernel) */       \
  X(levels_min, "PTRACE_LEVELS_MIN", 128)         /* spheres from which the 8- / 64-ball hierarchy is consulted */          \
  X(path_wg_per_cu, "PTRACE_PATH_WG_PER_CU", 0)   /* path tracer: workgroups per CU (0: 2 by regions, 3 one queue) */       \
  X(lds_frames, "PTRACE_LDS_FRAMES", 1)           /* 0: the frame stack always in HBM */                                    \
  X(tree, "PTRACE_TREE", 1)                       /* 0: num_of_rays > 1 never takes pt_path_tree_kernel */                  \
  X(tree_max_pixels, "PTRACE_TREE_MAX_PIXELS", 9000000) /* frames larger than this never take the tree kernel (16 B of unit list per pixel) */ \
  X(scene_lds, "PTRACE_SCENE_LDS", 1)             /* second pass: the shapes' records staged in LDS when they fit */        \
  X(tile_wg_per_cu, "PTRACE_TILE_WG_PER_CU", 0)   /* 8x8 tile kernels: cap on resident workgroups per CU (0: 8) */          \
  X(tile4, "PTRACE_TILE4", 1)                     /* 0: never pt_tile4_kernel */                                            \
  X(tile4_lds, "PTRACE_TILE4_LDS", 1)             /* pt_tile4_kernel<FLAT>: records staged in LDS for shading */            \
  X(qchoice, "PTRACE_QCHOICE", 1)                 /* 0: never the one-queue alternative, 2: always (measurement) */         \
  X(q_wg_per_cu, "PTRACE_Q_WG_PER_CU", 0)         /* the one-queue kernel: workgroups per CU (0: what its LDS frames allow) */  \
  X(q_frames_home, "PTRACE_Q_FRAMES_HOME", -1)    /* the one-queue kernel's frame stack: 1 LDS, 2 split (deepest slot in LDS); -1: by the plan */ \
  X(q_min_flagged, "PTRACE_Q_MIN_FLAGGED", -1)    /* >= 0: the flagged-pixel count from which the one-queue kernel works */ \
  X(q_budget, "PTRACE_Q_BUDGET", -1)              /* rays after which the one-queue kernel hands a pixel to the tree kernel (0: never; -1: by the plan) */ \
  X(q_tail_budget, "PTRACE_Q_TAIL_BUDGET", -1)    /* ... the same counted from the moment the pixel queue runs dry (0: never; -1: by the plan) */ \
  X(q_few_lanes, "PTRACE_Q_FEW_LANES", -1)        /* ... or when, the queue dry, a wave holds this many pixels or fewer (0: never; -1: by the plan) */ \
  X(p_maxpath, "PTRACE_P_MAXPATH", 0)             /* step batching of path_trace (0: by kernel) */                          \
  X(s_min, "PTRACE_S_MIN", 0)                                                                                               \
  X(unit_lanes_cap, "PTRACE_UNIT_LANES_CAP", -1)  /* lanes pt_unit_scatter may plan for (-1: the resident ones, 0: a unit = a region) */ \
  X(unit_min_rounds, "PTRACE_UNIT_MIN_ROUNDS", 0)                                                                           \
  X(spec_draws, "PTRACE_SPEC_DRAWS", -1)          /* PT_PCG_PIXEL: draws assumed per sample of an unknown pixel */          \
  X(tree_scene_lds, "PTRACE_TREE_SCENE_LDS", 1)   /* tree kernel (small worlds): the shapes' records staged in LDS */       \
  X(tree_jump, "PTRACE_TREE_JUMP", 1)             /* tree kernel: leaf rounds' state jumps from a table in LDS */           \
  X(trace_unit, "PTRACE_TRACE_UNIT", 0)                                                                                     \
  X(pixel_dome, "PTRACE_PIXEL_DOME", 1)           /* first pass: per-pixel dome classification */                           \
  X(hier_min, "PTRACE_HIER_MIN", 256)             /* shapes above which tiles cull cell lists (< 0: never) */               \
  X(block_h, "PTRACE_BLOCK_H", 0)                 /* first pass, BLOCKS variant: strips per block (0: by frame size) */     \
  X(small_query, "PTRACE_SMALL_QUERY", 1)         /* worlds without grid / ball hierarchy: the lean query variants */       \
  X(grid, "PTRACE_GRID", 1)                       /* upload: build the uniform grid */                                      \
  X(grid_min, "PTRACE_GRID_MIN", 1024)            /* ... from this many ordinary spheres */

struct PtTuning {
#define X(field, env, dflt) long long field = dflt;
  PT_TUNING_TABLE(X)
#undef X
  double grid_density = 4.0;  // PTRACE_GRID_DENSITY: cells per sphere
};

static inline PtTuning pt_tuning_from_env() {
  PtTuning t;
#define X(field, env, dflt) \
  if (const char *v = getenv(env)) t.field = atoll(v);
  PT_TUNING_TABLE(X)
#undef X
  if (const char *v = getenv("PTRACE_GRID_DENSITY")) t.grid_density = atof(v);
  return t;
}

// the process's table: the environment as it was at first use, then whatever pt_debug_set_tuning changed
static inline PtTuning &pt_tuning() {
  static PtTuning t = pt_tuning_from_env();
  return t;
}

static inline bool pt_tuning_set(PtTuning &t, const char *name, long long value) {
#define X(field, env, dflt)                                      \
  if (strcmp(name, #field) == 0 || strcmp(name, env) == 0) {     \
    t.field = value;                                             \
    return true;                                                 \
  }
  PT_TUNING_TABLE(X)
#undef X
  return false;
}

// ---- what the plan needs to know of a scene (pt_scene_upload's analysis; no pointer, no device) -----------------------------
struct PtSceneFacts {
  int n_shapes = 0, n_spheres = 0, n_diag = 0, n_lights = 0;
  int bs_levels = 0;       // >= 128 spheres: Morton order + ball hierarchy
  int has_grid = 0;        // a uniform grid was built (>= 1024 ordinary spheres)
  int grid_n_cells = 0;
  int n_cu = 256;
  int dome_shortcut = 1;   // pt_set_dome_shortcut
};

// LDS one workgroup may use (gfx950: 160 KiB per CU, all of it available to one workgroup), less the kernels' few static bytes
static const size_t PT_LDS_BUDGET = 160 * 1024 - 512;

enum PtTileMode { PT_TILE_PLAIN = 0, PT_TILE_HIER = 1, PT_TILE_ORTHO = 2, PT_TILE_BLOCKS = 3 };
enum PtSecondPass {  // the path tracer's second-pass kernel
  PT_SECOND_NONE = 0,
  PT_SECOND_REGIONS_LDS_SCENE_LEAN,   // pt_path_regions_kernel<true, true, 1>
  PT_SECOND_REGIONS_LDS_SCENE,        // <true, true, 0>
  PT_SECOND_REGIONS_LDS_NOGRID,       // <true, false, 2>
  PT_SECOND_REGIONS_LDS,              // <true, false, 0>
  PT_SECOND_REGIONS_HBM,              // <false, false, 0>
  PT_SECOND_TREE_LEAN,                // pt_path_tree_kernel<true>
  PT_SECOND_TREE,                     // pt_path_tree_kernel<false>
  PT_SECOND_TREE_LEAN_SCENE,          // pt_path_tree_kernel<true, true>: the shapes' records staged in LDS
};
#define PT_PLAN_HANDOVER_HEADER 8  // (pt_kernels.h: PT_HANDOVER_HEADER; asserted equal in ptrace.hip)
#ifndef PT_HANDOVER_CAP
#define PT_HANDOVER_CAP 65536      // records; a full table leaves a pixel with its lane
#endif
#ifndef PT_Q_BUDGET_DEFAULT
#define PT_Q_BUDGET_DEFAULT -1
#endif
#ifndef PT_Q_TAIL_BUDGET_DEFAULT
#define PT_Q_TAIL_BUDGET_DEFAULT 50
#endif
#ifndef PT_Q_FEW_LANES_DEFAULT
#define PT_Q_FEW_LANES_DEFAULT 16
#endif
enum PtAltPass {  // the one-queue alternative launched IN FRONT OF the tree kernel (PT_Q_CHOICE: which of the two the device lets work)
  PT_ALT_NONE = 0,
  PT_ALT_FLAGGED_LEAN_LDS,  // pt_path_flagged_kernel<1, 1>
  PT_ALT_FLAGGED_LDS,       // <0, 1>
  PT_ALT_FLAGGED_LEAN_SPLIT,  // pt_path_flagged_kernel<1, 2>: the deepest stack slot in LDS, the others in HBM
  PT_ALT_FLAGGED_SPLIT,       // <0, 2>
};

struct PtPlan {
  // the frame
  int rows = 0;
  long long npix = 0;
  int npass = 0;
  // which path
  int kernel = 0;            // PT_KERNEL_* of the render kernel proper (pt_stats.kernel before the device's choice)
  bool zero_frame = false;   // path tracer with max_depth < 0: a memset (render.py:100-101)
  bool ortho = false, hoist = false, tile = false, tile4 = false, path_tiled = false, hier = false, tree = false, q_alt = false;
  int tile_mode = PT_TILE_PLAIN;
  int second = PT_SECOND_NONE, alt = PT_ALT_NONE;
  bool simple_hoist = false;  // pt_simple_kernel<R, hoist>
  // tile4
  bool t4lds = false;
  unsigned grid4_x = 1, grid4_y = 1;
  // grids (workgroups of PT_BLOCK threads)
  int grid = 1;        // the render kernel proper (path tracer: the second pass)
  int grid_first = 0;  // path tracer: the first pass
  int grid_q = 0;      // ... the one-queue alternative
  int grid_scatter = 0;
  // cell kernel (hier)
  int cells_x = 0, ncells = 0, cell_stride = 0, cell_groups = 0, cell_chunks = 0, cell_chunk_len = 0;
  // dynamic LDS, bytes
  size_t lds_tile = 0;    // survivor masks of the 8x8 tile kernels / first pass
  size_t lds_main = 0;    // the render kernel proper
  size_t lds_q = 0;       // the one-queue alternative
  // path tracer
  bool lds_frames = false, q_lds_frames = false;
  int q_tail_budget = 0;  // ... or this many rays after the pixel queue ran dry
  int handover_cap = 0;   // records of pixels handed from the one-queue kernel to the tree kernel (PT_HANDOVER_HEADER + 20 doubles per node each)
  size_t handover_doubles = 0;
  int q_few_lanes = 0;    // ... or when a wave of the dry queue holds this many pixels or fewer
  int q_budget = 0;       // the one-queue alternative hands a pixel to the tree kernel once it has traced this many rays (0: never;
                          // -1: q_budget_per_flagged x the frame's flagged pixels, at least q_budget_min -- pt_unit_scatter)
  double q_budget_per_flagged = 0.0;
  int q_budget_min = 0;
  int q_home = 0;         // the one-queue alternative's frame stack: 1 LDS, 2 split (deepest slot in LDS, the rest in HBM)
                          // (all of it in HBM was never faster than the split stack and was deleted in round 6)
  int frame_doubles = 6;
  int diag_lds = -1, grid_occ_lds = -1, scene_lds = -1;  // offsets as the kernels take them (PtKArgs)
  int q_diag_lds = -1;
  int tree_jump_lds = -1;   // tree kernel: the leaf rounds' jump table (8-byte words), -1 = none
  int wg_per_cu = 0;
  int p_max_path = 0, s_min_path = 0, q_p_max_path = 48, q_s_min_path = 16;
  size_t ws_bytes = 0;    // frame stack in HBM (0: none needed)
  int nregions = 0, units_need = 0;
  long long lanes_cap = 0;
  int nsamp = 1, min_rounds = 0, spec_draws = 0;
  long long q_min = -1;   // flagged pixels from which the one-queue kernel takes the frame (-1: never)
  int block_h = 1;
  int bs_levels = 0;      // the ball hierarchy is consulted
  int nthreads = 0;
};

static inline int pt_plan_rows(const pt_params *p) {
  if (!p || p->height <= 0) return 0;
  const int rb = p->row_block > 0 ? p->row_block : 1;
  const int nr = p->n_ranks > 0 ? p->n_ranks : 1;
  int rows = 0;
  for (int b = 0; b * rb < p->height; ++b)
    if (b % nr == p->rank) rows += std::min(rb, p->height - b * rb);
  return rows;
}

// Sizes the kernels' records have (pt_layout.h; kept here as numbers so that the plan needs no device header): checked by
// static_asserts in ptrace.hip.
#define PT_PLAN_BLOCK 256
#define PT_PLAN_REGION 8
#define PT_PLAN_CELL 32
#define PT_PLAN_CELL_CHUNK 2048
#define PT_PLAN_TREE_FRAME 20
#define PT_PLAN_SCATTER_BLOCK 256
#define PT_PLAN_DIAG_BYTES 64
#define PT_PLAN_REC_BYTES 128
#define PT_PLAN_AUX_BYTES 256

static inline void pt_make_plan(const PtSceneFacts &s, const pt_camera *cam, const pt_params *p, const PtTuning &t, PtPlan &pl) {
  pl = PtPlan();
  const int B = PT_PLAN_BLOCK;
  const int row_block = p->row_block > 0 ? p->row_block : 1;
  const int n_ranks = p->n_ranks > 0 ? p->n_ranks : 1;
  const int rows = pt_plan_rows(p);
  pl.rows = rows;
  pl.npass = (s.n_shapes + 63) / 64;
  pl.npix = (long long)rows * p->width;
  pl.bs_levels = s.bs_levels && s.n_spheres >= t.levels_min;
  if (pl.npix == 0) return;

  // grid: one lane per pixel up to the resident capacity of the chip, grid-stride beyond
  const long long want = (pl.npix + B - 1) / B;
  long long cap = (long long)s.n_cu * 8;  // 8 x 256-thread workgroups per CU = 32 waves/CU
  const bool pathtracer = p->renderer == PT_RENDERER_PATHTRACER;
  size_t frame_lds = 0, staged_bytes = 0;  // staged_bytes: everything staged behind the frames (diag records, grid bits, scene)
  const bool regions = pathtracer && s.n_shapes > 0 && t.cull != 0;
  size_t mask_lds = 0;
  if (pathtracer) {
    // The path tracer hands pixels out dynamically; fewer resident lanes than pixels lets a lane that drew a cheap pixel
    // take several more while its neighbours finish an expensive one.  One queue for all pixels: 152 VGPRs, 3 waves per
    // SIMD; second pass by regions: built for 1-2
    int wg_per_cu = t.path_wg_per_cu > 0 ? (int)t.path_wg_per_cu : (regions ? 2 : 3);
    // a frame is pushed for depths 0 .. max_depth-1 only (a hit at max_depth spawns nothing that is traced)
    pl.frame_doubles = p->num_of_rays > 1 ? 20 : 6;
    frame_lds = (size_t)std::max(p->max_depth, 1) * pl.frame_doubles * B * sizeof(double);
    mask_lds = regions ? (size_t)4 * pl.npass * sizeof(unsigned long long) : 0;
    // num_of_rays > 1: one pixel per wave, a node's children on lanes (pt_path_tree_kernel); its stack holds one record
    // per NODE and wave, not one per lane.  One pixel per wave pays while the flagged pixels are few per resident wave;
    // the frame size stands in for their number when the kernels are enqueued (the device then picks: q_alt below).
    const size_t tree_lds = (size_t)std::max(p->max_depth, 1) * PT_PLAN_TREE_FRAME * (B / 64) * sizeof(double);
    pl.tree = regions && p->num_of_rays > 1 && t.tree != 0 && t.lds_frames != 0 && pl.npix <= t.tree_max_pixels &&
              tree_lds + mask_lds <= PT_LDS_BUDGET / 2;
    if (pl.tree) {
      pl.frame_doubles = PT_PLAN_TREE_FRAME;
      frame_lds = tree_lds;
    }
    // (the one-lane-per-pixel pt_path_kernel -- empty worlds, PTRACE_CULL=0 -- keeps its stack in HBM: its LDS variant went in round 6)
    pl.lds_frames = regions && t.lds_frames != 0 && frame_lds + mask_lds <= PT_LDS_BUDGET;
    // the scale+translate records ride along in LDS when they fit (world_query_lanes gathers them per lane)
    const size_t base_lds = mask_lds + (pl.lds_frames ? frame_lds : 0);
    staged_bytes = (size_t)s.n_diag * PT_PLAN_DIAG_BYTES;
    if (regions && s.n_diag > 0 && base_lds + staged_bytes <= PT_LDS_BUDGET && staged_bytes <= 48 * 1024) {
      pl.diag_lds = (int)(base_lds / 8);
    } else {
      staged_bytes = 0;
    }
    // ... and so do the occupancy bits of the grid (one per cell: at most 8 KB)
    const size_t occ_bytes = (regions && s.has_grid) ? (((size_t)s.grid_n_cells + 31) / 32 * 4 + 7) / 8 * 8 : 0;
    if (occ_bytes && base_lds + staged_bytes + occ_bytes <= PT_LDS_BUDGET) {
      pl.grid_occ_lds = (int)((base_lds + staged_bytes) / 4);
      staged_bytes += occ_bytes;
    }
    // ... and the shapes' own records (what shading gathers per lane), while two workgroups still fit a CU
    if (regions && pl.lds_frames && t.scene_lds && !pl.tree) {
      const size_t at = (base_lds + staged_bytes + 255) / 256 * 256;
      const size_t scene_bytes = (size_t)s.n_shapes * (PT_PLAN_REC_BYTES + PT_PLAN_AUX_BYTES);
      if (at + scene_bytes <= PT_LDS_BUDGET / 2) {
        pl.scene_lds = (int)(at / 8);
        staged_bytes = at + scene_bytes - base_lds;
      }
    }
    if (pl.lds_frames || staged_bytes)
      wg_per_cu = std::min<int>(wg_per_cu, (int)(PT_LDS_BUDGET / std::max<size_t>(1, base_lds + staged_bytes)));
    pl.wg_per_cu = wg_per_cu;
    cap = (long long)s.n_cu * wg_per_cu;
  }
  pl.ortho = cam->kind != PT_CAMERA_PERSPECTIVE;
  // the tiled path tracer: primary rays use the hoisted, culled tile query
  pl.path_tiled = regions;
  // per-camera constants of the shapes (invm * origin): only a perspective camera has a common origin
  pl.hoist = !pl.ortho && s.n_shapes > 0 && (!pathtracer || pl.path_tiled);
  // 8x8 tiles with culled shape lists: primary rays (OnOff, Flat, PointLight)
  pl.tile = s.n_shapes > 0 && t.cull != 0 && s.n_shapes >= 4 &&
            (p->renderer == PT_RENDERER_ONOFF || p->renderer == PT_RENDERER_FLAT || p->renderer == PT_RENDERER_POINTLIGHT);
  int grid = (int)std::max<long long>(1, std::min(want, cap));
  if (pl.tile) {
    const long long tcap = t.tile_wg_per_cu > 0 ? (long long)s.n_cu * t.tile_wg_per_cu : cap;
    const long long wave_tiles = (long long)((p->width + 7) / 8) * ((rows + 7) / 8);
    grid = (int)std::max<long long>(1, std::min<long long>((wave_tiles + 3) / 4, tcap));
  }
  // OnOff / Flat with pixel-centre rays of a perspective camera in worlds of at most 256 shapes: 16x16 tiles, four pixels
  // per lane (pt_tile4_kernel), a 2x2 block of tiles per workgroup of a 2-D grid
  pl.tile4 = pl.tile && t.tile4 != 0 && !pl.ortho && p->samples_per_side == 0 && s.n_shapes <= 256 &&
             (p->renderer == PT_RENDERER_ONOFF || p->renderer == PT_RENDERER_FLAT) && (n_ranks == 1 || row_block % 16 == 0);
  if (pl.tile4) {
    pl.grid4_x = (unsigned)(((p->width + 15) / 16 + 1) / 2);
    pl.grid4_y = (unsigned)(((rows + 15) / 16 + 1) / 2);
    grid = (int)(pl.grid4_x * pl.grid4_y);
    const size_t scene_bytes = (size_t)s.n_shapes * (PT_PLAN_REC_BYTES + PT_PLAN_AUX_BYTES);
    pl.t4lds = t.tile4_lds != 0 && p->renderer == PT_RENDERER_FLAT && scene_bytes <= 24 * 1024;
    pl.lds_main = pl.t4lds ? scene_bytes : 0;
  }
  // num_of_rays > 1: the tree kernel (one pixel per wave) is latency-bound and wins while flagged pixels are few; a frame
  // FULL of them is throughput-bound and a lane per pixel, refilled from one queue, wins.  Which of the two a frame is only
  // the first pass knows (F, on the device): both kernels are enqueued and pt_unit_scatter writes which one works
  // (PT_Q_CHOICE).  The one-queue kernel keeps its per-lane frame stack (20 doubles per depth and lane) in LDS where that
  // fits (D <= 3), in HBM beyond.
  const size_t q_frame_bytes = (size_t)std::max(p->max_depth, 1) * 20 * B * sizeof(double);  // per workgroup
  // Where that stack lives.  All of it in LDS while it fits (D <= 3 at 20 doubles per depth and lane): one workgroup per CU,
  // one wave per SIMD.  Beyond: SPLIT -- only the deepest slot in LDS (it takes N / (N + 1) of all frame visits), the
  // shallower ones in HBM, two workgroups per CU.  Measured (profiles/r05_q_frames_home.txt): at D <= 3 the split stack is
  // SLOWER than all-LDS (C2 + plane, N = 10: 10.6 against 9.2 ms; all in HBM: 11.4) -- a wave's iteration waits for its
  // slowest lane, and with 64 lanes one of them is at a shallow node nearly every iteration: two waves per SIMD do not
  // buy back an HBM round trip per iteration --; at D > 3 it equals or slightly beats the all-HBM stack it replaces.
  pl.q_home = q_frame_bytes <= PT_LDS_BUDGET ? 1 : 2;
  if (t.q_frames_home == 1 || t.q_frames_home == 2) pl.q_home = (int)t.q_frames_home;
  if (pl.q_home == 1 && q_frame_bytes > PT_LDS_BUDGET) pl.q_home = 2;
  pl.q_lds_frames = pl.q_home == 1;
  const size_t q_frame_lds = pl.q_home == 1 ? q_frame_bytes : (size_t)20 * B * sizeof(double);
  pl.q_alt = pl.tree && t.qchoice != 0;
  if (pl.q_alt) {
    pl.q_tail_budget = t.q_tail_budget >= 0 ? (int)t.q_tail_budget : PT_Q_TAIL_BUDGET_DEFAULT;
    pl.q_few_lanes = t.q_few_lanes >= 0 ? (int)t.q_few_lanes : PT_Q_FEW_LANES_DEFAULT;
    {  // records of PT_PLAN_HANDOVER_HEADER + 20 doubles per node of the deepest stack: at most PT_HANDOVER_CAP of them and 256 MB
      const size_t rec = (size_t)(PT_PLAN_HANDOVER_HEADER + 20 * std::max(p->max_depth, 1));
      const long long by_bytes = std::max<long long>(1024, (long long)(((size_t)256 << 20) / (rec * sizeof(double))));
      pl.handover_cap = (int)std::min<long long>(std::min<long long>(pl.npix, PT_HANDOVER_CAP), by_bytes);
      pl.handover_doubles = (size_t)pl.handover_cap * rec;
    }
    // workgroups of the one-queue kernel a CU holds: by the LDS a workgroup REALLY asks for -- its frames plus the diag records
    // staged behind them when they fit (decided below by the same rule: lds_q), not the frames alone (at D = 1
    // three were assumed where two fit, and the hand-over budget divided by lanes that were not resident)
    const size_t q_diag_here = (size_t)s.n_diag * PT_PLAN_DIAG_BYTES;
    const bool q_diag_fits = s.n_diag > 0 && q_diag_here <= 48 * 1024 &&
                             q_frame_lds + q_diag_here <= (pl.q_home == 1 ? PT_LDS_BUDGET : PT_LDS_BUDGET / 2);
    const size_t q_lds_per_wg = std::max<size_t>(1, q_frame_lds + (q_diag_fits ? q_diag_here : 0));
    int wgq = pl.q_home == 1 ? std::max<int>(1, std::min<int>(3, (int)(PT_LDS_BUDGET / q_lds_per_wg))) : 2;
    if (t.q_wg_per_cu > 0) wgq = (int)t.q_wg_per_cu;
    pl.grid_q = (int)std::max<long long>(1, std::min<long long>(want, (long long)s.n_cu * wgq));
    // The budget: what a lane traces in the whole frame were the work spread evenly -- F flagged pixels x nsamp samples x the
    // mean rays of a sample's tree / the kernel's lanes: a pixel's own chain should not outlast the frame's throughput time.
    // The mean tree: 1 + N (1 + pN + (pN)^2 + ... ), p = the share of rays that hit and scatter on: 0.13 on the three N = 10,
    // D = 3 frames of tools/ray_histogram.py (26 - 43 rays per flagged pixel of a full tree's 1 111).  The swept optimum of a
    // FIXED budget on those frames (300 - 400 at F = 470 - 630 k, profiles/r05_handover_sweep.txt) is this number; smaller
    // frames need a smaller one, shallow or narrow trees (budget > a full tree) none at all.
    pl.q_budget = t.q_budget >= 0 ? (int)t.q_budget : PT_Q_BUDGET_DEFAULT;
    {
      const double pn = 0.13 * (double)p->num_of_rays;
      double mean = 1.0, pw = 1.0;
      for (int d = 0; d < p->max_depth && mean < 1e9; ++d) {
        mean += (double)p->num_of_rays * pw;
        pw *= pn;
      }
      const int ns = p->samples_per_side > 0 ? p->samples_per_side * p->samples_per_side : 1;
      pl.q_budget_per_flagged = mean * (double)ns / ((double)pl.grid_q * B);
      // (never below N + 2: the primary ray, the root's N children and one more -- a pixel that gets there has a child that
      //  scatters, and on a frame with fewer flagged pixels than lanes, where every pixel has had a lane from the start, the
      //  tree kernel is where it belongs: C3 N = 10 1.32 -> 1.25 ms with the one-queue kernel in front, profiles/r05_tree_vs_queue.txt)
      pl.q_budget_min = p->num_of_rays + 2;
    }
  }
  if (pl.path_tiled) {  // first pass (pt_tile_kernel<PATHTRACER>): one wave per 8x8 region
    const long long nreg = (long long)((p->width + PT_PLAN_REGION - 1) / PT_PLAN_REGION) * ((rows + PT_PLAN_REGION - 1) / PT_PLAN_REGION);
    grid = (int)std::max<long long>(1, std::min<long long>((nreg + 3) / 4, cap));
    pl.grid_first = (int)std::max<long long>(1, std::min<long long>((nreg + 3) / 4, (long long)s.n_cu * 8));
    pl.nregions = (int)nreg;
  }
  pl.grid = grid;
  pl.nthreads = grid * B;
  if (pathtracer && p->max_depth < 0) {  // render.py:100-101: every primary call returns black without a world query
    pl.zero_frame = true;
    pl.kernel = PT_KERNEL_NONE;
    return;
  }
  if (pathtracer) {
    // step batching (path_trace): the second pass by regions never mixes the two kinds of step; the one-queue kernel, which
    // also carries the cheap background pixels, starts samples while fewer than 48 lanes hold a ray and queries scattered
    // rays once 16 wait
    pl.p_max_path = t.p_maxpath > 0 ? (int)t.p_maxpath : (pl.path_tiled ? 1 : 48);
    pl.s_min_path = t.s_min > 0 ? (int)t.s_min : (pl.path_tiled ? 1 : 16);
    size_t need = pl.lds_frames ? 0 : (size_t)std::max(p->max_depth, 1) * pl.frame_doubles * (size_t)pl.nthreads * sizeof(double);
    if (pl.q_alt && pl.q_home != 1)  // (the tree kernel's stack is in LDS: the workspace is the one-queue kernel's)
      need = std::max(need, q_frame_bytes * (size_t)pl.grid_q);
    pl.ws_bytes = need;
  }
  // lanes the second pass keeps resident: pt_unit_scatter cuts regions into smaller units (more lanes per pixel) as long
  // as all flagged pixels together still fit them
  pl.lanes_cap = t.unit_lanes_cap >= 0 ? t.unit_lanes_cap : (long long)grid * B;
  if (pl.path_tiled) {
    // (pt_unit_scatter may cut up to four units per resident wave; the tree kernel takes one PIXEL per unit)
    pl.units_need = pl.tree ? (int)std::min<long long>(pl.npix + 64, 0x7fffff00LL) : pl.nregions + (int)(4 * pl.lanes_cap / 64) + 64;
    // PT_PCG_PIXEL: what a sample is assumed to draw before anything is known about its pixel: two jitter numbers and one
    // diffuse bounce (a guess only costs a round when it is wrong, never a bit of the image)
    pl.spec_draws = t.spec_draws >= 0 ? (int)t.spec_draws : (p->samples_per_side > 0 ? 4 : 2);
    pl.grid_scatter = (pl.nregions + PT_PLAN_SCATTER_BLOCK - 1) / PT_PLAN_SCATTER_BLOCK;
  }
  // large scenes: two-level culling (cells of PT_CELL x PT_CELL global pixels, then 8x8 tiles)
  pl.cells_x = (p->width + PT_PLAN_CELL - 1) / PT_PLAN_CELL;
  const int cells_y = (p->height + PT_PLAN_CELL - 1) / PT_PLAN_CELL;
  pl.ncells = pl.cells_x * cells_y;
  pl.cell_stride = (s.n_shapes + 63) / 64 * 64;
  pl.hier = (pl.tile || pl.path_tiled) && !pl.tile4 && !pl.ortho && t.hier_min >= 0 && s.n_shapes > t.hier_min &&
            (n_ranks == 1 || row_block % 8 == 0) && (size_t)pl.ncells * pl.cell_stride * sizeof(unsigned int) <= ((size_t)2 << 30);
  if (pl.hier) {
    // enough (cell group, shape chunk) pairs to fill the chip; a chunk is a multiple of the block
    pl.cell_groups = ((pl.cells_x + 1) / 2) * ((cells_y + 1) / 2);
    const int max_chunks = (s.n_shapes + B - 1) / B;
    const int min_chunks = (s.n_shapes + PT_PLAN_CELL_CHUNK - 1) / PT_PLAN_CELL_CHUNK;  // a chunk's survivors fit in LDS
    pl.cell_chunks = std::max(min_chunks, std::min(max_chunks, (4 * s.n_cu + pl.cell_groups - 1) / pl.cell_groups));
    pl.cell_chunk_len = (max_chunks + pl.cell_chunks - 1) / pl.cell_chunks * B;
  }
  pl.block_h = 1;
  if (pl.path_tiled && !pl.ortho && !pl.hier) {
    // first pass: a workgroup culls a block of four (two) strips, 32 x 32 (32 x 16) pixels, before it looks at the strips
    // -- where that still leaves a block for every other workgroup: pt_tile_kernel<..., BLOCKS>
    const int tiles_x = (p->width + 7) / 8, tiles_y = (rows + 7) / 8;
    const long long strips_x = (tiles_x + 3) / 4;
    if (2 * strips_x * ((tiles_y + 3) / 4) >= (long long)pl.grid_first)  // (measured: still ahead with one block per two workgroups)
      pl.block_h = 4;
    else if (2 * strips_x * ((tiles_y + 1) / 2) >= (long long)pl.grid_first)
      pl.block_h = 2;
    if (t.block_h > 0) pl.block_h = (int)t.block_h;
  }

  // ---- which kernels ----
  if (pl.tile4) {
    pl.kernel = PT_KERNEL_TILE4;
    return;
  }
  if (pl.tile || pl.path_tiled) {
    pl.lds_tile = (size_t)4 * pl.npass * sizeof(unsigned long long);
    pl.kernel = pl.path_tiled ? PT_KERNEL_PATH_REGIONS : PT_KERNEL_TILE;
    if (pl.hier)
      pl.tile_mode = PT_TILE_HIER;
    else if (pl.ortho)
      pl.tile_mode = PT_TILE_ORTHO;
    else if (pl.path_tiled && pl.npass >= 2 && s.dome_shortcut && pl.block_h > 1)  // (big frames: blocks of strips)
      pl.tile_mode = PT_TILE_BLOCKS;
    if (!pl.path_tiled) {
      pl.lds_main = pl.lds_tile;
      return;
    }
    // second pass: the pixels the first one flagged, fullest regions first
    pl.nsamp = p->samples_per_side > 0 ? p->samples_per_side * p->samples_per_side : 1;
    const int pcg_mode = p->pcg_mode == PT_PCG_SEQ ? PT_PCG_PIXEL : p->pcg_mode;  // (SEQ is refused for the path tracer)
    pl.min_rounds = t.unit_min_rounds != 0 ? (int)t.unit_min_rounds : (pcg_mode == PT_PCG_SAMPLE ? -2 : 16);  // (< 0: see unit_ppu)
    // The flagged pixels from which the one-queue kernel takes the frame (q_alt): where its estimate falls below the tree
    // kernel's.  Both fitted to measurements on the MI355X (tools/tree_vs_queue.py: 25 frames of three scenes, N = 2 ... 20,
    // D = 2 ... 8; ns per flagged pixel and sample; round 4: profiles/r04_tree_vs_queue.txt, round 5 with the hand-over of
    // heavy pixels: profiles/r05_tree_vs_queue.txt):
    //   tree kernel       F x tT,  tT = 4.5 + 0.045 min(R, 500)   (R = sum of N^d, the rays of a full tree: one pixel at a
    //                     time on each of the 8 n_cu resident waves, a handful of rounds per family of children)
    //   one-queue kernel  min(R, 80) x step + F x tQ,  step = 6 + 0.02 n_shapes us: the chain of its deepest lane -- which,
    //                     since round 5, ends at the budget and goes on in the tree kernel at a sixth of the time per ray: a
    //                     frame of full trees (R = 1 111) costs under a millisecond on top of its throughput, where round 4
    //                     paid R x step = 7 ms --, tQ = (2 + 0.01 n_shapes)(1 + R / 800)
    // FITTED RANGE: 2 <= N <= 20, D <= 8 (D > 3 with the stack in HBM: x 1.3), <= 300 shapes; outside it the
    // estimate is clamped to that range's corner instead of extrapolated.
    pl.q_min = -1;
    if (pl.q_alt) {
      const int fit_n = std::min(std::max(p->num_of_rays, 2), 20), fit_d = std::min(std::max(p->max_depth, 0), 8);
      const double fit_shapes = (double)std::min(s.n_shapes, 300);
      double tree_rays = 1.0, pw = 1.0;
      for (int d = 1; d <= fit_d; ++d) {
        pw *= (double)fit_n;
        tree_rays += pw;
      }
      const double step_ns = (6.0 + 0.02 * fit_shapes) * 1e3;
      const double t_tree = (4.5 + 0.045 * std::min(tree_rays, 500.0)) * (2048.0 / (8.0 * s.n_cu));
      const double t_queue = (2.0 + 0.01 * fit_shapes) * (1.0 + tree_rays / 800.0);
      if (t_tree > t_queue) pl.q_min = (long long)std::min(1e15, 1.1 * std::min(tree_rays, 80.0) * step_ns / (t_tree - t_queue));
      if (t.q_min_flagged >= 0) pl.q_min = t.q_min_flagged;
      if (t.qchoice == 2) pl.q_min = 0;
    }
    // worlds without a grid and without the ball hierarchy (< 128 spheres): the query's large-world paths compiled out
    const bool small_world = t.small_query != 0 && !s.has_grid && s.bs_levels == 0;
    pl.lds_main = pl.lds_tile + (pl.lds_frames ? frame_lds : 0) + staged_bytes;
    if (pl.tree) {
      pl.kernel = PT_KERNEL_PATH_TREE;
      pl.second = small_world ? PT_SECOND_TREE_LEAN : PT_SECOND_TREE;
      if (t.tree_jump != 0 && p->num_of_rays <= 63) {  // the leaf rounds' jump coefficients: 2 x 4 x 64 pairs of 64-bit words
        const size_t at = (pl.lds_main + 7) / 8 * 8;
        if (at + 8192 <= PT_LDS_BUDGET / 2) {
          pl.tree_jump_lds = (int)(at / 8);
          pl.lds_main = at + 8192;
        }
      }
      // small worlds: the shapes' records ride in LDS for shading (a round is one dependent chain: every trip to L2 is in it)
      if (small_world && t.tree_scene_lds != 0) {
        const size_t at = (pl.lds_main + 255) / 256 * 256;
        const size_t scene_bytes = (size_t)s.n_shapes * (PT_PLAN_REC_BYTES + PT_PLAN_AUX_BYTES);
        if (scene_bytes <= 32 * 1024 && at + scene_bytes <= PT_LDS_BUDGET / 2) {
          pl.scene_lds = (int)(at / 8);
          pl.lds_main = at + scene_bytes;
          pl.second = PT_SECOND_TREE_LEAN_SCENE;
        }
      }
    } else if (pl.lds_frames && pl.scene_lds >= 0 && small_world) {
      pl.second = PT_SECOND_REGIONS_LDS_SCENE_LEAN;
    } else if (pl.lds_frames && pl.scene_lds >= 0) {
      pl.second = PT_SECOND_REGIONS_LDS_SCENE;
    } else if (pl.lds_frames && t.small_query != 0 && !s.has_grid) {  // (no grid: its walk compiled out)
      pl.second = PT_SECOND_REGIONS_LDS_NOGRID;
    } else if (pl.lds_frames) {
      pl.second = PT_SECOND_REGIONS_LDS;
    } else {
      pl.second = PT_SECOND_REGIONS_HBM;
    }
    if (pl.q_alt) {
      // ... and the one-queue kernel behind it (a lane per pixel: 20 doubles per depth and lane, its own grid, its own slots
      // for the ray counts); the scale+translate records behind the frame stack when they fit
      const size_t q_diag_bytes = (size_t)s.n_diag * PT_PLAN_DIAG_BYTES;
      const bool q_diag = s.n_diag > 0 && q_diag_bytes <= 48 * 1024 &&
                          q_frame_lds + q_diag_bytes <= (pl.q_home == 1 ? PT_LDS_BUDGET : PT_LDS_BUDGET / 2);
      pl.q_diag_lds = q_diag ? (int)(q_frame_lds / 8) : -1;
      pl.lds_q = q_frame_lds + (q_diag ? q_diag_bytes : 0);
      if (pl.q_home == 2)
        pl.alt = small_world ? PT_ALT_FLAGGED_LEAN_SPLIT : PT_ALT_FLAGGED_SPLIT;
      else
        pl.alt = small_world ? PT_ALT_FLAGGED_LEAN_LDS : PT_ALT_FLAGGED_LDS;
    }
    return;
  }
  // one lane per pixel
  pl.kernel = pathtracer ? PT_KERNEL_PATH : PT_KERNEL_SIMPLE;
  pl.simple_hoist = pl.hoist;
  if (pathtracer) pl.lds_main = 0;
}

static inline const char *pt_plan_kernel_name(const PtPlan &pl, int renderer, int which, char *buf, size_t n) {
  // which: 0 = pre-pass (pt_cell_kernel), 1 = first / tile kernel, 2 = second pass or the only render kernel, 3 = alternative
  static const char *R[4] = {"ONOFF", "FLAT", "PATHTRACER", "POINTLIGHT"};
  static const char *TM[4] = {"", ", HIER", ", ORTHO", ", BLOCKS"};
  buf[0] = 0;
  const char *r = R[renderer >= 0 && renderer < 4 ? renderer : 0];
  if (pl.npix == 0) return buf;
  if (which == 0) {
    if (pl.hier) snprintf(buf, n, "pt_cell_kernel");
  } else if (which == 1) {
    if (pl.path_tiled && !pl.zero_frame) snprintf(buf, n, "pt_tile_kernel<PATHTRACER%s>", TM[pl.tile_mode]);
  } else if (which == 2) {
    if (pl.zero_frame)
      snprintf(buf, n, "memset");
    else if (pl.tile4)
      snprintf(buf, n, "pt_tile4_kernel<%s, %s>", r, pl.t4lds ? "LDS" : "noLDS");
    else if (pl.path_tiled) {
      static const char *S2[9] = {"", "pt_path_regions_kernel<LDS, SCENE, LEAN>", "pt_path_regions_kernel<LDS, SCENE>",
                                  "pt_path_regions_kernel<LDS, NOGRID>", "pt_path_regions_kernel<LDS>", "pt_path_regions_kernel<HBM>",
                                  "pt_path_tree_kernel<LEAN>", "pt_path_tree_kernel", "pt_path_tree_kernel<LEAN, SCENE>"};
      snprintf(buf, n, "%s", S2[pl.second]);
    } else if (pl.tile)
      snprintf(buf, n, "pt_tile_kernel<%s%s>", r, TM[pl.tile_mode]);
    else if (pl.kernel == PT_KERNEL_PATH)
      snprintf(buf, n, "pt_path_kernel");
    else
      snprintf(buf, n, "pt_simple_kernel<%s, %s>", r, pl.simple_hoist ? "HOIST" : "noHOIST");
  } else if (which == 3) {
    static const char *A[5] = {"", "pt_path_flagged_kernel<LEAN, LDS>", "pt_path_flagged_kernel<LDS>",
                               "pt_path_flagged_kernel<LEAN, SPLIT>", "pt_path_flagged_kernel<SPLIT>"};
    snprintf(buf, n, "%s", A[pl.alt]);
  }
  return buf;
}
