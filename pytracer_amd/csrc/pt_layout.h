// pt_layout.h — device-side (HBM) layout of a flattened scene and the kernel argument block.
//
// The C-ABI hands over structure-of-arrays host arrays (include/ptrace.h); pt_scene_upload packs
// them into two array-of-records tables sized for how the kernels touch them:
//
//   PtShapeRec (128 B)  — what the per-ray shape loop reads for EVERY shape: the 3x4 inverse
//                         transform + kind.  The loop index is wave-uniform, so a record is
//                         fetched once per wave through the scalar cache (s_load_dwordx8/x16)
//                         into SGPRs, or staged in LDS for tiles of big scenes.
//   PtShapeAux (256 B)  — what only the CLOSEST hit needs: the forward transform and materials.
//                         Gathered per lane after the loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(128) PtShapeRec {
  double invm[12];  // rows 0..2 of transformation.invm, row-major
  int32_t kind;     // PT_SHAPE_*
  int32_t index;    // position in World.shapes (records are grouped spheres-first; ties between
                    // equal t go to the lower index, world.py:62)
  double fro2;      // spheres: upper bound of |invm 3x3|_F^2, +inf unless the scale is within 1e-6 .. 1e6
  double _pad[2];
};
static_assert(sizeof(PtShapeRec) == 128, "PtShapeRec must be 128 B");

struct alignas(256) PtShapeAux {
  double m[12];  // rows 0..2 of transformation.m
  double pig_c1[3], pig_c2[3];  // (needs_uv == 0, i.e. both pigments uniform: pig_c2 = pig_c1 + emi_c1, the Flat colour of the shape)
  double emi_c1[3], emi_c2[3];
  double pig_steps, emi_steps, brdf_param;
  int32_t brdf_kind, pig_kind, emi_kind, pig_tex, emi_tex, needs_uv;
  double _pad[2];
};
static_assert(sizeof(PtShapeAux) == 256, "PtShapeAux must be 256 B");

struct alignas(64) PtLight {
  double pos[3];
  double color[3];
  double radius;
  double _pad;
};

struct PtTex {
  int32_t w, h;
  int64_t offset;  // in doubles into tex_data
};

// Spheres whose inverse transform is "scale + translate" (the 3x3 block of invm is diagonal): the
// common case (translation * scaling).  For them x*0.0 terms only ever add a signed zero, so when no
// product can be a zero itself (wave-level guard in world_query) the object-space ray is
//   o' = o * s + t,  d' = d * s        (9 flop instead of 33) — bit-identical to the full product.
struct alignas(64) PtDiagRec {
  double s[3];     // invm[0], invm[5], invm[10]
  double t[3];     // invm[3], invm[7], invm[11]
  int32_t tnz;     // bit c set: t[c] != 0 (a zero o[c]*s[c] is then absorbed exactly)
  int32_t _pad[3];
};
static_assert(sizeof(PtDiagRec) == 64, "PtDiagRec must be 64 B");

// ... and their per-camera constants for hoisted primary rays
struct alignas(64) PtHoistDiag {
  double s[3];
  double o[3];  // invm * origin, full-product arithmetic (pt_prep_hoist)
  double c;     // |o'|^2 - 1
  double _pad;
};
static_assert(sizeof(PtHoistDiag) == 64, "PtHoistDiag must be 64 B");

// World-space bounding sphere of a shape as float4 (cx, cy, cz, r): centre = M*0, r >= largest singular
// value of M's 3x3 block, inflated for the fp32 rounding of the centre.  Used only to REJECT shapes a
// whole 8x8-pixel tile of primary rays cannot touch; every shape that survives still goes through the
// exact reference arithmetic.  r < 0: never rejected (planes, non-finite or inconsistent transforms).

// Per-shape constants of the primary rays of a perspective camera (all share one origin):
// o' = invm * origin and c = |o'|^2 - 1, computed in the reference's operation order by
// pt_prep_hoist so the hoisted loop reproduces the per-ray arithmetic bit for bit.
struct alignas(32) PtHoist {
  double ox, oy, oz, c;
};

struct PtKArgs {
  const PtKArgs *cold;              // this same block in device memory (see cold_args())
  const PtShapeRec *recs;
  const PtShapeAux *aux;
  const PtHoist *hoist;
  const PtDiagRec *diag;            // [n_diag], parallel to recs[0..n_diag)
  const PtHoistDiag *hoist_diag;    // [n_diag]
  const float4 *bounds;             // [n_shapes], slot order: (cx, cy, cz, r)
  const float *bsoa;                // the same as four arrays x[], y[], z[], r'^2[] of bs_stride floats (per-ray prefilter),
                                    // then the balls around every 8 (gs_stride) and every 64 (cs_stride) sphere slots
  int bs_stride, gs_stride, cs_stride;
  int bs_levels;                    // 1: the group/chunk balls are meaningful (>= 128 spheres, slots in Morton order)
  float bs_rmax[3];                 // the largest ordinary r' among the spheres' balls, the groups', the chunks' (the filter's margin for |o|)
  // Uniform grid over the bounded, ordinary-sized spheres: scattered and shadow rays of scenes of >= 128 spheres walk
  // it cell by cell (world_query_lanes).  Cell c holds items [grid_cells[c] >> 8, + (grid_cells[c] & 255)): the ball
  // (x, y, z, r') of a sphere for the conservative fp32 test and its slot; grid_occ has one bit per cell ("holds
  // something").  Spheres the grid does not hold (much larger than the rest: a sky dome; or without a bound) are
  // listed in grid_always and tested for every ray.
  const unsigned *grid_cells;       // null: no grid
  const unsigned *grid_occ;
  const float4 *grid_balls;
  const unsigned short *grid_slots;
  const int *grid_always;
  float grid_far_eo;                // a ray whose 1e-6 * max|origin component| exceeds this is too far away for the walk's margins
  int grid_n_always, grid_occ_lds;  // grid_occ_lds: where the kernel staged grid_occ in LDS (4-byte words), -1: read it from memory
  int grid_res[3];
  float grid_min[3], grid_max[3], grid_cell[3], grid_inv[3];
  int scene_lds;                    // second path-tracer pass: where recs[] then aux[] are staged in LDS (8-byte words, 256-B aligned), -1 = not
  int diag_lds;                     // second path-tracer pass: where its copy of diag[] starts in LDS (8-byte words), -1 = not staged
  const PtLight *lights;
  const PtTex *tex;
  const double *tex_data;
  void *out;                       // this rank's rows, compact
  double *ws;                      // path-tracer frame stack: [slot][field][thread]
  unsigned long long *ray_counter; // per-workgroup partial counts; may be null
  unsigned long long *queue;       // path tracer: the two queue blocks (pt_path.h: pt_queue)
  int block_h;                     // path tracer's first pass, BLOCKS variant: strips per block (4 or 2)
  int qpar;                        // ... and which of them this frame uses (by value; the device copy of the block holds 0)
  const int4 *units;               // path tracer, second pass: (region, first flagged pixel | pixels << 8, region mask) per work unit
  int dome_slot;                   // path tracer, first pass: the sphere the camera is deepest inside (uniform pigments), or -1
  int dome_shortcut;               // 0: every tile goes through rays (pt_set_dome_shortcut; a measurement switch)
  int4 *units_handed;              // num_of_rays > 1: the units of the pixels the one-queue kernel hands to the tree kernel (handover_cap of them)
  double *handover;                // num_of_rays > 1: records of the pixels the one-queue kernel hands to the tree kernel (PT_Q_HEAVY)
  int handover_cap;                // ... how many fit
  int q_budget;                    // one-queue kernel of num_of_rays > 1: a pixel that has traced this many rays is handed to the tree kernel (0: never)
  int q_tail_budget;               // ... or this many rays after the pixel queue has run dry (0: never)
  int q_few_lanes;                 // ... or when, the queue dry, this many lanes of its wave or fewer still hold a pixel (0: never)
  int tree_jump_lds;               // pt_path_tree_kernel: where the leaf rounds' state-jump coefficients live in LDS (8-byte words; 2 x 4 x 64 pairs), -1 = none
  int dbg_trace_unit;              // -DPT_DEBUG_TIME builds: the unit whose steps are traced (PTRACE_TRACE_UNIT)
  int spec_draws;                  // ... PT_PCG_PIXEL: draws per sample assumed for a pixel nothing is known about yet
  unsigned long long *region_mask; // path tracer: [region] pixels the first pass left to pt_path_kernel
  unsigned char *region_keys;      // path tracer: [region] their number
  unsigned int *cell_list;         // large scenes: [cell][cell_stride] surviving slots (pt_cell_kernel)
  int *cell_count;                 // [cell] survivors
  int cells_x, cell_stride;
  int p_max_path, s_min_path;      // path tracer step batching (see pt_path_kernel)
  int count_base;                  // pt_path_kernel: first slot of its per-workgroup ray counts in ray_counter
  long long npix;                  // pixels this launch covers (rows_local * W)
  int n_shapes, n_lights;
  int n_spheres;                   // recs[0..n_spheres) are spheres, recs[n_spheres..n_shapes) planes
  int n_diag;                      // recs[0..n_diag) are the scale+translate spheres
  int nthreads;                    // grid * block
  int frame_doubles;               // fields per stack frame
  int rows_local;                  // image rows this rank renders
  int npass;                       // ceil(n_shapes / 64): culling passes per tile
  // camera (camera.py)
  int cam_kind;
  double cam_m[12];
  double cam_dist, cam_aspect;
  // perspective primary direction as an affine function of the image position, fp32, for the tile cones only:
  // d(x, y) = cone_d0 + x * cone_dx + y * cone_dy  (x in pixels from the left, y in global rows from the top)
  float cone_d0[3], cone_dx[3], cone_dy[3], cone_apex[3];
  // image / renderer parameters (pt_params)
  int W, H, S;
  int N, D, rr;
  int pcg_mode;
  unsigned long long s0, q0;
  int row_block, n_ranks, rank;
  int out_f32;
  double bg[3], onoff[3], ambient[3];
};
