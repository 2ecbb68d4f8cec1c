/* ptrace_debug.h — diagnostic entry points of libptrace.so.  NOT part of the drop-in boundary (include/ptrace.h is):
 * nothing a renderer needs is declared here.  They exist so that the tests can look at pieces of the device path on
 * their own -- the arithmetic primitives, the culling predicate, the hit record, camera rays, BRDF scattering -- and
 * compare them with the reference's own fixtures (tests/golden/g3_shapes.npz, g4_camera.npz, g6_scatter_onb.npz:
 * the values of /root/reference/tests/test_all.py:607-869 and of the reference run on random inputs).
 * Same conventions as ptrace.h: extern "C", plain pointers, 0 or a negative PT_ERR_* code.
 */
#ifndef PTRACE_DEBUG_H
#define PTRACE_DEBUG_H

#include "ptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Elementwise device primitives (op 0 sqrt, 1 x/y, 2 sin, 3 cos, 4 atan2(x, y), 5 acos, 6 floor, 7 x*y+x unfused,
 * 8/9 PCG outputs / floats, 10 pcg_advance == n steps): IEEE exactness and ulp distance to libm. */
int pt_debug_probe(int op, const double *x, const double *y, double *out, int n);

/* keep[i] (World.shapes index): does the conservative cull of the primary rays through the image rectangle
 * [x0, x1] x [row0, row1 + 1] keep shape i?  pixel_x >= 0: the cone of that one pixel instead.   */
int pt_debug_cull_probe(pt_scene *scene, const pt_camera *cam, int width, int height, int x0, int x1, int row0, int row1,
                        int pixel_x, int pixel_row, int *keep);

/* Shape.ray_intersection / World.ray_intersection on the device (shapes.py:97-131, 163-189; world.py:51-69) through
 * the kernels' own world_query + hit_details.  rays: n x 8 doubles (origin, dir, tmin, tmax).  shape_index >= 0: that
 * shape of World.shapes alone; -1: the closest hit over the world.  out: n x 12 doubles
 * (hit 0/1, t, world_point[3], normal[3] normalised as world.py:66-68 does for the winner, u, v, World.shapes index, 0). */
int pt_debug_hit_probe(pt_scene *scene, int shape_index, const double *rays, int n, double *out);

/* World.ray_intersection (anyhit = 0; world.py:51-69) / World.is_point_visible's blocker search (anyhit = 1;
 * world.py:71-80) through the query the scattered and shadow rays use: per-lane candidates from the conservative fp32
 * filter -- or the cell walk in scenes with a grid -- then exact visits.  The 64 rays of a group of 64 run as one wave;
 * a ray with tmin < 0 is an idle lane.  rays: n x 8 doubles as above; out: n x 4 doubles (hit 0/1, t, World.shapes
 * index, 0), any-hit: (blocked 0/1, 0, 0, 0).  Must agree with pt_debug_hit_probe(scene, -1, ...), which runs every
 * shape through the exact test: the filter may only drop shapes the ray cannot meet. */
int pt_debug_lanes_probe(pt_scene *scene, int anyhit, const double *rays, int n, double *out);

/* ImageTracer.fire_ray + Camera.fire_ray on the device (imagetracer.py:48-58, camera.py:59-78, 103-124) through the
 * kernels' own primary_ray.  pix: n x 4 doubles (col, row, u_pixel, v_pixel); out: n x 7 doubles (origin, dir, tmin). */
int pt_debug_camera_probe(const pt_camera *cam, int width, int height, const double *pix, int n, double *out);

/* BRDF.scatter_ray on the device (materials.py:132-152, 175-196 with geometry.py:247-262).  in: n x 12 doubles
 * (brdf kind, PCG init_state, PCG init_seq, normal[3], incoming dir[3], point[3]); out: n x 7 doubles (origin, dir,
 * tmin); state_after: n generator states after the call. */
int pt_debug_scatter_probe(const double *in, int n, double *out, unsigned long long *state_after);

/* The 16 leading words of the path tracer's queue block of the last frame (unit counts; section sums of a
 * -DPT_DEBUG_TIME build). */
int pt_debug_read_queue(pt_scene *scene, unsigned long long *out16);
/* num_of_rays > 1, last frame: the pixels the one-queue kernel handed to the tree kernel in the middle of their trees
 * (0 when the tree kernel rendered the frame alone), and the ray budget the device derived from the flagged pixels. */
int pt_debug_handed_over(pt_scene *scene, unsigned long long *pixels, unsigned long long *budget);

/* What a frame WILL launch -- the plan pt_render / pt_render_device follow (csrc/pt_plan.h) -- computed on the HOST from a scene
 * DESCRIPTION, a camera and the parameters: no device is touched, so the choice of kernels, grids, LDS and thresholds is
 * testable on any machine (tests/test_plan.py).  n_cu: compute units to plan for (<= 0: 256, the MI355X); dome_shortcut as
 * pt_set_dome_shortcut.  Kernel names are the templates' with their variant spelled out, "" where a stage does not run. */
typedef struct pt_plan_info {
  int32_t kernel;            /* PT_KERNEL_* of the render kernel proper (before the device's choice, see alt_kernel) */
  int32_t rows;              /* image rows of this rank */
  int64_t npix;
  char pre_kernel[48];       /* "pt_cell_kernel" for worlds of more than 256 shapes, else "" */
  char first_kernel[64];     /* path tracer: the first pass */
  char main_kernel[64];      /* the render kernel proper / the path tracer's second pass */
  char alt_kernel[64];       /* num_of_rays > 1: the one-queue kernel enqueued in front of the tree kernel (the device picks; pixels over alt_budget go to the tree kernel) */
  int32_t grid, grid_first, grid_alt;         /* workgroups of 256 threads */
  int32_t grid4_x, grid4_y, npx;              /* pt_tile4_kernel: its 2-D grid and pixels per lane */
  int64_t lds_first, lds_main, lds_alt;       /* dynamic LDS per workgroup, bytes */
  int32_t frame_stack_home;  /* path tracer: 0 none, 1 LDS, 2 HBM, 3 split (the deepest slot in LDS, the others in HBM) */
  int32_t alt_frame_stack_home;
  int32_t frame_doubles;     /* fields per stack frame */
  int64_t workspace_bytes;   /* frame stack in HBM */
  int64_t q_min_flagged;     /* flagged pixels from which the device lets the one-queue kernel work (-1: never) */
  int32_t wg_per_cu, block_h, hier, ortho, hoist, tile4_lds;
  int32_t n_spheres, n_diag, has_grid, ball_levels;  /* the scene facts the plan was made from */
  int32_t units_need, nregions, min_rounds, spec_draws;
  int32_t alt_budget;        /* rays after which the one-queue kernel hands a pixel over to the tree kernel behind it (0: never; -1, the default: derived on the device per frame from the flagged-pixel count -- pt_plan.h q_budget_per_flagged, never below num_of_rays + 2) */
  int32_t _reserved[7];
} pt_plan_info;
int pt_debug_plan(const pt_scene_desc *desc, const pt_camera *cam, const pt_params *params, int n_cu, int dome_shortcut,
                  pt_plan_info *out);
/* The plan a handle's next frame follows (its own n_cu and switches). */
int pt_debug_plan_scene(pt_scene *scene, const pt_camera *cam, const pt_params *params, pt_plan_info *out);
/* One of the debug / measurement switches (csrc/pt_plan.h: PT_TUNING_TABLE; by field name or by its PTRACE_* environment
 * name).  The table is read from the environment when the library first needs it; this changes it afterwards, also between
 * scenes of one process.  None changes a pixel.  Returns PT_ERR_INVALID for an unknown name. */
int pt_debug_set_tuning(const char *name, long long value);
int pt_debug_get_tuning(const char *name, long long *value);

#ifdef PT_DEBUG_TIME /* instrumented builds only (tools/dbg*.py) */
int pt_debug_read_dbg(unsigned long long *out8, int reset);
int pt_debug_read_lat_hist(unsigned long long *out160, int clear);
int pt_debug_read_lat_events(unsigned long long *out, int clear);
int pt_debug_read_unitlog(unsigned long long *out, int n_units);
int pt_debug_read_trace(unsigned long long *out, int n);
#endif

#ifdef __cplusplus
}
#endif
#endif
