// pt_kernels.h — hand-written HIP kernels for gfx950 (CDNA4): the per-pixel ray-trace/shade path.
//
// What the kernels restate (reference paths relative to src/pytracer/):
//   ImageTracer.fire_all_rays   imagetracer.py:60-110     per-pixel driver, S x S stratified jitter
//   Camera.fire_ray             camera.py:59-78, 103-124  primary rays
//   World.ray_intersection      world.py:51-69            closest hit over all shapes, in list order
//   Sphere/Plane.ray_intersection shapes.py:97-131, 163-189
//   OnOff/Flat/PathTracer/PointLight renderers            render.py:42-193
//   pigments, BRDF scattering   materials.py:50-196, geometry.py:247-262
//   PCG                         pcg.py:23-62
//
// Numerics: fp64 throughout, compiled with -ffp-contract=off (the reference never fuses a*b+c);
// every expression keeps the reference's operation order, so wherever no libm transcendental is
// involved the result is bit-identical to the reference arithmetic with x*x for x**2
// (SURVEY.md H1/H2).  sqrt and '/' are IEEE-correct on gfx950.
//
// Execution model (MI355X, wave64, 256-thread workgroups; DESIGN.md section 4 has the measurements):
//   pt_tile4_kernel       OnOff / Flat, pixel-centre rays, perspective camera, <= 256 shapes: a wave owns a 16x16 tile,
//                         FOUR pixels per lane; one cone + one cull per tile, survivors in SGPR masks, every survivor's
//                         hoisted record (scalar loads) serves four independent rays per lane.
//   pt_tile_kernel        the other OnOff / Flat / PointLight frames and the path tracer's FIRST pass: a wave owns an
//                         8x8 tile, one pixel per lane, survivor masks in LDS replayed per jitter sample; strips and
//                         blocks of tiles share a cull; cell lists (pt_cell_kernel) for worlds of > 256 shapes.
//   pt_path_regions_kernel  the path tracer's second pass for num_of_rays = 1: work units of flagged pixels, a
//                         pixel's samples spread over lanes, speculated generator states committed in order;
//                         scattered rays walk per-lane candidate lists (world_query_lanes) or a uniform grid.
//   pt_path_tree_kernel   ... for num_of_rays > 1: one pixel per wave, a node's children on lanes.
//   pt_path_flagged_kernel  ... for frames FULL of scattering pixels: a lane per flagged pixel from one queue, in front of
//                         the tree kernel, which takes over the pixels whose trees outlast the frame (PT_Q_CHOICE, PT_Q_HEAVY).
//   pt_simple_kernel / pt_path_kernel   one lane per pixel (worlds of fewer than four shapes, PTRACE_CULL=0): the
//                         shape loop index is wave-uniform, records come through the scalar cache into SGPRs.
// MFMA is not used (no dense contraction on this path); what binds is vector issue and dependent latency
// (profiles/r03_issue_rates.txt: fp64 4, fp32 / int32 2, SALU 4 SIMD-cycles per wave-instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptrace.h"
#include "pt_layout.h"

#define PT_DEV static __device__ __forceinline__
#define PT_PI 3.141592653589793
#define PT_BLOCK 256
#ifndef PT_WAVES_SIMPLE
#define PT_WAVES_SIMPLE 4
#endif
#ifndef PT_WAVES_TILE
#define PT_WAVES_TILE 4
#endif
#ifndef PT_REGION
#define PT_REGION 8  // path tracer: a wave's region is PT_REGION x PT_REGION pixels
#endif

// Uniform (wave-invariant) reads go through the constant address space so the backend emits
// s_load_* (scalar cache -> SGPRs) instead of per-lane global loads.
typedef const __attribute__((address_space(4))) double *pt_kdouble;
typedef const __attribute__((address_space(4))) int32_t *pt_kint;
#define PT_KD(p) ((pt_kdouble)(const void *)(p))
#define PT_KI(p) ((pt_kint)(const void *)(p))

// Register budget: the argument block is ~90 dwords.  Only the fields the shape loop needs are read
// as ordinary by-value kernel arguments (they stay in SGPRs); everything else is re-read at its point
// of use from a copy of the block in DEVICE memory (a.cold; scalar cache / L2) through a laundered
// pointer, so the compiler cannot hoist those loads to the kernel entry and then spill them inside
// the hot loop.  (Not from the kernarg segment itself: that lives in host memory, ~1.5 us per miss.)
typedef const __attribute__((address_space(4))) PtKArgs *pt_kargs;
PT_DEV pt_kargs cold_args(const PtKArgs &a) {
  unsigned long long p = (unsigned long long)a.cold;
  asm volatile("" : "+s"(p));
  return (pt_kargs)p;
}

// The path tracer's queue block (layout: see pt_unit_scatter) exists twice; frame f uses block f & 1 (a.qpar, a
// by-value argument so that the device copy of the argument block stays the same from frame to frame) and its
// path kernel zeroes the other one for the next frame.
#ifndef PT_QUEUE_WORDS
#define PT_QUEUE_WORDS 512
#endif
#define PT_QUEUE_HEADS 256
// num_of_rays > 1 on a perspective camera: BOTH second-pass kernels are enqueued behind the first pass and this word of the
// frame's queue block, written by pt_unit_scatter from F (the flagged pixels the first pass counted), says which of
// them works -- 0: pt_path_tree_kernel (one pixel per wave: few flagged pixels, the frame waits for its deepest tree),
// 1: pt_path_flagged_kernel (a lane per flagged pixel, refilled from one queue: frames full of flagged pixels are
// throughput-bound).  The other one returns at once -- unless the one-queue kernel hands pixels over (round 5): a lane per
// pixel walks a pixel's rays one after the other, so a frame would wait for its heaviest pixels' chains (1 111 rays at the
// CLI's N = 10, D = 3) while the lanes that have finished idle.  Such a lane writes its pixel's state into a record and
// appends the pixel to a unit list of its own (PT_Q_HEAVY units in `units_handed`; path_trace says when), and the tree kernel, enqueued BEHIND the one-queue kernel, finishes those pixels from where they
// stand with a node's children on lanes.
#define PT_Q_CHOICE 200
#define PT_Q_HEAVY 201
#define PT_Q_BUDGET 202  // rays after which a lane hands its pixel over, from the flagged pixels the first pass counted (pt_unit_scatter)
#define PT_HANDOVER_HEADER 8  // doubles in front of a record's nodes: generator state and increment, sample, nodes, rays, the pixel's sum
#ifndef PT_UNIT_SHARDS
#define PT_UNIT_SHARDS 8
#endif
static_assert(PT_QUEUE_HEADS + 32 * PT_UNIT_SHARDS <= PT_QUEUE_WORDS, "the shard heads must lie inside the queue block");
PT_DEV unsigned long long *pt_queue(const PtKArgs &a) { return cold_args(a)->queue + (size_t)a.qpar * PT_QUEUE_WORDS; }
PT_DEV unsigned long long *pt_queue_next(const PtKArgs &a) { return cold_args(a)->queue + (size_t)(a.qpar ^ 1) * PT_QUEUE_WORDS; }

struct V3 {
  double x, y, z;
};
struct Ray {
  V3 o, d;
  double tmin;
};
struct Hit {
  V3 wp, n;
  double u, v;
};

// the parts, in order (each relies on the ones before it)
#include "pt_math.h"
#include "pt_query.h"
#include "pt_shade.h"
#include "pt_camera.h"
#include "pt_simple.h"
#include "pt_tile.h"
#include "pt_path.h"
#include "pt_tree.h"
#include "pt_probes.h"
