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
//   pt_simple_kernel / pt_path_kernel   one lane per pixel (tiny worlds, orthogonal path tracing, PTRACE_CULL=0): the
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
#ifndef PT_WAVES_PATH
#define PT_WAVES_PATH 3
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

// ---- PCG-XSH-RR 64/32 (pcg.py:23-62) -----------------------------------------------------------
struct Pcg {
  uint64_t state, inc;
  unsigned n;  // draws since the counter was last cleared (only the path tracer's second pass reads it)
};
PT_DEV uint32_t pcg_next(Pcg &p) {
  const uint64_t old = p.state;
  p.n++;
  p.state = old * 6364136223846793005ULL + p.inc;
  const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
  const uint32_t rot = (uint32_t)(old >> 59);
  return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
PT_DEV void pcg_seed(Pcg &p, uint64_t init_state, uint64_t init_seq) {
  p.state = 0;
  p.inc = (init_seq << 1) | 1ULL;
  pcg_next(p);
  p.state += init_state;
  pcg_next(p);
  p.n = 0;
}
// The state `delta` draws further on: state -> state * M^delta + inc * (M^(delta-1) + ... + 1) mod 2^64, by
// repeated squaring (the generator is a linear congruential one; identical to `delta` calls of pcg_next).
PT_DEV uint64_t pcg_advance(uint64_t state, uint64_t inc, unsigned delta) {
  uint64_t acc_mul = 1ULL, acc_add = 0ULL, cur_mul = 6364136223846793005ULL, cur_add = inc;
  while (delta) {
    if (delta & 1u) {
      acc_mul *= cur_mul;
      acc_add = acc_add * cur_mul + cur_add;
    }
    cur_add = (cur_mul + 1ULL) * cur_add;
    cur_mul *= cur_mul;
    delta >>= 1;
  }
  return acc_mul * state + acc_add;
}
// pcg.py:60-62: random() / 0xFFFFFFFF, an fp64 division (inclusive 1.0)
PT_DEV double pcg_float(Pcg &p) { return (double)pcg_next(p) / 4294967295.0; }

// ---- transformations.py:58-86 ----------------------------------------------------------------------
template <typename P>
PT_DEV V3 xf_point(P m, V3 p) {
  V3 r;
  r.x = p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3];
  r.y = p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7];
  r.z = p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11];
  return r;
}
template <typename P>
PT_DEV V3 xf_vec(P m, V3 v) {
  V3 r;
  r.x = v.x * m[0] + v.y * m[1] + v.z * m[2];
  r.y = v.x * m[4] + v.y * m[5] + v.z * m[6];
  r.z = v.x * m[8] + v.y * m[9] + v.z * m[10];
  return r;
}
template <typename P>
PT_DEV V3 xf_normal(P im, V3 n) {  // transpose of the inverse
  V3 r;
  r.x = n.x * im[0] + n.y * im[4] + n.z * im[8];
  r.y = n.x * im[1] + n.y * im[5] + n.z * im[9];
  r.z = n.x * im[2] + n.y * im[6] + n.z * im[10];
  return r;
}
PT_DEV double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vec.normalize / Normal.normalize (geometry.py:130-136, 219-225): x*x here
PT_DEV V3 normalize3(V3 a) {
  const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  V3 r = {a.x / n, a.y / n, a.z / n};
  return r;
}
// ocml's fp64 sin/cos/atan2/acos are polynomial kernels with ~25 double constants each.  Inlined, LICM
// hoists those constants out of the pixel loops into VGPRs that stay live for the whole kernel (~50
// registers for code that runs once per bounce at most).  Behind a call they live only in the callee.
#define PT_NOINLINE static __device__ __attribute__((noinline))
PT_NOINLINE double pt_sin(double x) { return sin(x); }
PT_NOINLINE double pt_cos(double x) { return cos(x); }
PT_NOINLINE double pt_atan2(double y, double x) { return atan2(y, x); }
PT_NOINLINE double pt_acos(double x) { return acos(x); }

PT_DEV double max2(double a, double b) { return (b > a) ? b : a; }  // Python max(a, b)

// ---- the shape loop: World.ray_intersection (world.py:51-69) ---------------------------------------
// Returns the record slot of the closest shape hit in (r.tmin, best_t) or -1; best_t is updated.
//
// Records are grouped [scale+translate spheres | other spheres | planes] so each loop body is
// branch-free on the shape kind; a tie in t between a plane and an earlier winner is resolved by the
// original list index, which reproduces "first shape in list order wins" (world.py:62, strict <).
// ANYHIT: leave as soon as every active lane has some hit (OnOff, shadow rays) — the hit/miss
//   answer is identical, only `which` shape is unspecified.
// HOIST: primary rays of a perspective camera share their origin, so invm*origin and c=|o'|^2-1
//   are per-shape constants, computed in the same operation order by pt_prep_hoist.
//
// Scale+translate fast path: with invm = diag(s) | t the reference's full product is
//   d'_x = (d.x*s0 + d.y*0) + d.z*0,   o'_x = ((o.x*s0 + o.y*0) + o.z*0) + t0.
// Adding a signed zero changes a value only if that value is itself a zero, so the short forms
// d.x*s0 and o.x*s0 + t0 are bit-identical unless a product is +-0 (or non-finite).  WaveGuard
// proves per wave, per ray, that no lane can be in that case; otherwise the full product runs.
struct WaveGuard {
  bool fast;        // every active lane: 1e-100 <= |d.c| <= 1e100 (and |o.c| <= 1e100)
  unsigned ozmask;  // bit c: some active lane has |o.c| < 1e-100 (its product may be a zero)
};

template <bool HOIST>
PT_DEV WaveGuard wave_guard(const Ray &r, bool active) {
  const double lo = 1e-100, hi = 1e100;
  const double ax = fabs(r.d.x), ay = fabs(r.d.y), az = fabs(r.d.z);
  bool bad = !(ax >= lo && ax <= hi && ay >= lo && ay <= hi && az >= lo && az <= hi);
  WaveGuard g;
  g.ozmask = 0;
  if (!HOIST) {
    const double px = fabs(r.o.x), py = fabs(r.o.y), pz = fabs(r.o.z);
    bad = bad || !(px <= hi && py <= hi && pz <= hi);
    g.ozmask = (__ballot(active && px < lo) ? 1u : 0u) | (__ballot(active && py < lo) ? 2u : 0u) |
               (__ballot(active && pz < lo) ? 4u : 0u);
  }
  g.fast = __ballot(active && bad) == 0ULL;
  return g;
}

// Exact tie in t with the current winner: the shape that comes first in World.shapes wins
// (world.py:62 replaces the closest hit only on a strict <).  Evaluated only when t == best_t.
PT_DEV bool tie_wins(const PtKArgs &a, int slot, int best) {
  return best >= 0 && *PT_KI(&a.recs[slot].index) < a.recs[best].index;
}

// shapes.py:103-121 given the object-space ray (ox..dz, aa = |d'|^2, cc = |o'|^2 - 1): the first root
// inside (tmin, tmax) is this shape's hit; it replaces the winner if closer (world.py:62).
#define PT_SPHERE_ROOTS(SLOT)                                                         \
  do {                                                                                \
    const double bb = 2.0 * (ox * dx + oy * dy + oz * dz);                            \
    const double delta = bb * bb - 4.0 * aa * cc;                                     \
    if (active && delta > 0.0) {                                                      \
      const double sd = sqrt(delta);                                                  \
      const double den = 2.0 * aa;                                                    \
      double t = (-bb - sd) / den;                                                    \
      bool ok = (t > tmin) && (t < tmax);                                             \
      if (!ok) {                                                                      \
        t = (-bb + sd) / den;                                                         \
        ok = (t > tmin) && (t < tmax);                                                \
      }                                                                               \
      if (ok && (t < best_t || (!ANYHIT && t == best_t && tie_wins(a, (SLOT), best)))) { \
        best_t = t;                                                                   \
        best = (SLOT);                                                                \
      }                                                                               \
    }                                                                                 \
  } while (0)

// shapes.py:168-175 given the z row of the object-space ray
#define PT_PLANE_HIT(SLOT)                                                            \
  do {                                                                                \
    if (active && !(fabs(dz) < 1e-5)) {                                               \
      const double t = -oz / dz;                                                      \
      if (!(t <= tmin) && !(t >= tmax) &&                                             \
          (t < best_t || (!ANYHIT && t == best_t && tie_wins(a, (SLOT), best)))) {    \
        best_t = t;                                                                   \
        best = (SLOT);                                                                \
      }                                                                               \
    }                                                                                 \
  } while (0)

#define PT_ANYHIT_EXIT()                                             \
  do {                                                               \
    if (ANYHIT) {                                                    \
      if (__ballot(active && best < 0) == 0ULL) return best;         \
    }                                                                \
  } while (0)

template <bool ANYHIT, bool HOIST>
PT_DEV int world_query(const PtKArgs &a, const Ray &r, double tmax, double &best_t, bool active) {
  int best = -1;
  best_t = INFINITY;
  const double tmin = r.tmin;
  const int nd = a.n_diag;
  const int ns = a.n_spheres;
  const int n = a.n_shapes;
  int first_general = 0;

  // ---- scale+translate spheres: 30 flop per test (18 hoisted) instead of 54 (30) ----
  if (nd > 0) {
    const WaveGuard g = wave_guard<HOIST>(r, active);
    if (g.fast) {
      first_general = nd;
      if (HOIST) {
        // software pipeline: the record of shape i+1 is requested (s_load) before shape i is evaluated
        pt_kdouble base = PT_KD(a.hoist_diag);
        double n0 = base[0], n1 = base[1], n2 = base[2], n3 = base[3], n4 = base[4], n5 = base[5], n6 = base[6];
        for (int i = 0; i < nd; ++i) {
          const double s0 = n0, s1 = n1, s2 = n2, ox = n3, oy = n4, oz = n5, cc = n6;
          pt_kdouble h = base + (size_t)((i + 1 < nd) ? i + 1 : i) * 8;
          n0 = h[0];
          n1 = h[1];
          n2 = h[2];
          n3 = h[3];
          n4 = h[4];
          n5 = h[5];
          n6 = h[6];
          const double dx = r.d.x * s0, dy = r.d.y * s1, dz = r.d.z * s2;
          const double aa = dx * dx + dy * dy + dz * dz;
          PT_SPHERE_ROOTS(i);
          PT_ANYHIT_EXIT();
        }
      } else {
        pt_kdouble base = PT_KD(a.diag);
        double n0 = base[0], n1 = base[1], n2 = base[2], n3 = base[3], n4 = base[4], n5 = base[5];
        int ntnz = *PT_KI(&a.diag[0].tnz);
        for (int i = 0; i < nd; ++i) {
          const double s0 = n0, s1 = n1, s2 = n2, t0 = n3, t1 = n4, t2 = n5;
          const int tnz = ntnz;
          const int nx = (i + 1 < nd) ? i + 1 : i;
          pt_kdouble h = base + (size_t)nx * 8;
          n0 = h[0];
          n1 = h[1];
          n2 = h[2];
          n3 = h[3];
          n4 = h[4];
          n5 = h[5];
          ntnz = *PT_KI(&a.diag[nx].tnz);
          double dx, dy, dz, ox, oy, oz;
          if ((g.ozmask & ~(unsigned)tnz) == 0u) {
            dx = r.d.x * s0;
            dy = r.d.y * s1;
            dz = r.d.z * s2;
            ox = r.o.x * s0 + t0;
            oy = r.o.y * s1 + t1;
            oz = r.o.z * s2 + t2;
          } else {  // a zero product could meet a zero translation: full product for this shape
            pt_kdouble m = PT_KD(a.recs[i].invm);
            dx = r.d.x * m[0] + r.d.y * m[1] + r.d.z * m[2];
            dy = r.d.x * m[4] + r.d.y * m[5] + r.d.z * m[6];
            dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
            ox = r.o.x * m[0] + r.o.y * m[1] + r.o.z * m[2] + m[3];
            oy = r.o.x * m[4] + r.o.y * m[5] + r.o.z * m[6] + m[7];
            oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
          }
          const double aa = dx * dx + dy * dy + dz * dz;
          const double cc = (ox * ox + oy * oy + oz * oz) - 1.0;
          PT_SPHERE_ROOTS(i);
          PT_ANYHIT_EXIT();
        }
      }
    }
  }
  // ---- spheres, full 3x4 product: shapes.py:102-121 (54 flop generic, 30 hoisted) ----
  for (int i = first_general; i < ns; ++i) {
    pt_kdouble m = PT_KD(a.recs[i].invm);
    const double dx = r.d.x * m[0] + r.d.y * m[1] + r.d.z * m[2];
    const double dy = r.d.x * m[4] + r.d.y * m[5] + r.d.z * m[6];
    const double dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
    const double aa = dx * dx + dy * dy + dz * dz;
    double ox, oy, oz, cc;
    if (HOIST) {
      pt_kdouble h = PT_KD(&a.hoist[i]);
      ox = h[0];
      oy = h[1];
      oz = h[2];
      cc = h[3];
    } else {
      ox = r.o.x * m[0] + r.o.y * m[1] + r.o.z * m[2] + m[3];
      oy = r.o.x * m[4] + r.o.y * m[5] + r.o.z * m[6] + m[7];
      oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
      cc = (ox * ox + oy * oy + oz * oz) - 1.0;
    }
    PT_SPHERE_ROOTS(i);
    PT_ANYHIT_EXIT();
  }
  // ---- planes: shapes.py:168-175, only the z row of the object-space ray decides ----
  for (int i = ns; i < n; ++i) {
    pt_kdouble m = PT_KD(a.recs[i].invm);
    const double dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
    double oz;
    if (HOIST) {
      oz = PT_KD(&a.hoist[i])[2];
    } else {
      oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
    }
    PT_PLANE_HIT(i);
    PT_ANYHIT_EXIT();
  }
  return best;
}

// ---- pieces of the scattered-ray query of the path tracer's second pass (world_query_lanes) ----------
// Same arithmetic as world_query<false, false>; what changes is which spheres are looked at and when.
//  * A sphere with bb > 0 and cc >= 0 (origin outside, moving away) is skipped without roots: then
//    4*aa*cc >= 0, so delta <= fl(bb*bb), sqrt(delta) <= sqrt(fl(bb*bb)) = bb exactly (radix 2, no
//    underflow: guarded by bb > 1e-100), hence both of the reference's computed roots are <= 0 < tmin.
//  * The far root is computed when some lane's near root fails its range test (the reference does so
//    per ray; a far root nobody selects changes nothing).
struct LatCand {
  double aa, bb, cc, delta;
};
#define PT_LAT_INRANGE(T) (((T) > tmin) && ((T) < tmax))
#define PT_LAT_ROOT1(C, T1) T1 = (-(C).bb - sqrt((C).delta)) / (2.0 * (C).aa)
#define PT_LAT_ROOT2(C, T2) T2 = (-(C).bb + sqrt((C).delta)) / (2.0 * (C).aa)
PT_DEV LatCand lat_cand(double ox, double oy, double oz, double dx, double dy, double dz) {
  LatCand c;
  c.aa = dx * dx + dy * dy + dz * dz;
  c.cc = (ox * ox + oy * oy + oz * oz) - 1.0;
  c.bb = 2.0 * (ox * dx + oy * dy + oz * dz);
  c.delta = c.bb * c.bb - 4.0 * c.aa * c.cc;
  return c;
}

#ifndef PT_SPARSE_RAYS
#define PT_SPARSE_RAYS 16  // world_query_lanes: at most this many live rays -> one ball per lane, rays take turns
#endif
#ifndef PT_SPARSE_MAX_SPHERES
#define PT_SPARSE_MAX_SPHERES 1024  // ... in scenes up to this size (beyond, skipping whole chunks and groups pays more)
#endif
#ifdef PT_DEBUG_TIME
__device__ unsigned long long pt_dbg[8];
#ifdef PT_DEBUG_TIME
// latency of single vector-memory operations, log2 buckets: [0] the load of a unit's descriptor, [1] what was still
// outstanding before it, [2] the returning atomic on a shard's head, [3] the sparse path's load of one ball per lane
__device__ unsigned long long pt_lat_hist[5][32];  // ([4]: scattered-ray queries by the number of live rays, bins of 2)
#define PT_VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0)
// ... and the slow ones one by one: (100 MHz wall clock at the end, cycles, which | xcc << 8 | HW_ID << 16)
#define PT_LAT_EVENTS 4096
__device__ unsigned long long pt_lat_events[PT_LAT_EVENTS * 3 + 1];
PT_DEV void lat_note(int which, unsigned long long dt) {
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&pt_lat_hist[which][63 - __clzll((long long)(dt | 1ULL))], 1ULL);
    if (dt >= 8192ULL) {
      const unsigned long long at = atomicAdd(&pt_lat_events[PT_LAT_EVENTS * 3], 1ULL);
      if (at < PT_LAT_EVENTS) {
        pt_lat_events[at * 3] = __builtin_amdgcn_s_memrealtime();
        pt_lat_events[at * 3 + 1] = dt;
        pt_lat_events[at * 3 + 2] = (unsigned long long)which | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 8) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 16);
      }
    }
  }
}
#endif  // world_query_lanes: prefilter cycles, walk cycles, iterations, calls; 4..7: units / rounds
#define PT_DBG_WAVES 16384
__device__ unsigned long long pt_dbg_wave[PT_DBG_WAVES * 8];  // the same, per wave, summed up at the end of the kernel
static __device__ void pt_dbg_flush() {
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
    for (int k = 0; k < 8; ++k) {
      if (wv[k]) atomicAdd(&pt_dbg[k], wv[k]);
      wv[k] = 0ULL;
    }
  }
}
#endif
// (the kernels' one dynamic LDS block, viewed as 64-bit words and as doubles; see pt_tile_kernel, path_trace)
extern __shared__ unsigned long long pt_lds_masks[];
extern __shared__ double pt_lds_f64[];
typedef const __attribute__((address_space(3))) PtShapeRec *pt_lds_rec;  // the shapes' records when a kernel staged them in LDS
typedef const __attribute__((address_space(3))) PtShapeAux *pt_lds_aux;

// ---- closest hit, every lane on its own candidate list ---------------------------------------------------
// The scattered rays of a wave point everywhere: for almost every sphere SOME lane's line meets it, so
// a wave-uniform loop runs the fp64 candidate (and mostly the roots) for all of them.  Here each lane
// first marks, in a 64-bit mask per 64 spheres, the spheres ITS ray can touch at all -- a conservative
// test against the bounding spheres in packed fp32 (two spheres per v_pk instruction) -- and then
// walks its own mask, fetching the records by lane-private index.  The exact arithmetic of a visited
// sphere is the reference's; a sphere that is not visited has delta <= 0 or both roots negative:
//  * line test: |v x d|^2 > (R'^2 + 8e-6 |v|^2) |d|^2 with v = C - o.  R' is the bounding radius
//    inflated at upload for the fp32 rounding of C (and 1e-5 relative), 1e-6 |o| covers the rounding
//    of the origin, 8e-6 |v|^2 the fp32 evaluation, the rounding of d and the slack the fp64 test
//    itself has around delta = 0 (~16 ulp of |v|^2).
//  * behind test: v.d < 0 and (v.d)^2 > 1.001 R'^2 |d|^2 + the same slack: the whole ball lies behind
//    the origin, both roots are negative by a margin far above fp64 rounding.
// NaN/inf on either side keep the sphere.  Order of visits differs from the list order only in WHEN a
// candidate is seen; ties in t go to the lower World.shapes index as everywhere.
// ANYHIT (shadow rays, world.py:71-80 / shapes.py:133-151): a lane stops at its first sphere with a root in
// (tmin, tmax); with a finite tmax the prefilter also drops balls that lie entirely beyond the end
// point ((v - tmax d).d > 0 and its square > 1.001 R'^2 |d|^2 + slack).
// SMALL (chosen by the host for worlds without a grid and without the ball hierarchy, i.e. fewer than 128 spheres):
// the grid walk and the chunk / group levels are compiled out -- less code and fewer live registers in kernels whose
// time is the latency of one wave's instruction stream.
// (LEAN = 1 is SMALL; LEAN = 2: worlds without a grid but with the ball hierarchy, 128 ... 1023 spheres: only the grid walk
//  is compiled out)
template <bool ANYHIT, int LEAN = 0>
PT_DEV int world_query_lanes(const PtKArgs &a, const Ray &r, double tmax, double &best_t, bool active, int diag_lds) {
  constexpr bool SMALL = LEAN == 1;
  constexpr bool NOGRID = LEAN != 0;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(4))) float *pt_kfloat;
  int best = -1;
  best_t = INFINITY;
  const double tmin = r.tmin;
  const int nd = a.n_diag, ns = a.n_spheres, n = a.n_shapes;
  const float tmaxf_dd = (float)tmax;  // (multiplied by |d|^2 below)

  const float ofx = (float)r.o.x, ofy = (float)r.o.y, ofz = (float)r.o.z;
  const float dfx = (float)r.d.x, dfy = (float)r.d.y, dfz = (float)r.d.z;
  const float dd = dfx * dfx + dfy * dfy + dfz * dfz;
  const float omax = fmaxf(fmaxf(fabsf(ofx), fabsf(ofy)), fabsf(ofz));
  const float eo = 1e-6f * omax;  // (the grid's measure of "far away")
  // The conservative filter: "is the centre of the ball farther from the ray (the segment, for shadow rays) than r'?", in
  // fp32 with the direction normalised: with v = C - o, vd = v.d^, vc = vd clamped to [0, length],
  //   dist^2 = |v|^2 - vd^2 + (vd - vc)^2   and the ball is rejected iff   (1 - 8e-6) |v|^2 - E - r'^2 - vd^2 + (vd - vc)^2 > 0
  // (evaluated divided by 1 - 8e-6: d^ and the length carry 1 / sqrt(1 - 8e-6), the tables r'^2 / (1 - 8e-6)).
  // 8e-6 |v|^2 covers the fp32 evaluation (|v|^2 and vd each within a few 2^-24, d^ within 3e-7 of unit length, vd^2 <= |v|^2);
  // E = e (2 r'max + e) >= (r' + e)^2 - r'^2 with e = 2e-7 max|o| covers the rounding of o to fp32 (<= sqrt(3) 2^-24 max|o|);
  // r' itself (pt_scene_upload) covers the rounding of C and the slack of the fp64 test around delta = 0.  The tables
  // hold r'^2 rounded up, 1e38 (never rejected) where there is no usable bound; a lane whose ray is not ordinary
  // (|o| > 1e17, |d|^2 outside 1e-30 .. 1e30, NaN) keeps everything (`wild`): with both guards no intermediate value
  // overflows or is a NaN, so the SIGN of the last operation is the verdict -- no compare, no select.
  const bool wild = !(omax <= 1e17f && dd >= 1e-30f && dd <= 1e30f);
  const float rn = __builtin_amdgcn_rsqf(dd) * 1.0000041f;  // (1 / sqrt(1 - 8e-6) = 1.0000040000240...: rounded up)
  const float hx = dfx * rn, hy = dfy * rn, hz = dfz * rn;
  const float e7 = 2e-7f * omax;
  const float Ek0 = e7 * (2.0f * a.bs_rmax[0] + e7) * 1.0001f, Ek1 = e7 * (2.0f * a.bs_rmax[1] + e7) * 1.0001f,
              Ek2 = e7 * (2.0f * a.bs_rmax[2] + e7) * 1.0001f;
  const float tlen = ANYHIT ? (float)tmax * (dd * rn) * (1.0f + 1e-5f) : 0.0f;  // (tmax = inf: inf)
  pt_kfloat bsx = (pt_kfloat)(const void *)a.bsoa, bsy = bsx + a.bs_stride, bsz = bsy + a.bs_stride, bsr = bsz + a.bs_stride;

  // this lane may use o*s + t, d*s for a scale+translate sphere whose translation absorbs its zero products
  const double lo = 1e-100, hi = 1e100;
  const double adx = fabs(r.d.x), ady = fabs(r.d.y), adz = fabs(r.d.z);
  const double aox = fabs(r.o.x), aoy = fabs(r.o.y), aoz = fabs(r.o.z);
  const bool lane_fast = adx >= lo && adx <= hi && ady >= lo && ady <= hi && adz >= lo && adz <= hi && aox <= hi && aoy <= hi &&
                         aoz <= hi;
  const unsigned ozmask = (aox < lo ? 1u : 0u) | (aoy < lo ? 2u : 0u) | (aoz < lo ? 4u : 0u);

#ifdef PT_DEBUG_TIME
  unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_pre = 0, dbg_walk = 0, dbg_it = 0;
  {
    const int np_ = __popcll(__ballot(active));
    if (!ANYHIT && (threadIdx.x & 63) == 0) atomicAdd(&pt_lat_hist[4][np_ >= 62 ? 31 : np_ >> 1], 1ULL);
  }
#endif
  typedef float f8 __attribute__((ext_vector_type(8)));
  typedef const __attribute__((address_space(4))) f8 *pt_kf8;
  // two balls at a time (cx, cy, cz | cr2 = r'^2): NEGATIVE = this lane's ray cannot meet that ball
  auto far2 = [&](f2 cx, f2 cy, f2 cz, f2 cr2, float Ek) -> f2 {
#pragma clang fp contract(fast)  // (a conservative fp32 filter, not reference arithmetic: fused multiply-adds only make it more exact)
    const f2 vx = cx - ofx, vy = cy - ofy, vz = cz - ofz;
    const f2 vd = vx * hx + vy * hy + vz * hz;
    const f2 vvm = vz * vz + (vy * vy + (vx * vx - Ek));
    const f2 P = vvm - cr2;
    if (!ANYHIT) {
      const f2 vc = {__builtin_fmaxf(vd.x, 0.0f), __builtin_fmaxf(vd.y, 0.0f)};
      return vc * vc - P;
    } else {
      const f2 vc = {__builtin_amdgcn_fmed3f(vd.x, 0.0f, tlen), __builtin_amdgcn_fmed3f(vd.y, 0.0f, tlen)};
      const f2 e = vd - vc;
      return vd * vd - (e * e + P);
    }
  };
  // Scenes of >= 128 spheres: the slots are in Morton order (pt_scene_upload), every 8 consecutive
  // spheres have a ball around their bounding spheres and so have every 64; a chunk or a group that no
  // lane's ray can touch is skipped whole.
  const int levels = SMALL ? 0 : a.bs_levels;
  pt_kfloat gsx = bsr + a.bs_stride, gsy = gsx + a.gs_stride, gsz = gsy + a.gs_stride, gsr = gsz + a.gs_stride;
  pt_kfloat csx = gsr + a.gs_stride, csy = csx + a.cs_stride, csz = csy + a.cs_stride, csr = csz + a.cs_stride;
  // ---- the exact test of a candidate (shared by every way of finding candidates below) ----
  struct DiagL {
    double s0, s1, s2, t0, t1, t2;
    int tnz;
  };
  auto fetch = [&](int slot) {
    DiagL g;
    g.s0 = g.s1 = g.s2 = g.t0 = g.t1 = g.t2 = 0.0;
    g.tnz = 0;
    if (slot < nd) {
      if (diag_lds >= 0) {  // the table was staged in LDS by the kernel (path_trace)
        const int o = diag_lds + slot * 8;
        g.s0 = pt_lds_f64[o];
        g.s1 = pt_lds_f64[o + 1];
        g.s2 = pt_lds_f64[o + 2];
        g.t0 = pt_lds_f64[o + 3];
        g.t1 = pt_lds_f64[o + 4];
        g.t2 = pt_lds_f64[o + 5];
        g.tnz = (int)(unsigned)pt_lds_masks[o + 6];
      } else {
        const PtDiagRec *q = a.diag + slot;
        g.s0 = q->s[0];
        g.s1 = q->s[1];
        g.s2 = q->s[2];
        g.t0 = q->t[0];
        g.t1 = q->t[1];
        g.t2 = q->t[2];
        g.tnz = q->tnz;
      }
    }
    return g;
  };
  // the object-space ray of candidate `slot`: shapes.py:102, full product or (bit-identical under the guard) o*s + t, d*s
  auto object_ray = [&](int slot, bool has, const DiagL &g, double &ox, double &oy, double &oz, double &dx, double &dy,
                        double &dz) {
    if (!has || (slot < nd && lane_fast && (ozmask & ~(unsigned)g.tnz) == 0u)) {  // (!has: values unused)
      dx = r.d.x * g.s0;
      dy = r.d.y * g.s1;
      dz = r.d.z * g.s2;
      ox = r.o.x * g.s0 + g.t0;
      oy = r.o.y * g.s1 + g.t1;
      oz = r.o.z * g.s2 + g.t2;
    } else {
      const double *m = a.recs[slot].invm;
      dx = r.d.x * m[0] + r.d.y * m[1] + r.d.z * m[2];
      dy = r.d.x * m[4] + r.d.y * m[5] + r.d.z * m[6];
      dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
      ox = r.o.x * m[0] + r.o.y * m[1] + r.o.z * m[2] + m[3];
      oy = r.o.x * m[4] + r.o.y * m[5] + r.o.z * m[6] + m[7];
      oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
    }
  };
  auto take_if_closer = [&](int slot, bool need, bool ok, double t) {
    if (need && ok) {
      bool take = t < best_t;
      if (!ANYHIT && !take && t == best_t && best >= 0) take = a.recs[slot].index < a.recs[best].index;
      if (take) {
        best_t = t;
        best = slot;
      }
    }
  };
  // Two candidates per call: a visit is a chain of dependent fp64 operations (transform, discriminant, sqrt,
  // division) that a single wave cannot overlap with anything but another, independent visit.  The winner does
  // not depend on the order of visits (ties go by World.shapes index).
  auto visit2 = [&](int slot_a, bool has_a, int slot_b, bool has_b) {
    const DiagL ga = fetch(slot_a), gb = fetch(slot_b);
    double oxa, oya, oza, dxa, dya, dza, oxb, oyb, ozb, dxb, dyb, dzb;
    object_ray(slot_a, has_a, ga, oxa, oya, oza, dxa, dya, dza);
    object_ray(slot_b, has_b, gb, oxb, oyb, ozb, dxb, dyb, dzb);
    const LatCand ca = lat_cand(oxa, oya, oza, dxa, dya, dza), cb = lat_cand(oxb, oyb, ozb, dxb, dyb, dzb);
    const bool need_a = has_a && ca.delta > 0.0 && !(ca.bb > 1e-100 && ca.cc >= 0.0);
    const bool need_b = has_b && cb.delta > 0.0 && !(cb.bb > 1e-100 && cb.cc >= 0.0);
    if (__ballot(need_a || need_b) != 0ULL) {
      double t1a, t1b, t2a = 0.0, t2b = 0.0;
      PT_LAT_ROOT1(ca, t1a);
      PT_LAT_ROOT1(cb, t1b);
      if (__ballot((need_a && !PT_LAT_INRANGE(t1a)) || (need_b && !PT_LAT_INRANGE(t1b))) != 0ULL) {
        PT_LAT_ROOT2(ca, t2a);
        PT_LAT_ROOT2(cb, t2b);
      }
      const bool ok1a = PT_LAT_INRANGE(t1a), ok1b = PT_LAT_INRANGE(t1b);
      take_if_closer(slot_a, need_a, ok1a || PT_LAT_INRANGE(t2a), ok1a ? t1a : t2a);
      take_if_closer(slot_b, need_b, ok1b || PT_LAT_INRANGE(t2b), ok1b ? t1b : t2b);
    }
  };
  // the same test for ONE ball, the ray's constants given explicitly (scalar form, see the sparse path below)
  auto far1 = [&](float cx, float cy, float cz, float cr2, float sox, float soy, float soz, float shx, float shy, float shz,
                  float sEk) -> float {
#pragma clang fp contract(fast)
    const float vx = cx - sox, vy = cy - soy, vz = cz - soz;
    const float vd = vx * shx + vy * shy + vz * shz;
    const float vvm = vz * vz + (vy * vy + (vx * vx - sEk));
    const float P = vvm - cr2;
    if (!ANYHIT) {
      const float vc = __builtin_fmaxf(vd, 0.0f);
      return vc * vc - P;
    } else {
      const float vc = __builtin_amdgcn_fmed3f(vd, 0.0f, tlen);
      const float e = vd - vc;
      return vd * vd - (e * e + P);
    }
  };
  // ---- scenes with a grid: every lane walks the cells its ray crosses ----
  // Three phases, repeated until every lane's walk has left the grid: (1) a 3D-DDA in fp32 on the fp32 copy of
  // the ray collects up to eight OCCUPIED cells (one bit per cell, from LDS when the kernel staged it); (2) the
  // balls of those cells' spheres go through the conservative fp32 test of the prefilter, survivors join the lane's
  // candidate list (eight 16-bit slots); (3) the candidates are visited two at a time.  Why no hit can be lost:
  // pt_scene_upload (the margin a sphere is entered with covers the fp32 ray's deviation and the DDA's rounding).
  if (!NOGRID && a.grid_cells) {
    pt_kargs ga = cold_args(a);
    // spheres outside the grid (a dome, unbounded transforms): tested for every ray
    const int n_always = ga->grid_n_always;
    for (int k = 0; k < n_always; k += 2) {
      const int sa = PT_KI(ga->grid_always)[k], sb = k + 1 < n_always ? PT_KI(ga->grid_always)[k + 1] : 0;
      visit2(sa, active && !(ANYHIT && best >= 0), sb, !ANYHIT && active && k + 1 < n_always);
      if (ANYHIT && k + 1 < n_always) visit2(sb, active && best < 0, 0, false);
    }
    const int rx = ga->grid_res[0], ry = ga->grid_res[1], rz = ga->grid_res[2];
    const float bx0 = ga->grid_min[0], by0 = ga->grid_min[1], bz0 = ga->grid_min[2];
    const float cwx = ga->grid_cell[0], cwy = ga->grid_cell[1], cwz = ga->grid_cell[2];
    const unsigned *cells = ga->grid_cells;
    const unsigned *occ_mem = ga->grid_occ;
    const float4 *balls = ga->grid_balls;
    const unsigned short *slots = ga->grid_slots;
    const int occ_lds = ga->grid_occ_lds;
    const unsigned *occ_shared = (const unsigned *)pt_lds_masks;
    // the part of the ray inside the grid's box: [t0, t1] (slabs; a zero component: inside the slab or never)
    float t0 = 0.0f, t1 = ANYHIT ? (float)tmax * (1.0f + 1e-5f) : INFINITY;
    // A ray that starts very far from the grid (a far point of an unbounded plane, a distant mirror): the fp32 copy of
    // its origin is off by ~6e-8 |o|, which the margin the spheres were entered with (sized from the GRID's coordinates,
    // pt_scene_upload) no longer covers.  Such a lane does not walk; it runs every sphere's ball through the
    // conservative filter below, whose slack does scale with |o| (eo).
    const bool far = active && !(eo <= ga->grid_far_eo);  // (NaN origin: far)
    bool walking = active && !far && !(ANYHIT && best >= 0);
    {
      const float lo_[3] = {bx0, by0, bz0}, hi_[3] = {ga->grid_max[0], ga->grid_max[1], ga->grid_max[2]};
      const float o_[3] = {ofx, ofy, ofz}, d_[3] = {dfx, dfy, dfz};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (fabsf(d_[q]) > 1e-30f) {
          const float inv = 1.0f / d_[q];
          const float ta = (lo_[q] - o_[q]) * inv, tb = (hi_[q] - o_[q]) * inv;
          t0 = fmaxf(t0, fminf(ta, tb));
          t1 = fminf(t1, fmaxf(ta, tb));
        } else {
          walking = walking && o_[q] >= lo_[q] && o_[q] <= hi_[q];
        }
      }
    }
    walking = walking && (t0 <= t1);  // (NaN: no walk.  The box is padded far beyond every entered ball: no margin needed here)
    int cx = 0, cy = 0, cz = 0;
    float tmx = 3.0e38f, tmy = 3.0e38f, tmz = 3.0e38f;
    const float big = 3.0e38f;
    const int sx = dfx > 0.0f ? 1 : -1, sy = dfy > 0.0f ? 1 : -1, sz = dfz > 0.0f ? 1 : -1;
    const float tdx = fabsf(dfx) > 1e-30f ? fabsf(cwx / dfx) : big, tdy = fabsf(dfy) > 1e-30f ? fabsf(cwy / dfy) : big,
                tdz = fabsf(dfz) > 1e-30f ? fabsf(cwz / dfz) : big;
    if (walking) {
      const float px = ofx + dfx * t0, py = ofy + dfy * t0, pz = ofz + dfz * t0;
      cx = (int)floorf((px - bx0) * ga->grid_inv[0]);
      cy = (int)floorf((py - by0) * ga->grid_inv[1]);
      cz = (int)floorf((pz - bz0) * ga->grid_inv[2]);
      cx = cx < 0 ? 0 : (cx >= rx ? rx - 1 : cx);
      cy = cy < 0 ? 0 : (cy >= ry ? ry - 1 : cy);
      cz = cz < 0 ? 0 : (cz >= rz ? rz - 1 : cz);
      tmx = fabsf(dfx) > 1e-30f ? (bx0 + (float)(cx + (sx > 0)) * cwx - ofx) / dfx : big;
      tmy = fabsf(dfy) > 1e-30f ? (by0 + (float)(cy + (sy > 0)) * cwy - ofy) / dfy : big;
      tmz = fabsf(dfz) > 1e-30f ? (bz0 + (float)(cz + (sz > 0)) * cwz - ofz) / dfz : big;
    }
    int guard = rx + ry + rz + 3;  // (a walk crosses at most that many cell walls)
    while (__ballot(walking) != 0ULL) {
      // (1) up to eight occupied cells of this lane's walk (cell ids are < 2^18: three per 64-bit word would do, two words of 4 x 16 bits hold ids < 65536, so larger grids use the low 16 bits of (id) only when they fit: res <= 32^3)
      unsigned long long ce_lo = 0ULL, ce_hi = 0ULL;
      int n_ce = 0;
      while (__ballot(walking && n_ce < 8) != 0ULL) {
        if (walking && n_ce < 8) {
          const int cid = (cz * ry + cy) * rx + cx;
          const unsigned w = occ_lds >= 0 ? occ_shared[occ_lds + (cid >> 5)] : occ_mem[cid >> 5];
          if ((w >> (cid & 31)) & 1u) {
            if (n_ce < 4)
              ce_lo |= (unsigned long long)(unsigned)cid << (16 * n_ce);
            else
              ce_hi |= (unsigned long long)(unsigned)cid << (16 * (n_ce - 4));
            n_ce++;
          }
          // next cell: across the nearest of the three cell walls ahead
          const float tn = fminf(tmx, fminf(tmy, tmz));
          bool out = tn > t1 || --guard <= 0;
          if (tmx <= tmy && tmx <= tmz) {
            cx += sx;
            tmx += tdx;
            out = out || cx < 0 || cx >= rx;
          } else if (tmy <= tmz) {
            cy += sy;
            tmy += tdy;
            out = out || cy < 0 || cy >= ry;
          } else {
            cz += sz;
            tmz += tdz;
            out = out || cz < 0 || cz >= rz;
          }
          if (out) walking = false;
        }
      }
      // (2) the spheres of those cells against the conservative fp32 test; (3) visit the survivors
      unsigned long long ca_lo = 0ULL, ca_hi = 0ULL;
      int n_ca = 0, last = -1;
      auto flush = [&]() {
        for (int k = 0; __ballot(k < n_ca) != 0ULL; k += ANYHIT ? 1 : 2) {
          const bool has_a = k < n_ca && !(ANYHIT && best >= 0), has_b = !ANYHIT && k + 1 < n_ca;
          const int slot_a = (int)(((k < 4 ? ca_lo : ca_hi) >> (16 * (k & 3))) & 0xffffULL);
          const int slot_b = (int)((((k + 1) < 4 ? ca_lo : ca_hi) >> (16 * ((k + 1) & 3))) & 0xffffULL);
          visit2(slot_a, has_a, slot_b, has_b);
        }
        ca_lo = 0ULL;
        ca_hi = 0ULL;
        n_ca = 0;
      };
      for (int k = 0; __ballot(k < n_ce) != 0ULL; ++k) {
        const bool has_c = k < n_ce;
        const int cid = has_c ? (int)(((k < 4 ? ce_lo : ce_hi) >> (16 * (k & 3))) & 0xffffULL) : 0;
        const unsigned wv = cells[cid];
        const unsigned cnt_c = has_c ? (wv & 255u) : 0u, off_c = wv >> 8;
        for (unsigned q = 0; __ballot(q < cnt_c) != 0ULL; ++q) {
          const bool has_i = q < cnt_c;
          const float4 b = balls[off_c + (has_i ? q : 0u)];
          const int slot = (int)slots[off_c + (has_i ? q : 0u)];
          const bool rej = !wild && far1(b.x, b.y, b.z, b.w, ofx, ofy, ofz, hx, hy, hz, Ek0) < 0.0f;
          if (has_i && !rej && slot != last && !(ANYHIT && best >= 0)) {
            last = slot;
            if (n_ca < 4)
              ca_lo |= (unsigned long long)(unsigned)slot << (16 * n_ca);
            else
              ca_hi |= (unsigned long long)(unsigned)slot << (16 * (n_ca - 4));
            n_ca++;
          }
          if (__ballot(n_ca >= 8) != 0ULL) flush();  // (a lane's list is full: visit what everybody has so far)
        }
      }
      flush();
      if (ANYHIT && best >= 0) walking = false;
    }
    if (__ballot(far) != 0ULL) {  // (rare: see above) every sphere for the far lanes, one ball per turn
      for (int slot = 0; slot < ns; ++slot) {
        const bool cand = far && !(ANYHIT && best >= 0) && (wild || !(far1(bsx[slot], bsy[slot], bsz[slot], bsr[slot], ofx, ofy, ofz, hx, hy, hz, Ek0) < 0.0f));
        if (__ballot(cand) != 0ULL) visit2(slot, cand, 0, false);
      }
    }
    // planes, then done
    for (int k = ns; k < n; ++k) {
      pt_kdouble m = PT_KD(a.recs[k].invm);
      const double dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
      const double oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
      PT_PLANE_HIT(k);
    }
    return best;
  }
  for (int base = 0; base < ns; base += 64) {
    const int cnt = ns - base < 64 ? ns - base : 64;
    unsigned long long mask = 0ULL;
    unsigned gtouch = 0xffu;  // groups of the chunk some lane may touch (wave-uniform)
    const bool live = active && !(ANYHIT && best >= 0);
    const unsigned long long live_lanes = __ballot(live);
    if (live_lanes == 0ULL) continue;
    if (!ANYHIT && ns <= PT_SPARSE_MAX_SPHERES && __popcll(live_lanes) <= PT_SPARSE_RAYS) {
      // Few rays in flight (the deep stragglers of a round): turn the loop around.  Every lane holds ONE ball of
      // the chunk (coalesced load, once) and the rays take turns: a ray's constants are broadcast from its
      // lane, all 64 balls are tested at once, and the ballot IS that ray's candidate mask.  ~35 instructions
      // per ray and chunk instead of ~800 per chunk for the whole wave.
      const int sl = base + (threadIdx.x & 63);
#ifdef PT_DEBUG_TIME
      PT_VM_DRAIN();
      const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
#endif
      const float bx = ((const float *)a.bsoa)[sl], by = ((const float *)a.bsoa)[a.bs_stride + sl],
                  bz = ((const float *)a.bsoa)[2 * a.bs_stride + sl], br = ((const float *)a.bsoa)[3 * a.bs_stride + sl];
#ifdef PT_DEBUG_TIME
      asm volatile("s_waitcnt vmcnt(0)" : : "v"(bx), "v"(by), "v"(bz), "v"(br) : "memory");
      lat_note(3, __builtin_amdgcn_s_memtime() - lt0);
#endif
      const bool mine = (int)(threadIdx.x & 63) < cnt;
      unsigned long long todo = live_lanes;
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ULL;
#define PT_BCAST(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src))
        const bool rej = far1(bx, by, bz, br, PT_BCAST(ofx), PT_BCAST(ofy), PT_BCAST(ofz), PT_BCAST(hx), PT_BCAST(hy),
                              PT_BCAST(hz), PT_BCAST(Ek0)) < 0.0f;
#undef PT_BCAST
        const unsigned long long m = __ballot(mine && !rej);
        if ((int)(threadIdx.x & 63) == src) mask = m;
      }
    } else {
    if (levels) {
      const int c = base >> 6;
      const f2 nc = far2((f2){csx[c], csx[c]}, (f2){csy[c], csy[c]}, (f2){csz[c], csz[c]}, (f2){csr[c], csr[c]}, Ek2);
      if (__ballot(live && (wild || !(nc.x < 0.0f))) == 0ULL) continue;
      const f8 X = *(pt_kf8)(gsx + c * 8), Y = *(pt_kf8)(gsy + c * 8), Z = *(pt_kf8)(gsz + c * 8), R = *(pt_kf8)(gsr + c * 8);
      gtouch = 0u;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const f2 ng = far2((f2){X[k], X[k + 1]}, (f2){Y[k], Y[k + 1]}, (f2){Z[k], Z[k + 1]}, (f2){R[k], R[k + 1]}, Ek1);
        gtouch |= (__ballot(live && (wild || !(ng.x < 0.0f))) != 0ULL ? 1u << k : 0u) |
                  (__ballot(live && (wild || !(ng.y < 0.0f))) != 0ULL ? 2u << k : 0u);
      }
    }
    // eight spheres per round of scalar loads (the arrays are padded); a verdict is a sign bit, shifted into the word of
    // its 32 spheres (first sphere = highest bit: reversed below)
    unsigned rejw[2];
#pragma unroll
    for (int hw = 0; hw < 2; ++hw) {
      unsigned rej = 0u;
      for (int j = hw * 32; j < hw * 32 + 32; j += 8) {
        if (j >= cnt || !((gtouch >> (j >> 3)) & 1u)) {
          rej = (rej << 8) | 0xffu;
          continue;
        }
        const f8 X = *(pt_kf8)(bsx + base + j), Y = *(pt_kf8)(bsy + base + j), Z = *(pt_kf8)(bsz + base + j),
                 R = *(pt_kf8)(bsr + base + j);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const f2 nx = far2((f2){X[k], X[k + 1]}, (f2){Y[k], Y[k + 1]}, (f2){Z[k], Z[k + 1]}, (f2){R[k], R[k + 1]}, Ek0);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(nx.x), 31);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(nx.y), 31);
        }
      }
      rejw[hw] = rej;
    }
    mask = ~(((unsigned long long)__builtin_bitreverse32(rejw[1]) << 32) | (unsigned long long)__builtin_bitreverse32(rejw[0]));
    }
    if (wild) mask = ~0ULL;
    if (cnt < 64) mask &= (1ULL << cnt) - 1ULL;
    if (!live) mask = 0ULL;
#ifdef PT_DEBUG_TIME
    {
      const unsigned long long tn = __builtin_amdgcn_s_memtime();
      dbg_pre += tn - dbg_t0;
      dbg_t0 = tn;
    }
#endif
    // walk the mask, two candidates per turn (visit2)
    while (__ballot(mask != 0ULL) != 0ULL) {
      const bool has_a = mask != 0ULL;
      const int slot_a = base + (has_a ? __ffsll((long long)mask) - 1 : 0);
      mask &= mask - 1ULL;
      // (shadow rays stop at their first blocker and run at three waves per SIMD: one visit at a time there)
      const bool has_b = !ANYHIT && mask != 0ULL;
      const int slot_b = base + (has_b ? __ffsll((long long)mask) - 1 : 0);
      if (!ANYHIT) mask &= mask - 1ULL;
      visit2(slot_a, has_a, slot_b, has_b);
      if (ANYHIT && best >= 0) mask = 0ULL;  // this lane is blocked: nothing more to look at
#ifdef PT_DEBUG_TIME
      dbg_it++;
#endif
    }
#ifdef PT_DEBUG_TIME
    {
      const unsigned long long tn = __builtin_amdgcn_s_memtime();
      dbg_walk += tn - dbg_t0;
      dbg_t0 = tn;
    }
#endif
  }
#ifdef PT_DEBUG_TIME
  if ((threadIdx.x & 63) == 0) {  // per-wave partial sums (flushed once per kernel by pt_dbg_flush): no atomics in the hot loop
    unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
    wv[0] += dbg_pre;
    wv[1] += dbg_walk;
    wv[2] += dbg_it;
    wv[3] += 1ULL;
  }
#endif
  // ---- planes: shapes.py:168-175, only the z row of the object-space ray decides (wave-uniform loop) ----
  for (int k = ns; k < n; ++k) {
    pt_kdouble m = PT_KD(a.recs[k].invm);
    const double dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
    const double oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
    PT_PLANE_HIT(k);
  }
  return best;
}

// ---- the closest hit's HitRecord (shapes.py:123-131, 177-189; world.py:66-67) ----------------------
// Computed once per ray for the winner only; every value is a pure function of (ray, shape, t), so
// it equals what the reference computed for that candidate.
// INL: the transcendental functions inline (the latency-bound second pass of the path tracer, which has
// registers to spare) instead of behind a call (everything that runs at 4-5 waves per SIMD).
// (RP / AP: where the records live -- generic pointers into HBM, or address_space(3) pointers when the second
//  pass of the path tracer has staged the scene in LDS)
template <bool INL = false, typename RP = const PtShapeRec *, typename AP = const PtShapeAux *>
PT_DEV void hit_details(RP rec, AP ax, const Ray &r, double t, Hit &h, bool need_uv) {
  // rec / ax: the winner's records (same grouped slot in both tables)
  double im[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) im[k] = rec->invm[k];
  const V3 o = xf_point(im, r.o);
  const V3 d = xf_vec(im, r.d);
  const V3 hp = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};  // ray.py:52-57
  double fm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) fm[k] = ax->m[k];
  h.wp = xf_point(fm, hp);
  V3 nn;
  h.u = 0.0;
  h.v = 0.0;
  if (rec->kind == PT_SHAPE_SPHERE) {
    const bool keep = dot3(hp, d) < 0.0;  // shapes.py:45-54
    nn.x = keep ? hp.x : -hp.x;
    nn.y = keep ? hp.y : -hp.y;
    nn.z = keep ? hp.z : -hp.z;
    if (need_uv) {  // shapes.py:36-42
      const double uu = (INL ? atan2(hp.y, hp.x) : pt_atan2(hp.y, hp.x)) / (2.0 * PT_PI);
      h.u = (uu >= 0.0) ? uu : uu + 1.0;
      double z = hp.z;  // the reference raises ValueError outside [-1, 1]: clamp
      z = (z > 1.0) ? 1.0 : ((z < -1.0) ? -1.0 : z);
      h.v = (INL ? acos(z) : pt_acos(z)) / PT_PI;
    }
  } else {
    nn.x = 0.0;
    nn.y = 0.0;
    nn.z = (d.z < 0.0) ? 1.0 : -1.0;
    if (need_uv) {
      h.u = hp.x - floor(hp.x);
      h.v = hp.y - floor(hp.y);
    }
  }
  h.n = normalize3(xf_normal(im, nn));
}

// Out-of-line entry for the path tracer: its kernel keeps ~40 VGPRs of path state alive; inlining the
// HitRecord code (24 matrix doubles in flight) on top of that costs a wave of occupancy.
PT_NOINLINE void hit_details_call(const PtShapeRec *rec, const PtShapeAux *ax, const Ray *r, double t, Hit *h,
                                  bool need_uv) {
  hit_details(rec, ax, *r, t, *h, need_uv);
}

// ---- pigments (materials.py:50-100) --------------------------------------------------------------------
template <typename CP>
PT_DEV V3 pigment_color(const PtKArgs &a, int kind, CP c1, CP c2, double steps, int tex, double u, double v) {
  if (kind == PT_PIGMENT_IMAGE) {
    pt_kargs ca = cold_args(a);
    const PtTex *tx = ca->tex + tex;
    const int w = tx->w, hh = tx->h;
    long long col = (long long)(u * (double)w);  // int() truncates toward zero
    long long row = (long long)(v * (double)hh);
    if (col >= w) col = w - 1;
    if (row >= hh) row = hh - 1;
    const double *c = ca->tex_data + tx->offset + (row * w + col) * 3;
    V3 r = {c[0], c[1], c[2]};
    return r;
  }
  CP c = c1;
  if (kind == PT_PIGMENT_CHECKERED) {
    const long long iu = (long long)floor(u * steps);
    const long long iv = (long long)floor(v * steps);
    // Python's % 2 is non-negative; (x & 1) is the same parity for negative x in two's complement
    c = ((iu & 1LL) == (iv & 1LL)) ? c1 : c2;
  }
  V3 r = {c[0], c[1], c[2]};
  return r;
}
template <typename AP>
PT_DEV V3 brdf_pigment(const PtKArgs &a, AP ax, double u, double v) {
  return pigment_color(a, ax->pig_kind, &ax->pig_c1[0], &ax->pig_c2[0], ax->pig_steps, ax->pig_tex, u, v);
}
template <typename AP>
PT_DEV V3 emitted_pigment(const PtKArgs &a, AP ax, double u, double v) {
  return pigment_color(a, ax->emi_kind, &ax->emi_c1[0], &ax->emi_c2[0], ax->emi_steps, ax->emi_tex, u, v);
}

// ---- BRDF.scatter_ray (materials.py:132-152, 175-196; geometry.py:247-262) -------------------------
template <bool INL = false>
PT_DEV Ray scatter_ray(int brdf_kind, Pcg &pcg, V3 incoming, V3 point, V3 n) {
  Ray r;
  r.o = point;
  if (brdf_kind == PT_BRDF_DIFFUSE) {
    const double sign = (n.z > 0.0) ? 1.0 : -1.0;
    const double aa = -1.0 / (sign + n.z);
    const double bb = n.x * n.y * aa;
    const V3 e1 = {1.0 + sign * n.x * n.x * aa, sign * bb, -sign * n.x};
    const V3 e2 = {bb, sign + n.y * n.y * aa, -n.y};
    const double cts = pcg_float(pcg);
    const double ct = sqrt(cts), st = sqrt(1.0 - cts);
    const double phi = 2.0 * PT_PI * pcg_float(pcg);
    const double cp = INL ? cos(phi) : pt_cos(phi), sp = INL ? sin(phi) : pt_sin(phi);
    r.d.x = ct * (cp * e1.x) + ct * (sp * e2.x) + st * n.x;
    r.d.y = ct * (cp * e1.y) + ct * (sp * e2.y) + st * n.y;
    r.d.z = ct * (cp * e1.z) + ct * (sp * e2.z) + st * n.z;
    r.tmin = 1.0e-3;
  } else {
    const V3 rd = normalize3(incoming);
    const V3 nn = normalize3(n);
    const double dp = dot3(nn, rd);
    r.d.x = rd.x - dp * (2.0 * nn.x);
    r.d.y = rd.y - dp * (2.0 * nn.y);
    r.d.z = rd.z - dp * (2.0 * nn.z);
    r.tmin = 1e-5;
  }
  return r;
}

PT_NOINLINE void scatter_ray_call(int brdf_kind, Pcg *pcg, const V3 *incoming, const V3 *point, const V3 *n, Ray *out) {
  *out = scatter_ray(brdf_kind, *pcg, *incoming, *point, *n);
}

// ---- ImageTracer.fire_ray + Camera.fire_ray (imagetracer.py:48-58; camera.py:59-78, 103-124) -----
PT_DEV Ray primary_ray(const PtKArgs &a, int col, int row, double up, double vp) {
  pt_kargs c = cold_args(a);
  const double u = ((double)col + up) / (double)c->W;
  const double v = 1.0 - ((double)row + vp) / (double)c->H;
  V3 o, d;
  const double dist = c->cam_dist, aspect = c->cam_aspect;
  if (c->cam_kind == PT_CAMERA_PERSPECTIVE) {
    o.x = -dist;
    o.y = 0.0;
    o.z = 0.0;
    d.x = dist;
    d.y = (1.0 - 2.0 * u) * aspect;
    d.z = 2.0 * v - 1.0;
  } else {
    o.x = -1.0;
    o.y = (1.0 - 2.0 * u) * aspect;
    o.z = 2.0 * v - 1.0;
    d.x = 1.0;
    d.y = 0.0;
    d.z = 0.0;
  }
  Ray r;
  r.o = xf_point(c->cam_m, o);
  r.d = xf_vec(c->cam_m, d);
  r.tmin = 1.0e-5;
  return r;
}

// local (rank-compact) pixel index -> column and GLOBAL row (pt_params partition)
PT_DEV void pixel_coords(const PtKArgs &a, long long pix, int &col, int &grow) {
  pt_kargs c = cold_args(a);
  const int W = c->W, rb = c->row_block;
  const int lr = (int)(pix / W);
  col = (int)(pix - (long long)lr * W);
  const int blk = lr / rb;
  grow = (blk * c->n_ranks + c->rank) * rb + (lr - blk * rb);
}

// local (rank-compact) row -> GLOBAL row, 32-bit arithmetic only
PT_DEV int global_row(const PtKArgs &a, int lrow) {
  pt_kargs c = cold_args(a);
  const int nr = c->n_ranks;
  if (nr == 1) return lrow;
  const int rb = c->row_block;
  const int blk = lrow / rb;
  return (blk * nr + c->rank) * rb + (lrow - blk * rb);
}

// (f32: the output format, read once by the caller -- every read of the argument block is a scalar load of its own)
PT_DEV void store_pixel(const PtKArgs &a, long long pix, V3 v, bool f32) {
  if (f32) {
    float *o = (float *)a.out + pix * 3;
    o[0] = (float)v.x;
    o[1] = (float)v.y;
    o[2] = (float)v.z;
  } else {
    double *o = (double *)a.out + pix * 3;
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
  }
}
PT_DEV void store_pixel(const PtKArgs &a, long long pix, V3 v) {
  pt_kargs c = cold_args(a);
  if (c->out_f32) {
    float *o = (float *)a.out + pix * 3;
    o[0] = (float)v.x;
    o[1] = (float)v.y;
    o[2] = (float)v.z;
  } else {
    double *o = (double *)a.out + pix * 3;
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
  }
}

// Ray accounting without a contended atomic: wave reduction -> LDS -> one plain store per workgroup
// into a.ray_counter[blockIdx.x]; pt_sum_counts folds the per-workgroup partials afterwards.
// A partial carries two counts: all rays of the workgroup and, of those, the rays that were RESOLVED without
// being traced (tiles / pixels settled by the dome shortcut, pt_tile_kernel).
// first pass of the path tracer: a region with k flagged pixels -> F (queue[11]) and the histogram over k
// (queue[16 + k]) that pt_unit_scatter turns into the offsets of the work units
PT_DEV void note_flagged(unsigned long long *queue, int k) {
  atomicAdd(queue + 11, (unsigned long long)k);
  atomicAdd(queue + 16 + k, 1ULL);
}
PT_DEV void add_ray_count(const PtKArgs &a, unsigned long long n, int base = 0, unsigned long long resolved = 0, int block = -1) {
  if (block < 0) block = blockIdx.x;  // (a 2-D grid passes its linear workgroup index)
  unsigned long long *counter = cold_args(a)->ray_counter;
  if (counter) {
    __shared__ unsigned long long partial[2 * (PT_BLOCK / 64)];
    for (int off = 32; off > 0; off >>= 1) {
      n += __shfl_down(n, off, 64);
      resolved += __shfl_down(resolved, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      partial[threadIdx.x >> 6] = n;
      partial[PT_BLOCK / 64 + (threadIdx.x >> 6)] = resolved;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0, r = 0;
      for (int w = 0; w < PT_BLOCK / 64; ++w) {
        t += partial[w];
        r += partial[PT_BLOCK / 64 + w];
      }
      counter[2 * (base + block)] = t;
      counter[2 * (base + block) + 1] = r;
    }
  }
}

// partials: [n][2] (all rays, resolved rays) -> total[0], total[1]
__global__ void pt_sum_counts(const unsigned long long *partials, int n, unsigned long long *total) {
  __shared__ unsigned long long acc[2][256];
  unsigned long long t = 0, r = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    t += partials[2 * i];
    r += partials[2 * i + 1];
  }
  acc[0][threadIdx.x] = t;
  acc[1][threadIdx.x] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      acc[0][threadIdx.x] += acc[0][threadIdx.x + s];
      acc[1][threadIdx.x] += acc[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    total[0] = acc[0][0];
    total[1] = acc[1][0];
  }
}

// ---- pt_prep_hoist: per-shape constants of the primary rays (perspective camera) ----------------------
__global__ void pt_prep_hoist(const PtShapeRec *recs, PtHoist *hoist, PtHoistDiag *hoist_diag, int n,
                              int n_diag, V3 origin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 o = xf_point(recs[i].invm, origin);
  PtHoist h;
  h.ox = o.x;
  h.oy = o.y;
  h.oz = o.z;
  h.c = (o.x * o.x + o.y * o.y + o.z * o.z) - 1.0;
  hoist[i] = h;
  if (i < n_diag) {
    PtHoistDiag d;
    d.s[0] = recs[i].invm[0];
    d.s[1] = recs[i].invm[5];
    d.s[2] = recs[i].invm[10];
    d.o[0] = o.x;
    d.o[1] = o.y;
    d.o[2] = o.z;
    d.c = h.c;
    d._pad = 0.0;
    hoist_diag[i] = d;
  }
}

// ---- OnOff / Flat / PointLight: one world query per sample (+ shadow rays) ----------------------------
// ---- PointLightRenderer (render.py:157-193): ambient + emitted + the lights the hit point sees --------
// `bg` is what a miss returns; shadow rays handed to the world are counted in `nrays`.
PT_DEV V3 pointlight_shade(const PtKArgs &a, const Ray &ray, int hit, double best_t, bool active, V3 bg,
                           unsigned long long &nrays) {
  const V3 c = bg;
  const bool lit = active && hit >= 0;
  Hit h;
  h.wp = {0.0, 0.0, 0.0};
  h.n = {0.0, 0.0, 1.0};
  h.u = 0.0;
  h.v = 0.0;
  pt_kargs ca = cold_args(a);
  const PtShapeAux *ax = ca->aux + (hit >= 0 ? hit : 0);
  V3 res = c;
  if (lit) {
    hit_details(a.recs + hit, ax, ray, best_t, h, ax->needs_uv != 0);
    const V3 em = emitted_pigment(a, ax, h.u, h.v);
    res.x = ca->ambient[0] + em.x;
    res.y = ca->ambient[1] + em.y;
    res.z = ca->ambient[2] + em.z;
  }
  const int n_lights = ca->n_lights;
  const PtLight *lights = ca->lights;
  for (int l = 0; l < n_lights; ++l) {
    pt_kdouble L = PT_KD(&lights[l]);
    const V3 lp = {L[0], L[1], L[2]};
    // world.py:71-80: shadow ray from the hit point towards the light, any-hit in (1e-2/|d|, 1)
    Ray sh;
    sh.o = lit ? h.wp : lp;
    sh.d.x = lp.x - sh.o.x;
    sh.d.y = lp.y - sh.o.y;
    sh.d.z = lp.z - sh.o.z;
    const double dn = sqrt(sh.d.x * sh.d.x + sh.d.y * sh.d.y + sh.d.z * sh.d.z);
    sh.tmin = 1e-2 / dn;
    double tlim;
    const int blocked = world_query_lanes<true>(a, sh, 1.0, tlim, lit, -1);
    if (lit) nrays++;
    if (lit && blocked < 0) {
      const V3 dv = {h.wp.x - lp.x, h.wp.y - lp.y, h.wp.z - lp.z};
      const double dist = sqrt(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z);
      const double inv = 1.0 / dist;
      const V3 in_dir = {inv * dv.x, inv * dv.y, inv * dv.z};
      const V3 neg_in = {-in_dir.x, -in_dir.y, -in_dir.z};
      const double cos_theta = max2(0.0, dot3(normalize3(neg_in), normalize3(h.n)));
      const double lr = L[6];
      const double q = lr / dist;
      const double df = (lr > 0) ? q * q : 1.0;
      V3 bc = {0.0, 0.0, 0.0};
      if (ax->brdf_kind == PT_BRDF_DIFFUSE) {  // materials.py:129-130
        const V3 pc = brdf_pigment(a, ax, h.u, h.v);
        const double k = 1.0 / PT_PI;
        bc.x = pc.x * k;
        bc.y = pc.y * k;
        bc.z = pc.z * k;
      } else {  // materials.py:164-173
        const V3 out_dir = {-ray.d.x, -ray.d.y, -ray.d.z};
        const double th_in = pt_acos(dot3(normalize3(h.n), normalize3(in_dir)));
        const double th_out = pt_acos(dot3(normalize3(h.n), normalize3(out_dir)));
        if (fabs(th_in - th_out) < ax->brdf_param) bc = brdf_pigment(a, ax, h.u, h.v);
      }
      res.x = res.x + bc.x * L[3] * cos_theta * df;
      res.y = res.y + bc.y * L[4] * cos_theta * df;
      res.z = res.z + bc.z * L[5] * cos_theta * df;
    }
  }
  return res;
}

template <int RENDERER, bool HOIST>
__global__ __launch_bounds__(PT_BLOCK)
    __attribute__((amdgpu_waves_per_eu(RENDERER == PT_RENDERER_POINTLIGHT ? 3 : PT_WAVES_SIMPLE, 8))) void pt_simple_kernel(const PtKArgs a) {
  const int S = cold_args(a)->S;
  const int nsamp = S > 0 ? S * S : 1;
  unsigned long long nrays = 0;
  for (long long base = (long long)blockIdx.x * PT_BLOCK; base < a.npix; base += a.nthreads) {
    const long long pix = base + threadIdx.x;
    const bool active = pix < a.npix;
    int col = 0, grow = 0;
    if (active) pixel_coords(a, pix, col, grow);
    Pcg pcg;
    unsigned long long gpix = 0;
    if (S > 0) {
      pt_kargs c = cold_args(a);
      gpix = (unsigned long long)grow * c->W + col;
      if (c->pcg_mode == PT_PCG_PIXEL) pcg_seed(pcg, c->s0, c->q0 + gpix);
    }
    V3 cum = {0.0, 0.0, 0.0};
    for (int s = 0; s < nsamp; ++s) {
      double up = 0.5, vp = 0.5;
      if (S > 0) {  // imagetracer.py:86-93: u drawn first, then v; sub_row outer, sub_col inner
        pt_kargs c = cold_args(a);
        if (c->pcg_mode == PT_PCG_SAMPLE) pcg_seed(pcg, c->s0, c->q0 + gpix * (unsigned)nsamp + (unsigned)s);
        const int sr = s / S, sc = s - sr * S;
        up = ((double)sc + pcg_float(pcg)) / (double)S;
        vp = ((double)sr + pcg_float(pcg)) / (double)S;
      }
      const Ray ray = primary_ray(a, col, grow, up, vp);
      double best_t;
      const int hit = world_query<RENDERER == PT_RENDERER_ONOFF, HOIST>(a, ray, INFINITY, best_t, active);
      if (active) nrays++;
      V3 c;
      {
        pt_kargs ca = cold_args(a);
        c.x = ca->bg[0];
        c.y = ca->bg[1];
        c.z = ca->bg[2];
      }
      if (RENDERER == PT_RENDERER_ONOFF) {  // render.py:52-53
        if (hit >= 0) {
          pt_kargs ca = cold_args(a);
          c.x = ca->onoff[0];
          c.y = ca->onoff[1];
          c.z = ca->onoff[2];
        }
      } else if (RENDERER == PT_RENDERER_FLAT) {  // render.py:65-74
        if (hit >= 0) {
          const PtShapeAux *ax = cold_args(a)->aux + hit;
          Hit h;
          h.u = 0.0;
          h.v = 0.0;
          // Flat needs only (u, v); skip the whole HitRecord when both pigments are uniform
          if (ax->needs_uv) hit_details(a.recs + hit, ax, ray, best_t, h, true);
          const V3 p1 = brdf_pigment(a, ax, h.u, h.v);
          const V3 p2 = emitted_pigment(a, ax, h.u, h.v);
          c.x = p1.x + p2.x;
          c.y = p1.y + p2.y;
          c.z = p1.z + p2.z;
        }
      } else {  // PointLight, render.py:157-193
        c = pointlight_shade(a, ray, hit, best_t, active, c, nrays);
      }
      if (S > 0) {
        cum.x = cum.x + c.x;
        cum.y = cum.y + c.y;
        cum.z = cum.z + c.z;
      } else {
        cum = c;
      }
    }
    if (S > 0) {  // imagetracer.py:99-101
      const double k = 1.0 / (double)(S * S);
      cum.x = cum.x * k;
      cum.y = cum.y * k;
      cum.z = cum.z * k;
    }
    if (active) store_pixel(a, pix, cum);
  }
#ifdef PT_DEBUG_TIME
  pt_dbg_flush();
#endif
  add_ray_count(a, nrays);
}

// ---- tile culling for primary rays -------------------------------------------------------------------
// A wave owns an 8x8-pixel tile.  All its primary rays (every jittered sample of every pixel) lie in
// the convex cone spanned by the pixels' corner rays, so a shape whose bounding sphere misses that
// cone (with a 1e-6 relative margin, ~1e9 times the rounding error of the fp64 test) cannot yield
// delta > 0 for any lane: skipping it cannot change a single bit of the result.  Each lane tests one
// bounding sphere per pass; the survivors come back as one 64-bit ballot per pass, staged in LDS
// (wave-private slice) and replayed for every sample.  Survivors run the exact reference arithmetic
// in ascending slot order; ties go to the lower World.shapes index as everywhere else.
// The cone and the rejection test run in fp32 (sqrt/rcp are single instructions there) with explicit
// conservative margins: every rounding error of the fp32 evaluation (<~1e-6 relative, plus the
// absolute error of C - O for large coordinates) is covered by widening the cone by 2e-6 in cos and
// the test by 1e-5*L + eps_abs.  The margin only ever KEEPS more shapes; it never touches the exact
// fp64 arithmetic the survivors go through.
struct TileCone {
  float ox, oy, oz;  // apex
  float ax, ay, az;  // unit axis
  float cos_t, sin_t;
  float oabs;        // max |apex component| (error scale of C - O)
  bool all;          // wide cone / degenerate: keep everything
  float kx, ky, kz;  // THIS lane's corner direction (corner lane & 3), un-normalised
  float dmax2;       // upper bound of |d|^2 over the tile (|d|^2 is convex: max at a corner)
  float dmin;        // lower bound of |d| over the tile (axis . d is affine: min at a corner)
  float rbeam;       // orthogonal camera: the tile's rays fill a beam of this radius around the axis line
  bool ortho;        // ... then (ox, oy, oz) is a point of that line, cos_t = 1, sin_t = 0, and
                     // (kx, ky, kz) is this lane's corner ORIGIN
};

// Rows of the tile are [grow0, grow1] (global image rows, inclusive), columns [x0, x1): the tile's
// pixels (all jitter samples included) lie inside the rectangle [x0, x1] x [grow0, grow1 + 1] of the
// image plane; primary directions are affine in image position, so the convex cone spanned by the
// four corner rays contains every ray of the tile.
// the host folded camera.py:116-124 and imagetracer.py:56-58 into d(x, y) = d0 + x*dx + y*dy (fp32)
// Orthogonal camera (camera.py:59-78): the roles swap -- the ORIGIN is affine in the image position,
// o(x, y) = d0 + x*dx + y*dy, and `apex` holds the common direction.
struct ConeCam {
  float d0[3], dx[3], dy[3], apex[3];
  bool ortho;
};

PT_DEV ConeCam cone_cam(const PtKArgs &a) {
  ConeCam k;
  pt_kargs c = cold_args(a);
  for (int i = 0; i < 3; ++i) {
    k.d0[i] = c->cone_d0[i];
    k.dx[i] = c->cone_dx[i];
    k.dy[i] = c->cone_dy[i];
    k.apex[i] = c->cone_apex[i];
  }
  k.ortho = c->cam_kind != PT_CAMERA_PERSPECTIVE;
  return k;
}

PT_DEV TileCone tile_cone(const ConeCam &k, int x0, int x1, int grow0, int grow1) {
  TileCone tc;
  const int lane = threadIdx.x & 63;
  const float d0x = k.d0[0], d0y = k.d0[1], d0z = k.d0[2];
  const float dxx = k.dx[0], dxy = k.dx[1], dxz = k.dx[2];
  const float dyx = k.dy[0], dyy = k.dy[1], dyz = k.dy[2];
  tc.ortho = k.ortho;
  tc.rbeam = 0.0f;
  if (k.ortho) {
    // Parallel rays: every ray of the tile starts inside the parallelogram spanned by the four corner
    // origins and runs along the common direction, i.e. inside the cylinder around the line through
    // the parallelogram's centre whose radius is the largest corner distance from that line (the
    // distance is convex in the image position).  cone_keeps() treats it as a cone with t = 0 whose
    // spheres are widened by rbeam.  Spheres behind the image plane are simply kept.
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)grow0, fy1 = (float)(grow1 + 1);
    const float xm = 0.5f * (fx0 + fx1), ym = 0.5f * (fy0 + fy1);
    tc.ox = d0x + xm * dxx + ym * dyx;
    tc.oy = d0y + xm * dxy + ym * dyy;
    tc.oz = d0z + xm * dxz + ym * dyz;
    const float xk = (lane & 1) ? fx1 : fx0, yk = (lane & 2) ? fy1 : fy0;
    tc.kx = d0x + xk * dxx + yk * dyx;
    tc.ky = d0y + xk * dxy + yk * dyy;
    tc.kz = d0z + xk * dxz + yk * dyz;
    const float rd = __frsqrt_rn(k.apex[0] * k.apex[0] + k.apex[1] * k.apex[1] + k.apex[2] * k.apex[2]);
    tc.ax = k.apex[0] * rd;
    tc.ay = k.apex[1] * rd;
    tc.az = k.apex[2] * rd;
    const float ex = tc.kx - tc.ox, ey = tc.ky - tc.oy, ez = tc.kz - tc.oz;
    const float ep = ex * tc.ax + ey * tc.ay + ez * tc.az;
    const float px = ex - ep * tc.ax, py = ey - ep * tc.ay, pz = ez - ep * tc.az;
    float rb = __fsqrt_rn(px * px + py * py + pz * pz);
    rb = fmaxf(rb, __shfl_xor(rb, 1, 64));
    rb = fmaxf(rb, __shfl_xor(rb, 2, 64));
    const float kabs = fmaxf(fmaxf(fabsf(tc.kx), fabsf(tc.ky)), fabsf(tc.kz));
    tc.oabs = fmaxf(fmaxf(fmaxf(fabsf(tc.ox), fabsf(tc.oy)), fabsf(tc.oz)), kabs);
    tc.rbeam = rb * (1.0f + 1e-4f) + 4e-6f * tc.oabs;  // fp32 model of the origins: ~3e-7 relative each
    tc.cos_t = 1.0f;
    tc.sin_t = 0.0f;
    tc.dmax2 = 0.0f;
    tc.dmin = 0.0f;  // (no dome shortcut for parallel rays)
    tc.all = !(rd > 0.0f) || !(tc.rbeam >= 0.0f);  // degenerate direction or NaN: keep everything
    return tc;
  }
  tc.ox = k.apex[0];
  tc.oy = k.apex[1];
  tc.oz = k.apex[2];
  tc.oabs = fmaxf(fmaxf(fabsf(tc.ox), fabsf(tc.oy)), fabsf(tc.oz));
  const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)grow0, fy1 = (float)(grow1 + 1);
  const float xm = 0.5f * (fx0 + fx1), ym = 0.5f * (fy0 + fy1);
  const float cx = d0x + xm * dxx + ym * dyx, cy = d0y + xm * dxy + ym * dyy, cz = d0z + xm * dxz + ym * dyz;
  // lane k computes corner k & 3; the min over lanes 0..3 is the min over the whole wave
  const float xk = (lane & 1) ? fx1 : fx0, yk = (lane & 2) ? fy1 : fy0;
  const float kx = d0x + xk * dxx + yk * dyx, ky = d0y + xk * dxy + yk * dyy, kz = d0z + xk * dxz + yk * dyz;
  const float rc = __frsqrt_rn(cx * cx + cy * cy + cz * cz);
  tc.ax = cx * rc;
  tc.ay = cy * rc;
  tc.az = cz * rc;
  const float rk = __frsqrt_rn(kx * kx + ky * ky + kz * kz);
  float cs = (tc.ax * kx + tc.ay * ky + tc.az * kz) * rk;
  cs = fminf(cs, __shfl_xor(cs, 1, 64));
  cs = fminf(cs, __shfl_xor(cs, 2, 64));
  cs -= 4e-6f;  // fp32 evaluation of the directions (~3e-7 relative) + of the dot product
  tc.kx = kx;
  tc.ky = ky;
  tc.kz = kz;
  float k2 = kx * kx + ky * ky + kz * kz, pj = tc.ax * kx + tc.ay * ky + tc.az * kz;
  k2 = fmaxf(k2, __shfl_xor(k2, 1, 64));
  k2 = fmaxf(k2, __shfl_xor(k2, 2, 64));
  pj = fminf(pj, __shfl_xor(pj, 1, 64));
  pj = fminf(pj, __shfl_xor(pj, 2, 64));
  tc.dmax2 = k2 * (1.0f + 1e-5f);
  tc.dmin = pj * (1.0f - 1e-5f);
  tc.all = !(cs > 0.05f);  // also catches NaN
  tc.cos_t = cs;
  tc.sin_t = __fsqrt_rn(fmaxf(0.0f, 1.0f - cs * cs)) * (1.0f + 1e-5f) + 1e-7f;
  return tc;
}

PT_DEV TileCone tile_cone(const PtKArgs &a, int x0, int x1, int grow0, int grow1) {
  return tile_cone(cone_cam(a), x0, x1, grow0, grow1);
}

// The cone of ONE pixel's primary rays (perspective camera), computed by every lane for its own pixel
// (x, global row grow): all jittered rays of the pixel pass through [x, x+1] x [grow, grow+1] of the image
// plane and directions are affine in the image position, so they lie in the circular cone around the pixel
// centre's direction whose half-angle is the largest of the four corner angles.  A pixel's cone is ~1e-3 rad
// wide or less, where 1 - cos is below fp32 resolution: the opening is taken from the SINE, |axis x k| / |k|
// (relative error ~1e-6), widened by 1e-5 relative + 3e-6 absolute for the fp32 model of the directions
// (~3e-7 relative, the same model tile_cone uses) and the evaluation; cos t only scales `perp` in cone_keeps
// and is rounded down.  Apex and error scale come from the tile's cone.
PT_DEV TileCone pixel_cone(const ConeCam &k, const TileCone &tile, int x, int grow) {
  TileCone pc = tile;
  const float fx0 = (float)x, fx1 = (float)(x + 1), fy0 = (float)grow, fy1 = (float)(grow + 1);
  const float xm = 0.5f * (fx0 + fx1), ym = 0.5f * (fy0 + fy1);
  const float cx = k.d0[0] + xm * k.dx[0] + ym * k.dy[0], cy = k.d0[1] + xm * k.dx[1] + ym * k.dy[1],
              cz = k.d0[2] + xm * k.dx[2] + ym * k.dy[2];
  const float rc = __frsqrt_rn(cx * cx + cy * cy + cz * cz);
  pc.ax = cx * rc;
  pc.ay = cy * rc;
  pc.az = cz * rc;
  float sn = 0.0f, cs = 1.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float xk = (q & 1) ? fx1 : fx0, yk = (q & 2) ? fy1 : fy0;
    const float kx = k.d0[0] + xk * k.dx[0] + yk * k.dy[0], ky = k.d0[1] + xk * k.dx[1] + yk * k.dy[1],
                kz = k.d0[2] + xk * k.dx[2] + yk * k.dy[2];
    const float rk = __frsqrt_rn(kx * kx + ky * ky + kz * kz);
    const float wx = pc.ay * kz - pc.az * ky, wy = pc.az * kx - pc.ax * kz, wz = pc.ax * ky - pc.ay * kx;
    sn = fmaxf(sn, __fsqrt_rn(wx * wx + wy * wy + wz * wz) * rk);
    cs = fminf(cs, (pc.ax * kx + pc.ay * ky + pc.az * kz) * rk);
  }
  pc.cos_t = cs - 4e-6f;
  pc.sin_t = sn * (1.0f + 1e-5f) + 3e-6f;
  pc.rbeam = 0.0f;
  pc.ortho = false;
  pc.all = tile.all || !(pc.cos_t > 0.05f) || !(pc.sin_t < 0.5f);  // also NaN
  return pc;
}

// may the bounding sphere touch the cone?  (conservative: true when in doubt)
// In the half-plane (d, perp) = (distance along the axis, distance from the axis) the solid cone lies
// on the side q <= 0 of the line through the apex with direction (cos t, sin t), where
// q = perp*cos t - d*sin t; a point with q > 0 is at least q away from every point of the cone (also
// behind the apex, where the true distance |v| is larger still).  So q > R proves a miss.  perp is
// taken from the rejection vector v - d*axis (no cancellation between squares): the fp32 error of q
// is a few 1e-7*|v| plus the error of C - O; behind the apex (d < 0) the deliberately enlarged sin t
// adds up to 1.02e-5*|d|.  The margin is 4e-5*(|d| + perp) + 3*eps_abs.
PT_DEV bool cone_keeps(const TileCone &tc, float4 b) {
  if (tc.all || !(b.w >= 0.0f)) return true;
  const float vx = b.x - tc.ox, vy = b.y - tc.oy, vz = b.z - tc.oz;
  const float eps_abs = 1e-6f * (fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fabsf(b.z)) + tc.oabs);
  const float d = vx * tc.ax + vy * tc.ay + vz * tc.az;
  const float wx = vx - d * tc.ax, wy = vy - d * tc.ay, wz = vz - d * tc.az;
  const float perp = __builtin_amdgcn_sqrtf(wx * wx + wy * wy + wz * wz);  // v_sqrt_f32, 1 ulp
  const float q = perp * tc.cos_t - d * tc.sin_t;
  const float R = b.w * (1.0f + 1e-5f) + 4e-5f * (fabsf(d) + perp) + 3.0f * eps_abs + tc.rbeam;
  return !(q > R);  // also keeps NaN
}

// May a plane be hit by some ray of the tile?  (conservative: true when in doubt.)  shapes.py:168-175
// hits only when t = -o'.z / d'.z is positive, i.e. when o'.z and d'.z have opposite signs.  o'.z is
// the same for every primary ray (its sign is taken from an fp32 evaluation, and only when the value
// is 1e-4 clear of zero relative to its terms); d'.z = row2(invm) . d is affine in the pixel
// position, so if it has the sign of o'.z -- by a margin of 1e-4 |row2| |d|, ~100 times the fp32
// error of this evaluation -- at the four corner directions it has that sign for every ray of the
// tile and none of them can hit.  Called by the whole wave (it gathers the corners from lanes 0..3).
PT_DEV bool plane_keeps(const TileCone &tc, float4 b, bool isplane) {
  // b = (row2(invm) as fp32, invm[11] as fp32): the plane slots of the bounds table (pt_scene_upload)
  float cxs[4], cys[4], czs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    cxs[q] = __shfl(tc.kx, q, 64);
    cys[q] = __shfl(tc.ky, q, 64);
    czs[q] = __shfl(tc.kz, q, 64);
  }
  if (!isplane || tc.all) return true;
  const float rx = b.x, ry = b.y, rz = b.z;
  const float rn = __fsqrt_rn(rx * rx + ry * ry + rz * rz);
  if (tc.ortho) {
    // parallel rays: d'.z = row . d is one number for the whole frame, o'.z = row . o + invm[11] is
    // affine in the image position; no hit anywhere in the tile when o'.z has the sign of d'.z at the
    // four corner origins (same margins)
    const float dz = rx * tc.ax + ry * tc.ay + rz * tc.az;  // (along the unit direction: only the sign matters)
    if (!(fabsf(dz) > 1e-4f * rn)) return true;
    bool away = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float oz = rx * cxs[q] + ry * cys[q] + rz * czs[q] + b.w;
      const float thr = 1e-4f * (rn * (fabsf(cxs[q]) + fabsf(cys[q]) + fabsf(czs[q])) + fabsf(b.w));
      away = away && ((dz > 0.0f) ? (oz > thr) : (oz < -thr));
    }
    return !away;
  }
  // the sign of o'.z from fp32: trusted only when |o'.z| stands clear of the rounding (else keep)
  const float oz = rx * tc.ox + ry * tc.oy + rz * tc.oz + b.w;
  if (!(fabsf(oz) > 1e-4f * (2.0f * rn * tc.oabs + fabsf(b.w)))) return true;  // also NaN
  bool away = true;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float dz = rx * cxs[q] + ry * cys[q] + rz * czs[q];
    const float thr = 1e-4f * rn * __fsqrt_rn(cxs[q] * cxs[q] + cys[q] * cys[q] + czs[q] * czs[q]);
    away = away && ((oz > 0.0f) ? (dz > thr) : (dz < -thr));  // NaN: false
  }
  return !away;
}

// The survivor masks live in LDS and are always addressed through this array (never through a generic
// pointer): DS reads and writes of one wave execute in order, FLAT accesses to the LDS aperture do not.
extern __shared__ unsigned long long pt_lds_masks[];

// HIER: the mask bits index the tile's cell list (pt_cell_kernel), which holds the slots.
// HOISTED = false (orthogonal camera: no common origin): the object-space origin is computed per ray.
template <bool ANYHIT, bool HIER = false, bool HOISTED = true>
PT_DEV int world_query_tile(const PtKArgs &a, const Ray &r, int mbase, int npass, double &best_t, bool active,
                            const unsigned int *list = nullptr) {
  int best = -1;
  best_t = INFINITY;
  const double tmin = r.tmin, tmax = INFINITY;
  const int nd = a.n_diag, ns = a.n_spheres;
  const WaveGuard g = wave_guard<HOISTED>(r, active);
  for (int p = 0; p < npass; ++p) {
    const unsigned long long mv = pt_lds_masks[mbase + p];
    // readfirstlane returns a signed int: go through unsigned or bit 31 smears over the high half
    const unsigned m_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mv >> 32));
    const unsigned m_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mv);
    unsigned long long mask = ((unsigned long long)m_hi << 32) | (unsigned long long)m_lo;
    while (mask) {
      const int idx = p * 64 + (__ffsll((long long)mask) - 1);
      mask &= mask - 1;
      const int slot = HIER ? PT_KI(list)[idx] : idx;
      if (slot < ns) {
        double dx, dy, dz, ox, oy, oz, cc;
        if (HOISTED) {
          if (slot < nd && g.fast) {
            pt_kdouble h = PT_KD(&a.hoist_diag[slot]);
            dx = r.d.x * h[0];
            dy = r.d.y * h[1];
            dz = r.d.z * h[2];
            ox = h[3];
            oy = h[4];
            oz = h[5];
            cc = h[6];
          } else {
            pt_kdouble m = PT_KD(a.recs[slot].invm);
            pt_kdouble h = PT_KD(&a.hoist[slot]);
            dx = r.d.x * m[0] + r.d.y * m[1] + r.d.z * m[2];
            dy = r.d.x * m[4] + r.d.y * m[5] + r.d.z * m[6];
            dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
            ox = h[0];
            oy = h[1];
            oz = h[2];
            cc = h[3];
          }
        } else {
          bool done = false;
          if (slot < nd && g.fast) {
            pt_kdouble h = PT_KD(&a.diag[slot]);
            if ((g.ozmask & ~(unsigned)*PT_KI(&a.diag[slot].tnz)) == 0u) {  // (see world_query)
              dx = r.d.x * h[0];
              dy = r.d.y * h[1];
              dz = r.d.z * h[2];
              ox = r.o.x * h[0] + h[3];
              oy = r.o.y * h[1] + h[4];
              oz = r.o.z * h[2] + h[5];
              done = true;
            }
          }
          if (!done) {
            pt_kdouble m = PT_KD(a.recs[slot].invm);
            dx = r.d.x * m[0] + r.d.y * m[1] + r.d.z * m[2];
            dy = r.d.x * m[4] + r.d.y * m[5] + r.d.z * m[6];
            dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
            ox = r.o.x * m[0] + r.o.y * m[1] + r.o.z * m[2] + m[3];
            oy = r.o.x * m[4] + r.o.y * m[5] + r.o.z * m[6] + m[7];
            oz = r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
          }
          cc = (ox * ox + oy * oy + oz * oz) - 1.0;
        }
        const double aa = dx * dx + dy * dy + dz * dz;
        PT_SPHERE_ROOTS(slot);
      } else {
        pt_kdouble m = PT_KD(a.recs[slot].invm);
        const double dz = r.d.x * m[8] + r.d.y * m[9] + r.d.z * m[10];
        const double oz = HOISTED ? PT_KD(&a.hoist[slot])[2] : r.o.x * m[8] + r.o.y * m[9] + r.o.z * m[10] + m[11];
        PT_PLANE_HIT(slot);
      }
      PT_ANYHIT_EXIT();
    }
  }
  return best;
}

// Large scenes: a pre-pass culls the world once per PT_CELL x PT_CELL block of GLOBAL image pixels
// (same cone test, same margins) into a slot list per cell; a tile then only looks at its cell's
// list.  A workgroup takes a 2x2 group of cells and one chunk of at most PT_CELL_CHUNK shapes (each
// bounding sphere is loaded once for four cells), collects the survivors in LDS (LDS atomics: global
// round trips would serialise the passes) and appends them to the cells' lists with one global atomic
// per cell.  A list is therefore in no particular order -- which cannot matter: the exact tests pick
// the closest hit, ties by World.shapes index.  cell_count is zeroed before the launch.
#ifndef PT_CELL
#define PT_CELL 32
#endif
#define PT_CELL_CHUNK 2048
__global__ __launch_bounds__(PT_BLOCK) void pt_cell_kernel(const PtKArgs a, int nchunks, int chunk_len) {
  __shared__ unsigned short found[4][PT_CELL_CHUNK];  // offsets from the chunk's first slot
  __shared__ int nfound[4], gbase[4];
  int W, H;
  {
    pt_kargs c = cold_args(a);
    W = c->W;
    H = c->H;
  }
  const ConeCam cam = cone_cam(a);
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.x / nchunks, chunk = blockIdx.x - group * nchunks;
  const int groups_x = (a.cells_x + 1) >> 1;
  const int gy = group / groups_x, gx = group - gy * groups_x;
  const int cells_y = (H + PT_CELL - 1) / PT_CELL;
  const int gx1 = (gx + 1) * 2 * PT_CELL < W ? (gx + 1) * 2 * PT_CELL : W;
  const int gr1 = (gy + 1) * 2 * PT_CELL - 1 < H - 1 ? (gy + 1) * 2 * PT_CELL - 1 : H - 1;
  const TileCone tg = tile_cone(cam, gx * 2 * PT_CELL, gx1, gy * 2 * PT_CELL, gr1);
  TileCone tc[4];
  int cell[4];
  for (int k = 0; k < 4; ++k) {
    const int cx = gx * 2 + (k & 1), cy = gy * 2 + (k >> 1);
    cell[k] = (cx < a.cells_x && cy < cells_y) ? cy * a.cells_x + cx : -1;
    const int ccx = cx < a.cells_x ? cx : a.cells_x - 1, ccy = cy < cells_y ? cy : cells_y - 1;
    const int x1 = (ccx + 1) * PT_CELL < W ? (ccx + 1) * PT_CELL : W;
    const int r1 = (ccy + 1) * PT_CELL - 1 < H - 1 ? (ccy + 1) * PT_CELL - 1 : H - 1;
    tc[k] = tile_cone(cam, ccx * PT_CELL, x1, ccy * PT_CELL, r1);
  }
  if (threadIdx.x < 4) nfound[threadIdx.x] = 0;
  __syncthreads();
  const int n = a.n_shapes;
  const int s0 = chunk * chunk_len, s1 = s0 + chunk_len < n ? s0 + chunk_len : n;
  float4 b_next = a.bounds[s0 + (int)threadIdx.x < s1 ? s0 + (int)threadIdx.x : 0];
  for (int p0 = s0; p0 < s1; p0 += PT_BLOCK) {
    const int slot = p0 + (int)threadIdx.x;
    const bool in = slot < s1;
    const float4 b = b_next;
    b_next = a.bounds[slot + PT_BLOCK < s1 ? slot + PT_BLOCK : 0];
    if (a.bs_levels) {
      // Morton-ordered slots: this wave's 64 slots are one chunk of the ball hierarchy (pt_scene_upload);
      // a chunk whose ball misses the group cone has nothing for any of the four cells
      const int cb = __builtin_amdgcn_readfirstlane(slot) >> 6;
      if ((cb + 1) * 64 <= a.n_spheres) {
        typedef const __attribute__((address_space(4))) float *pt_kfloat;
        pt_kfloat cs = (pt_kfloat)(const void *)a.bsoa + 4 * (a.bs_stride + a.gs_stride);
        const float4 ball = {cs[cb], cs[a.cs_stride + cb], cs[2 * a.cs_stride + cb], cs[3 * a.cs_stride + cb]};
        if (!cone_keeps(tg, ball)) continue;
      }
    }
    const bool isplane = slot >= a.n_spheres;  // planes carry no bounding sphere: every cell keeps them
    if (!__ballot(in && (isplane || cone_keeps(tg, b)))) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (cell[k] < 0) continue;
      const unsigned long long m = __ballot(in && (isplane || cone_keeps(tc[k], b)));
      if (!m) continue;
      int base = 0;
      if (lane == 0) base = atomicAdd(&nfound[k], __popcll(m));  // ds_add_rtn
      base = __builtin_amdgcn_readfirstlane(base);
      if ((m >> lane) & 1ull) found[k][base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(slot - s0);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    gbase[k] = (cell[k] >= 0 && nfound[k] > 0) ? atomicAdd(a.cell_count + cell[k], nfound[k]) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (cell[k] < 0) continue;
    unsigned int *dst = a.cell_list + (size_t)cell[k] * a.cell_stride + gbase[k];
    for (int i = threadIdx.x; i < nfound[k]; i += PT_BLOCK) dst[i] = (unsigned)(s0 + found[k][i]);
  }
}

// OnOff / Flat / PointLight with a perspective camera: 8x8 tiles, culled shape lists.
// WAVES = waves per SIMD the register allocator must make room for.  With the transcendental
// functions out of line the Flat kernel needs 93 VGPRs: 5 waves per SIMD, no scratch.
// HIER (large scenes): the tile culls its 32x32 cell's survivor list instead of the whole world.
//
// RENDERER == PATHTRACER is the path tracer's first pass.  A sample whose primary ray misses, or hits
// a surface whose BRDF pigment is black (hit_color_lum == 0: render.py:126 spawns nothing), ends at
// depth 0 with radiance = background resp. emitted + 0*(1/N) and has drawn nothing but its two jitter
// numbers -- exactly what this loop does.  A pixel all of whose samples end like that (sky, lamps) is
// finished here at Flat speed; a pixel that meets anything else is abandoned (nothing stored, its
// rays not counted) and flagged in region_mask for pt_path_kernel, which renders it from its seed.
//
// ORTHO: orthogonal camera -- the tile's rays fill a beam instead of a cone (tile_cone), nothing is
// hoisted (HOISTED = false queries), no dome shortcut.
// BLOCKS (path tracer's first pass on big frames; chosen by the host): four strips at a time, see below.
template <int RENDERER, int WAVES, bool HIER, bool ORTHO = false, bool BLOCKS = false>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void pt_tile_kernel(const PtKArgs a, int count_base) {
  int S, W, rows_local, npass, dome_slot;
  bool dome_on, out_f32;
  unsigned long long *rmask = nullptr;  // path tracer's first pass: [region] flagged pixels, and their number
  unsigned char *rkeys = nullptr;
  {
    pt_kargs c = cold_args(a);
    out_f32 = c->out_f32 != 0;
    S = c->S;
    if (RENDERER == PT_RENDERER_PATHTRACER) {  // (read once: a scalar load per tile otherwise, in front of every sky tile's two stores)
      rmask = c->region_mask;
      rkeys = c->region_keys;
    }
    W = c->W;
    rows_local = c->rows_local;
    npass = c->npass;
    dome_slot = c->dome_slot;  // -1: the camera is inside no sphere with uniform pigments
    dome_on = c->dome_shortcut != 0;
  }
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int mbase = wib * npass;  // this wave's slice of pt_lds_masks
  const int nsamp = S > 0 ? S * S : 1;
  const int tiles_x = (W + 7) >> 3, tiles_y = (rows_local + 7) >> 3;
  const int ntiles = tiles_x * tiles_y;
  const int nwaves = gridDim.x * (PT_BLOCK / 64);
  unsigned long long nrays = 0, nres = 0;  // rays accounted for; of those, resolved by the dome shortcut
#ifdef PT_DEBUG_TIME
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define PT_TSTAMP(k) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[k] += tn - tprev; tprev = tn; } while (0)
#else
#define PT_TSTAMP(k) do { } while (0)
#endif
  // The value of a pixel all of whose primary rays are certain to end on sphere `only` with the camera well
  // inside it (hoisted c = hc_ < -0.5; see the comment at the per-tile check below): false when the conditions
  // do not hold for a tile whose directions are bounded by dmax2 / dmin; else `cum` is the pixel (the S*S
  // additions and the final scaling of imagetracer.py:83-101 replayed) and `settled` says whether it is final
  // (PointLight needs the hit point; a path tracer whose dome scatters light goes to the second pass).
  // The shape-dependent half of that (records, pigments, the replayed sum) is the same for every tile that meets
  // the same sphere -- in practice ONE sky dome per frame: it is worked out once per wave and kept (dc_*), so that
  // a dome tile or strip costs no dependent loads, only the few comparisons below.
  // (Path tracer's first pass only: there strips and dome tiles are nearly all of the work and the kernel runs at
  //  four waves per SIMD anyway; OnOff / Flat would pay for the ~25 registers with their fifth wave: C2 +8 %.)
  constexpr bool KEEP = RENDERER == PT_RENDERER_PATHTRACER;
  int dc_slot = -1;
  bool dc_usable = false, dc_settled = false;
  float dc_fro2 = 0.0f;
  double dc_hc = 0.0;  // perspective camera: the hoisted c = |o'|^2 - 1 of the sphere
  V3 dc_cum = {0.0, 0.0, 0.0};
  auto dome_prepare = [&](int only) {  // (`only` wave-uniform)
    if (KEEP && only == dc_slot) return;
    dc_slot = only;
    pt_kargs ca = cold_args(a);
    const PtShapeAux *ax = ca->aux + only;
    dc_fro2 = (float)PT_KD(&a.recs[only])[13];  // PtShapeRec::fro2
    dc_hc = ORTHO ? 0.0 : ((only < a.n_diag) ? PT_KD(&a.hoist_diag[only])[6] : PT_KD(&a.hoist[only])[3]);
    dc_settled = false;
    dc_cum = {0.0, 0.0, 0.0};
    dc_usable = RENDERER != PT_RENDERER_POINTLIGHT && ax->needs_uv == 0;
    if (!dc_usable) return;
    V3 c;
    dc_settled = true;
    if (RENDERER == PT_RENDERER_ONOFF) {
      c.x = ca->onoff[0];
      c.y = ca->onoff[1];
      c.z = ca->onoff[2];
    } else if (RENDERER == PT_RENDERER_FLAT) {
      const V3 p1 = brdf_pigment(a, ax, 0.0, 0.0), p2 = emitted_pigment(a, ax, 0.0, 0.0);
      c.x = p1.x + p2.x;
      c.y = p1.y + p2.y;
      c.z = p1.z + p2.z;
    } else {
      const V3 hc = brdf_pigment(a, ax, 0.0, 0.0), em = emitted_pigment(a, ax, 0.0, 0.0);
      const double lum = max2(max2(hc.x, hc.y), hc.z);
      dc_settled = !(ca->rr <= 0 || lum > 0.0);  // else every pixel goes to the second pass
      const double invN = 1.0 / (double)ca->N;
      c.x = em.x + 0.0 * invN;
      c.y = em.y + 0.0 * invN;
      c.z = em.z + 0.0 * invN;
    }
    dc_cum = c;
    if (S > 0) {  // imagetracer.py:83-101: the same additions, the same final scaling
      V3 sum = {0.0, 0.0, 0.0};
      for (int s = 0; s < nsamp; ++s) {
        sum.x = sum.x + c.x;
        sum.y = sum.y + c.y;
        sum.z = sum.z + c.z;
      }
      const double k = 1.0 / (double)(S * S);
      dc_cum.x = sum.x * k;
      dc_cum.y = sum.y * k;
      dc_cum.z = sum.z * k;
    }
  };
  // (hc_: the caller's |o'|^2 - 1 for an orthogonal camera, where it depends on the tile; else dc_hc is used)
  auto dome_value = [&](int only, double hc_, float dmax2, float dmin, bool all, V3 &cum, bool &settled) -> bool {
    dome_prepare(only);
    settled = false;
    cum = {0.0, 0.0, 0.0};
    if (!ORTHO) hc_ = dc_hc;
    if (!dc_usable || !(hc_ < -0.5 && dc_fro2 * dmax2 < 1e6f && dmin > 1e-6f && !all)) return false;
    settled = dc_settled;
    cum = dc_cum;
    return true;
  };
  // A workgroup takes a STRIP of four tiles (32 x 8 pixels, one block of rows), one tile per wave.  Where the
  // whole strip can only see the dome -- most of a frame under an open sky -- one cull settles all four:
  // the four waves share its passes (wave w looks at shapes [64 w, 64 w + 64), [64 (w + 4), ...), ...) and add
  // their survivor counts up through LDS.  Otherwise every wave culls its own tile as before.
  const int strips_x = (tiles_x + 3) >> 2;
  const int nstrips = strips_x * tiles_y;
  (void)ntiles;
  (void)nwaves;
  // (with a single pass per cull there is nothing to share: the strip's verdict would only delay the tiles)
  const bool use_strips = dome_on && !ORTHO && !HIER && RENDERER != PT_RENDERER_POINTLIGHT && npass >= 2;
  __shared__ int strip_ns[2][PT_BLOCK / 64], strip_only[2][PT_BLOCK / 64];
  int parity = 0;
  // the bounding spheres a wave looks at first are the same for every tile and strip it takes: loaded once
  float4 b_kept = {0.0f, 0.0f, 0.0f, -1.0f}, sb_kept = {0.0f, 0.0f, 0.0f, -1.0f};
  if (KEEP) {
    b_kept = a.bounds[(!HIER && lane < a.n_shapes) ? lane : 0];
    if (use_strips && wib * 64 + lane < a.n_shapes) sb_kept = a.bounds[wib * 64 + lane];
  }
  // The cull the four waves of a workgroup share: the shapes the cone over tile columns [tx0, tx1) and local rows
  // [lr0, lr1] can touch are counted (wave w looks at passes w, w + 4, ...); true if that is one sphere and the dome
  // shortcut holds for it (then `cum` / `settled` as dome_value gives them).  One workgroup barrier per call.
  auto shared_cull = [&](int tx0, int tx1, int lr0, int lr1, V3 &cum, bool &settled) -> bool {
    const int sgr0 = global_row(a, lr0);
    const int sgr1 = global_row(a, lr1 < rows_local ? lr1 : rows_local - 1);
    // (a rank's rows interleave with other ranks': the cone over [sgr0, sgr1] covers those too -- a superset)
    const TileCone sc = tile_cone(a, tx0 * 8, (tx1 * 8 < W) ? tx1 * 8 : W, sgr0, sgr1);
    int ns_ = 0, only_ = 0;
    for (int p = wib; p < npass && ns_ <= 1; p += PT_BLOCK / 64) {
      const int slot = p * 64 + lane;
      bool keep = false;
      float4 b = {0.0f, 0.0f, 0.0f, -1.0f};
      if (slot < a.n_shapes) b = (KEEP && p == wib) ? sb_kept : a.bounds[slot];
      const bool isplane = slot >= a.n_spheres && slot < a.n_shapes;
      if (slot < a.n_spheres) keep = cone_keeps(sc, b);
      if (__any(isplane)) {
        const bool pk = plane_keeps(sc, b, isplane);
        if (isplane) keep = pk;
      }
      const unsigned long long m = __ballot(keep);
      ns_ += __popcll(m);
      if (m) only_ = p * 64 + (__ffsll((long long)m) - 1);
    }
    if (lane == 0) {
      strip_ns[parity][wib] = ns_;
      strip_only[parity][wib] = only_;
    }
    __syncthreads();  // (one barrier per cull: the buffers alternate, so nobody overwrites what a slower wave still reads)
    int tot = 0, only_all = 0;
#pragma unroll
    for (int w = 0; w < PT_BLOCK / 64; ++w) {
      const int nw = strip_ns[parity][w];
      tot += nw;
      if (nw) only_all = strip_only[parity][w];
    }
    parity ^= 1;
    settled = false;
    if (tot == 1 && only_all < a.n_spheres)
      return dome_value(__builtin_amdgcn_readfirstlane(only_all), 0.0, sc.dmax2, sc.dmin, sc.all, cum, settled);
    return false;
  };
  // Path tracer's first pass on frames with many more strips than workgroups (4K): a workgroup takes a BLOCK of four
  // (or two) strips, one below the other (32 x 32 pixels), and culls the block first -- under an open sky that one cull
  // settles sixteen tiles, which then cost a store each.  A block that sees more than the dome is worked through
  // strip by strip as before.
  const bool use_blocks = BLOCKS && KEEP && use_strips;  // (the host asks for it where there are blocks enough for the workgroups)
  int block_h = 1;  // strips per block: 4 or 2
  if constexpr (BLOCKS) block_h = use_blocks ? cold_args(a)->block_h : 1;
  const int blocks_y = (tiles_y + block_h - 1) / block_h;
  const int nwork = use_blocks ? strips_x * blocks_y : nstrips;
  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
   const int wy = work / strips_x, tx_first = (work - wy * strips_x) * 4;
   const int tx_end = tx_first + 4 < tiles_x ? tx_first + 4 : tiles_x;
   const int ty_first = BLOCKS ? wy * block_h : wy;
   const int ty_end = BLOCKS ? (ty_first + block_h < tiles_y ? ty_first + block_h : tiles_y) : ty_first + 1;
   bool block_dome = false, block_settled = false;
   V3 strip_cum = {0.0, 0.0, 0.0};  // (KEEP: the dome's value is dc_cum, this copy is not used)
   if (use_blocks) block_dome = shared_cull(tx_first, tx_end, ty_first * 8, ty_end * 8 - 1, strip_cum, block_settled);
   for (int ty = ty_first; ty < ty_end; ++ty) {
   bool strip_dome = block_dome, strip_settled = block_settled;
   if (use_strips && !block_dome) strip_dome = shared_cull(tx_first, tx_end, ty * 8, ty * 8 + 7, strip_cum, strip_settled);
   {
    const int tx = tx_first + wib;
    if (tx >= tx_end) continue;
    const int tile = ty * tiles_x + tx;
    PT_TSTAMP(7);
    const int col = tx * 8 + (lane & 7), lrow = ty * 8 + (lane >> 3);
    const bool active = col < W && lrow < rows_local;
    // clamp so that idle lanes of edge tiles stand on a real pixel (they only widen nothing)
    const int ccol = col < W ? col : W - 1, clrow = lrow < rows_local ? lrow : rows_local - 1;
    const long long pix = (long long)clrow * W + ccol;
    if (strip_dome) {  // (settled by the strip's cull: nothing but the dome can be seen from these four tiles)
      if (strip_settled && active) {
        store_pixel(a, pix, KEEP ? dc_cum : strip_cum, out_f32);
        nrays += (unsigned long long)nsamp;
        nres += (unsigned long long)nsamp;
      }
      if (RENDERER == PT_RENDERER_PATHTRACER) {
        const unsigned long long todo = strip_settled ? 0ULL : __ballot(active);
        if (lane == 0) {
          rmask[tile] = todo;
          rkeys[tile] = (unsigned char)__popcll(todo);
            if (todo) note_flagged(pt_queue(a), __popcll(todo));
        }
      }
      continue;
    }
    const int pcol = ccol, grow = global_row(a, clrow);

    PT_TSTAMP(0);
    // ---- cull: one bounding sphere per lane per pass -> ballot -> LDS ----
    // tile rectangle: columns [tx*8, ..), global rows of its first/last local row
    const int gr0 = global_row(a, ty * 8);
    const int gr1 = global_row(a, (ty * 8 + 7 < rows_local) ? ty * 8 + 7 : rows_local - 1);
    // (else: the first pass's bounding sphere is requested before the cone arithmetic so that the two overlap)
    const float4 b_first = KEEP ? b_kept : a.bounds[(!HIER && lane < a.n_shapes) ? lane : 0];
    const TileCone tc = tile_cone(a, tx * 8, (tx * 8 + 8 < W) ? tx * 8 + 8 : W, gr0, gr1);
    PT_TSTAMP(1);
    int tpass = npass;
    const unsigned int *list = nullptr;
    int list_cnt = 0;         // HIER: entries of the cell's list
    int nsurv = 0, only = 0;  // survivors of this tile; the slot of the last one (wave-uniform)
    bool dome_here = false;   // PATHTRACER: the frame's dome candidate (a.dome_slot) is among them
    if (HIER) {
      // the tile's 8 rows are consecutive global rows starting at a multiple of 8 (the host checks
      // row_block % 8 == 0), so they lie in one cell row
      const int cell = __builtin_amdgcn_readfirstlane((gr0 / PT_CELL) * a.cells_x + (tx * 8) / PT_CELL);
      const int cnt = PT_KI(a.cell_count)[cell];
      list_cnt = cnt;
      list = a.cell_list + (size_t)cell * a.cell_stride;
      tpass = (cnt + 63) >> 6;
      for (int p = 0; p < tpass; ++p) {
        const int idx = p * 64 + lane;
        bool keep = false;
        int slot = 0;
        float4 b = {0.0f, 0.0f, 0.0f, -1.0f};
        if (idx < cnt) {
          slot = (int)list[idx];
          b = a.bounds[slot];
        }
        const bool isplane = idx < cnt && slot >= a.n_spheres;
        if (idx < cnt && !isplane) keep = cone_keeps(tc, b);
        if (__any(isplane)) {
          const bool pk = plane_keeps(tc, b, isplane);
          if (isplane) keep = pk;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) pt_lds_masks[mbase + p] = m;
        nsurv += __popcll(m);
        if (m) only = (int)list[p * 64 + (__ffsll((long long)m) - 1)];
        if (RENDERER == PT_RENDERER_PATHTRACER && __ballot(keep && slot == dome_slot)) dome_here = true;
      }
    } else {
      for (int p = 0; p < npass; ++p) {
        const int slot = p * 64 + lane;
        bool keep = false;
        float4 b = {0.0f, 0.0f, 0.0f, -1.0f};
        if (slot < a.n_shapes) b = p == 0 ? b_first : a.bounds[slot];  // 16 B per lane, coalesced
        const bool isplane = slot >= a.n_spheres && slot < a.n_shapes;
        if (slot < a.n_spheres) keep = cone_keeps(tc, b);
        if (__any(isplane)) {
          const bool pk = plane_keeps(tc, b, isplane);
          if (isplane) keep = pk;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) pt_lds_masks[mbase + p] = m;
        nsurv += __popcll(m);
        if (m) only = p * 64 + (__ffsll((long long)m) - 1);
        if (RENDERER == PT_RENDERER_PATHTRACER && p == (dome_slot >> 6) && ((m >> (dome_slot & 63)) & 1ULL)) dome_here = true;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PT_TSTAMP(2);

    // ---- one survivor that every ray of the tile is certain to hit (the sky dome): no rays needed ----
    // The only shape that can be hit at all is a sphere with the camera well inside it (hoisted
    // c = |o'|^2 - 1 < -0.5).  Then for every primary ray delta = bb^2 + 4 aa |c| > 0 and the far root
    // is >= (1 - |o'|) / |d'| >= 0.29 / sqrt(|invm|_F^2 |d|^2) > 2.9e-4 > tmin (the product is
    // checked < 1e6; pt_scene_upload stores |invm|_F^2 = +inf for shapes whose scale is not within
    // 1e-6 .. 1e6, and |d| >= 1e-6 is checked, so nothing under- or overflows): the reference finds
    // exactly this hit for every sample.  With uniform pigments its colour does not depend on the hit
    // point, so each sample's value is known without generating the ray or drawing its jitter.
    // Orthogonal camera: the origins differ, but |o'|^2 is convex in the image position, so it is below
    // 0.5 for every ray when it is (by a margin, in fp32) at the tile's four corner origins; |d'| is one
    // number for the frame.
    if (dome_on && nsurv == 1 && only < a.n_spheres) {
      only = __builtin_amdgcn_readfirstlane(only);
      double hc_;
      float dmax2 = tc.dmax2, dmin = tc.dmin;
      if (ORTHO) {
        const float fro2 = (float)PT_KD(&a.recs[only])[13];  // PtShapeRec::fro2
        pt_kdouble m = PT_KD(a.recs[only].invm);
        const float ox = (float)m[0] * tc.kx + (float)m[1] * tc.ky + (float)m[2] * tc.kz + (float)m[3];
        const float oy = (float)m[4] * tc.kx + (float)m[5] * tc.ky + (float)m[6] * tc.kz + (float)m[7];
        const float oz = (float)m[8] * tc.kx + (float)m[9] * tc.ky + (float)m[10] * tc.kz + (float)m[11];
        float o2 = ox * ox + oy * oy + oz * oz;  // this lane's corner (lane & 3)
        o2 = fmaxf(o2, __shfl_xor(o2, 1, 64));
        o2 = fmaxf(o2, __shfl_xor(o2, 2, 64));
        // fp32 evaluation: relative 1e-6 of the terms; |o'| <= |invm|_F (|k| + 1)-ish, hence the slack
        const float slack = 1e-5f * (fro2 * (tc.oabs * tc.oabs * 3.0f + 1.0f) + 1.0f);
        hc_ = (o2 + slack < 0.45f) ? -0.55 : 0.0;  // NaN: 0.0
        pt_kargs cc_ = cold_args(a);
        const float d2 = cc_->cone_apex[0] * cc_->cone_apex[0] + cc_->cone_apex[1] * cc_->cone_apex[1] +
                         cc_->cone_apex[2] * cc_->cone_apex[2];
        dmax2 = d2 * (1.0f + 1e-5f);
        dmin = __fsqrt_rn(d2) * (1.0f - 1e-5f);
      } else {
        hc_ = 0.0;  // (dome_value takes the hoisted constant of `only` itself)
      }
      V3 cum;
      bool settled;
      if (dome_value(only, hc_, dmax2, dmin, tc.all, cum, settled)) {
        if (settled && active) {
          store_pixel(a, pix, cum, out_f32);
          nrays += (unsigned long long)nsamp;
          nres += (unsigned long long)nsamp;
        }
        if (RENDERER == PT_RENDERER_PATHTRACER) {
          const unsigned long long todo = settled ? 0ULL : __ballot(active);
          if (lane == 0) {
            rmask[tile] = todo;
            rkeys[tile] = (unsigned char)__popcll(todo);
            if (todo) note_flagged(pt_queue(a), __popcll(todo));
          }
        }
        __builtin_amdgcn_wave_barrier();  // the next tile overwrites this wave's mask slice
        continue;
      }
    }

    // ---- path tracer, a dome among several survivors: the first pass only CLASSIFIES ----
    // The dome (same conditions as above, for this tile) is hit by every primary ray of the tile; a pixel
    // whose own cone misses the bounding spheres of all the other survivors can hit nothing else, so all
    // its samples end on the dome at depth 0 (black BRDF pigment, no Russian roulette at depth 0): its value
    // is the same replayed sum, no ray needed.  Every other pixel of the tile is left to the second pass,
    // untraced: there a pixel's samples are spread over lanes, here they would be walked one by one by a
    // wave that 60 finished lanes wait for.  (Planes carry no bounding sphere: a pixel of a tile some plane
    // survived in is always left over.)
    if (RENDERER == PT_RENDERER_PATHTRACER && !ORTHO && dome_on && dome_here && nsurv > 1) {
      pt_kargs ca = cold_args(a);
      dome_prepare(dome_slot);
      // (dc_settled for the path tracer: Russian roulette on and a black BRDF pigment -- the sample ends on the dome)
      if (dc_usable && dc_settled && dc_hc < -0.5 && dc_fro2 * tc.dmax2 < 1e6f && tc.dmin > 1e-6f && !tc.all) {
        const TileCone pc = pixel_cone(cone_cam(a), tc, pcol, grow);
        bool hitable = pc.all;
        for (int p = 0; p < tpass; ++p) {
          const unsigned long long mv = pt_lds_masks[mbase + p];
          const unsigned m_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mv >> 32));
          const unsigned m_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mv);
          unsigned long long mask = ((unsigned long long)m_hi << 32) | (unsigned long long)m_lo;
          if (!mask) continue;
          // this pass's 64 bounding spheres, one per lane (one coalesced load), handed out by v_readlane: a
          // dependent scalar load per survivor would cost its latency per survivor -- dozens per tile where the
          // spheres crowd
          const int myi = p * 64 + lane;
          int slot_q = myi;
          if (HIER) slot_q = myi < list_cnt ? (int)list[myi] : 0;
          const float4 bq = a.bounds[slot_q < a.n_shapes ? slot_q : 0];
          while (mask) {
            const int bit = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int slot = __builtin_amdgcn_readlane(slot_q, bit);
            if (slot == dome_slot) continue;
            if (slot >= a.n_spheres) {
              hitable = true;
            } else {
              const float4 b = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(bq.x), bit)),
                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bq.y), bit)),
                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bq.z), bit)),
                                __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bq.w), bit))};
              hitable = hitable || cone_keeps(pc, b);
            }
          }
        }
        const V3 cum = dc_cum;  // render.py:139 with cum_radiance = 0, imagetracer.py:83-101 replayed
        if (active && !hitable) {
          store_pixel(a, pix, cum, out_f32);
          nrays += (unsigned long long)nsamp;
          nres += (unsigned long long)nsamp;
        }
        const unsigned long long todo = __ballot(active && hitable);
        if (lane == 0) {
          rmask[tile] = todo;
          rkeys[tile] = (unsigned char)__popcll(todo);
            if (todo) note_flagged(pt_queue(a), __popcll(todo));
        }
        __builtin_amdgcn_wave_barrier();  // the next tile overwrites this wave's mask slice
        continue;
      }
    }

    Pcg pcg;
    unsigned long long gpix = 0;
    if (S > 0) {
      pt_kargs c = cold_args(a);
      gpix = (unsigned long long)grow * c->W + pcol;
      if (c->pcg_mode == PT_PCG_PIXEL) pcg_seed(pcg, c->s0, c->q0 + gpix);
    }
    V3 cum = {0.0, 0.0, 0.0};
    bool alive = active;  // PATHTRACER: still a pixel this pass can finish
    int pix_rays = 0;
    for (int s = 0; s < nsamp; ++s) {
      if (RENDERER == PT_RENDERER_PATHTRACER && !__any(alive)) break;
      double up = 0.5, vp = 0.5;
      if (S > 0) {  // imagetracer.py:86-93
        pt_kargs c = cold_args(a);
        if (c->pcg_mode == PT_PCG_SAMPLE) pcg_seed(pcg, c->s0, c->q0 + gpix * (unsigned)nsamp + (unsigned)s);
        const int sr = s / S, sc = s - sr * S;
        up = ((double)sc + pcg_float(pcg)) / (double)S;
        vp = ((double)sr + pcg_float(pcg)) / (double)S;
      }
      const Ray ray = primary_ray(a, pcol, grow, up, vp);
      PT_TSTAMP(3);
      double best_t;
      const int hit = world_query_tile<RENDERER == PT_RENDERER_ONOFF, HIER, !ORTHO>(a, ray, mbase, tpass, best_t, alive, list);
      PT_TSTAMP(4);
      if (alive) pix_rays++;
      V3 c;
      {
        pt_kargs ca = cold_args(a);
        c.x = ca->bg[0];
        c.y = ca->bg[1];
        c.z = ca->bg[2];
      }
      if (RENDERER == PT_RENDERER_ONOFF) {  // render.py:52-53
        if (hit >= 0) {
          pt_kargs ca = cold_args(a);
          c.x = ca->onoff[0];
          c.y = ca->onoff[1];
          c.z = ca->onoff[2];
        }
      } else if (RENDERER == PT_RENDERER_POINTLIGHT) {
        unsigned long long shadow_rays = 0;
        c = pointlight_shade(a, ray, hit, best_t, alive, c, shadow_rays);
        pix_rays += (int)shadow_rays;
      } else if (RENDERER == PT_RENDERER_PATHTRACER) {  // render.py:103-139 at depth 0, no recursion
        if (hit >= 0) {
          pt_kargs ca = cold_args(a);
          const PtShapeAux *ax = ca->aux + hit;
          Hit h;
          h.u = 0.0;
          h.v = 0.0;
          if (ax->needs_uv) hit_details(a.recs + hit, ax, ray, best_t, h, true);
          const V3 hc = brdf_pigment(a, ax, h.u, h.v);
          const V3 em = emitted_pigment(a, ax, h.u, h.v);
          const double lum = max2(max2(hc.x, hc.y), hc.z);
          // Russian roulette already at depth 0 (rr_limit <= 0) draws a number: not for this pass
          if (ca->rr <= 0 || lum > 0.0) alive = false;
          const double invN = 1.0 / (double)ca->N;
          c.x = em.x + 0.0 * invN;  // render.py:139 with cum_radiance = 0
          c.y = em.y + 0.0 * invN;
          c.z = em.z + 0.0 * invN;
        }
      } else {  // render.py:65-74
        if (hit >= 0) {
          const PtShapeAux *ax = cold_args(a)->aux + hit;
          Hit h;
          h.u = 0.0;
          h.v = 0.0;
          if (ax->needs_uv) hit_details(a.recs + hit, ax, ray, best_t, h, true);
          const V3 p1 = brdf_pigment(a, ax, h.u, h.v);
          const V3 p2 = emitted_pigment(a, ax, h.u, h.v);
          c.x = p1.x + p2.x;
          c.y = p1.y + p2.y;
          c.z = p1.z + p2.z;
        }
      }
      if (S > 0) {
        cum.x = cum.x + c.x;
        cum.y = cum.y + c.y;
        cum.z = cum.z + c.z;
      } else {
        cum = c;
      }
      PT_TSTAMP(5);
    }
    if (S > 0) {  // imagetracer.py:99-101
      const double k = 1.0 / (double)(S * S);
      cum.x = cum.x * k;
      cum.y = cum.y * k;
      cum.z = cum.z * k;
    }
    if (alive) {
      store_pixel(a, pix, cum, out_f32);
      nrays += (unsigned long long)pix_rays;
    }
    if (RENDERER == PT_RENDERER_PATHTRACER) {
      const unsigned long long todo = __ballot(active && !alive);
      if (lane == 0) {
        rmask[tile] = todo;
        rkeys[tile] = (unsigned char)__popcll(todo);
        if (todo) note_flagged(pt_queue(a), __popcll(todo));
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next tile overwrites this wave's mask slice
    PT_TSTAMP(6);
   }
   }
  }
#ifdef PT_DEBUG_TIME
  // sampled (every 64th workgroup) so that the report's own atomics do not disturb the other waves
  if (RENDERER != PT_RENDERER_PATHTRACER && (threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
  pt_dbg_flush();
#endif
  add_ray_count(a, nrays, count_base, nres);
}

// ---- OnOff / Flat, pixel-centre rays (S = 0), perspective camera: 16x16 tiles, FOUR pixels per lane ------------
// pt_tile_kernel spends more than half of a tile's instructions and most of its dependent latency on what is
// per TILE, not per pixel (profiles/r03_tile_sections.txt: cone 14 %, cull 21 %, loop bookkeeping and prologue
// 23 % of a wave's cycles on C2).  Here a wave owns a 16 x 16-pixel tile = four 8 x 8 quadrants, lane l holding
// pixel (l & 7, l >> 3) of EACH quadrant: one cone, one cull (the survivors stay in SGPR masks: with one sample
// per pixel nothing is replayed, so no LDS), one walk over the survivors whose scalar-loaded record serves four
// independent rays per lane (four dependency chains for the fp64 pipe to overlap), no tile loop (a 2 x 2 block of
// tiles per workgroup, taken from a 2-D grid).  The rays share arithmetic bit for bit: u depends on the column
// only, v on the row only (imagetracer.py:56-58), and in M*(d, (1-2u)a, 2v-1) the partial sum of the first two
// terms is the same for the two pixels of a column pair (transformations.py:58-86 adds left to right).
// Every ray still goes through exactly the reference arithmetic of world_query_tile / hit_details; a larger
// tile only means a wider cone, i.e. more survivors.  Used when the rows of a tile are consecutive image rows
// (one rank, or row blocks that are multiples of 16).
struct Hit4 {
  double best_t[4];
  int best[4];
};

// shapes.py:103-121 for one ray given the object-space ray (PT_SPHERE_ROOTS as a function)
template <bool ANYHIT>
PT_DEV void sphere_roots1(const PtKArgs &a, int slot, bool active, double tmin, double ox, double oy, double oz, double dx,
                          double dy, double dz, double aa, double cc, double &best_t, int &best) {
  const double tmax = INFINITY;
  PT_SPHERE_ROOTS(slot);
}

// SLDS (small worlds, Flat): the shapes' records (128 B + 256 B each) are staged in LDS by the workgroup and shading
//   gathers from there instead of through the vector memory path (C2: 14.4 -> 13.9 us per frame).
// NPX = 4: 16x16 tiles, four pixels per lane.  NPX = 2: 16x8 tiles, two pixels per lane (the upper two quadrants only) --
//   twice the waves with half the pixels each, for frames whose 16x16 tiles would not fill the chip.
template <int RENDERER, bool SLDS = false, int NPX = 4>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void pt_tile4_kernel(const PtKArgs a) {
  static_assert(NPX == 2 || NPX == 4, "two or four pixels per lane");
  constexpr int TH = NPX == 4 ? 16 : 8;  // tile height
  constexpr bool ANYHIT = RENDERER == PT_RENDERER_ONOFF;
#ifdef PT_DEBUG_TIME
  // cycles of this wave in: 0 prologue, 1 cone, 2 cull, 3 dome tile, 4 rays, 5 query, 6 shade, 7 store
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define PT_T4(k) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[k] += tn - tprev; tprev = tn; } while (0)
#else
#define PT_T4(k) do { } while (0)
#endif
  int W, rows_local, npass;
  bool dome_on, out_f32;
  {
    pt_kargs c = cold_args(a);
    out_f32 = c->out_f32 != 0;
    W = c->W;
    rows_local = c->rows_local;
    npass = c->npass;
    dome_on = c->dome_shortcut != 0;
  }
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int tx = blockIdx.x * 2 + (wib & 1), ty = blockIdx.y * 2 + (wib >> 1);
  unsigned long long nrays = 0, nres = 0;
  const bool valid = tx * 16 < W && ty * TH < rows_local;  // (wave-uniform; the ray count below needs every wave)
  if (SLDS) {  // recs[] then aux[] into LDS (8-byte words; every wave of the workgroup takes part)
    const unsigned long long *src = (const unsigned long long *)a.recs;
    for (int k = threadIdx.x; k < a.n_shapes * 16; k += PT_BLOCK) pt_lds_masks[k] = src[k];
    src = (const unsigned long long *)a.aux;
    for (int k = threadIdx.x; k < a.n_shapes * 32; k += PT_BLOCK) pt_lds_masks[a.n_shapes * 16 + k] = src[k];
    __syncthreads();  // (measured: placing this barrier before the shading instead, behind cone / cull / query, gains nothing)
  }
  if (valid) {
    // ---- cone + cull of the 16 x 16 tile ----
    const int lr0 = ty * TH, lr1 = (lr0 + TH - 1 < rows_local) ? lr0 + TH - 1 : rows_local - 1;
    const int gr0 = global_row(a, lr0);  // the tile's rows are consecutive image rows (host: n_ranks == 1 or row_block % 16 == 0)
    const float4 b_first = a.bounds[lane < a.n_shapes ? lane : 0];
    PT_T4(0);
    const TileCone tc = tile_cone(a, tx * 16, (tx * 16 + 16 < W) ? tx * 16 + 16 : W, gr0, gr0 + (lr1 - lr0));
    PT_T4(1);
    unsigned long long masks[4] = {0ULL, 0ULL, 0ULL, 0ULL};  // (npass <= 4: the host sends larger worlds elsewhere)
    int nsurv = 0, only = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p < npass) {
        const int slot = p * 64 + lane;
        bool keep = false;
        float4 b = {0.0f, 0.0f, 0.0f, -1.0f};
        if (slot < a.n_shapes) b = p == 0 ? b_first : a.bounds[slot];
        const bool isplane = slot >= a.n_spheres && slot < a.n_shapes;
        if (slot < a.n_spheres) keep = cone_keeps(tc, b);
        if (__any(isplane)) {
          const bool pk = plane_keeps(tc, b, isplane);
          if (isplane) keep = pk;
        }
        const unsigned long long m = __ballot(keep);
        masks[p] = m;
        nsurv += __popcll(m);
        if (m) only = p * 64 + (__ffsll((long long)m) - 1);
      }
    }
    PT_T4(2);
    // pixel k of this lane: quadrant (k & 1, k >> 1)
    const int colA = tx * 16 + (lane & 7), colB = colA + 8;
    const int lrowA = lr0 + (lane >> 3), lrowB = lrowA + 8;
    const bool okcA = colA < W, okcB = colB < W, okrA = lrowA < rows_local, okrB = NPX == 4 && lrowB < rows_local;
    const bool act[4] = {okcA && okrA, okcB && okrA, okcA && okrB, okcB && okrB};
    const int ccA = okcA ? colA : W - 1, ccB = okcB ? colB : W - 1;  // idle lanes stand on a real pixel
    const int crA = okrA ? lrowA : rows_local - 1, crB = okrB ? lrowB : rows_local - 1;
    const long long pix[4] = {(long long)crA * W + ccA, (long long)crA * W + ccB, (long long)crB * W + ccA, (long long)crB * W + ccB};
    bool done = false;
    // ---- the dome shortcut (see pt_tile_kernel): one survivor, the camera well inside it, uniform pigments ----
    if (dome_on && nsurv == 1 && only < a.n_spheres) {
      only = __builtin_amdgcn_readfirstlane(only);
      pt_kargs ca = cold_args(a);
      const PtShapeAux *ax = ca->aux + only;
      const float fro2 = (float)PT_KD(&a.recs[only])[13];  // PtShapeRec::fro2
      const double hc = (only < a.n_diag) ? PT_KD(&a.hoist_diag[only])[6] : PT_KD(&a.hoist[only])[3];
      if (ax->needs_uv == 0 && hc < -0.5 && fro2 * tc.dmax2 < 1e6f && tc.dmin > 1e-6f && !tc.all) {
        V3 c;
        if (RENDERER == PT_RENDERER_ONOFF) {
          c.x = ca->onoff[0];
          c.y = ca->onoff[1];
          c.z = ca->onoff[2];
        } else {
          const V3 p1 = brdf_pigment(a, ax, 0.0, 0.0), p2 = emitted_pigment(a, ax, 0.0, 0.0);
          c.x = p1.x + p2.x;
          c.y = p1.y + p2.y;
          c.z = p1.z + p2.z;
        }
#pragma unroll
        for (int k = 0; k < NPX; ++k)
          if (act[k]) {
            store_pixel(a, pix[k], c, out_f32);
            nrays += 1ULL;
            nres += 1ULL;
          }
        done = true;
      }
    }
    PT_T4(3);
    if (!done) {
      // ---- the four primary rays (imagetracer.py:48-58, camera.py:103-124) ----
      V3 org, dir[4];
      double bgx, bgy, bgz;
      {
        pt_kargs c = cold_args(a);
        const double dist = c->cam_dist, aspect = c->cam_aspect;
        const double Wd = (double)c->W, Hd = (double)c->H;
        const int growA = gr0 + (crA - lr0), growB = gr0 + (crB - lr0);
        const double uA = ((double)ccA + 0.5) / Wd, uB = ((double)ccB + 0.5) / Wd;
        const double vA = 1.0 - ((double)growA + 0.5) / Hd, vB = 1.0 - ((double)growB + 0.5) / Hd;
        const double dyA = (1.0 - 2.0 * uA) * aspect, dyB = (1.0 - 2.0 * uB) * aspect;
        const double dzA = 2.0 * vA - 1.0, dzB = 2.0 * vB - 1.0;
        const double m0 = c->cam_m[0], m1 = c->cam_m[1], m2 = c->cam_m[2], m3 = c->cam_m[3];
        const double m4 = c->cam_m[4], m5 = c->cam_m[5], m6 = c->cam_m[6], m7 = c->cam_m[7];
        const double m8 = c->cam_m[8], m9 = c->cam_m[9], m10 = c->cam_m[10], m11 = c->cam_m[11];
        // xf_vec: (d.x*m[0] + d.y*m[1]) + d.z*m[2] -- the bracket depends on the column only
        const double xA = dist * m0 + dyA * m1, xB = dist * m0 + dyB * m1;
        const double yA = dist * m4 + dyA * m5, yB = dist * m4 + dyB * m5;
        const double zA = dist * m8 + dyA * m9, zB = dist * m8 + dyB * m9;
        dir[0] = {xA + dzA * m2, yA + dzA * m6, zA + dzA * m10};
        dir[1] = {xB + dzA * m2, yB + dzA * m6, zB + dzA * m10};
        dir[2] = {xA + dzB * m2, yA + dzB * m6, zA + dzB * m10};
        dir[3] = {xB + dzB * m2, yB + dzB * m6, zB + dzB * m10};
        // xf_point of (-dist, 0, 0): ((o.x*m[0] + 0*m[1]) + 0*m[2]) + m[3], exactly as primary_ray evaluates it
        const double ox_ = -dist, oy_ = 0.0, oz_ = 0.0;
        org.x = ox_ * m0 + oy_ * m1 + oz_ * m2 + m3;
        org.y = ox_ * m4 + oy_ * m5 + oz_ * m6 + m7;
        org.z = ox_ * m8 + oy_ * m9 + oz_ * m10 + m11;
        bgx = c->bg[0];
        bgy = c->bg[1];
        bgz = c->bg[2];
      }
      PT_T4(4);
      // ---- World.ray_intersection over the survivors, four rays per visit ----
      const double tmin = 1.0e-5;
      Hit4 h4;
      bool fast = true;
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        h4.best_t[k] = INFINITY;
        h4.best[k] = -1;
        Ray rk;
        rk.o = org;
        rk.d = dir[k];
        rk.tmin = tmin;
        fast = fast && wave_guard<true>(rk, act[k]).fast;
      }
      const int nd = a.n_diag, ns = a.n_spheres;
      bool all_hit = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned long long mask = masks[p];
        while (mask && !all_hit) {
          const int slot = p * 64 + (__ffsll((long long)mask) - 1);
          mask &= mask - 1;
          if (slot < ns) {
            if (slot < nd && fast) {
              pt_kdouble h = PT_KD(&a.hoist_diag[slot]);
              const double s0 = h[0], s1 = h[1], s2 = h[2], ox = h[3], oy = h[4], oz = h[5], cc = h[6];
#pragma unroll
              for (int k = 0; k < NPX; ++k) {
                const double dx = dir[k].x * s0, dy = dir[k].y * s1, dz = dir[k].z * s2;
                const double aa = dx * dx + dy * dy + dz * dz;
                sphere_roots1<ANYHIT>(a, slot, act[k], tmin, ox, oy, oz, dx, dy, dz, aa, cc, h4.best_t[k], h4.best[k]);
              }
            } else {
              pt_kdouble m = PT_KD(a.recs[slot].invm);
              pt_kdouble h = PT_KD(&a.hoist[slot]);
              const double ox = h[0], oy = h[1], oz = h[2], cc = h[3];
#pragma unroll
              for (int k = 0; k < NPX; ++k) {
                const double dx = dir[k].x * m[0] + dir[k].y * m[1] + dir[k].z * m[2];
                const double dy = dir[k].x * m[4] + dir[k].y * m[5] + dir[k].z * m[6];
                const double dz = dir[k].x * m[8] + dir[k].y * m[9] + dir[k].z * m[10];
                const double aa = dx * dx + dy * dy + dz * dz;
                sphere_roots1<ANYHIT>(a, slot, act[k], tmin, ox, oy, oz, dx, dy, dz, aa, cc, h4.best_t[k], h4.best[k]);
              }
            }
          } else {
            pt_kdouble m = PT_KD(a.recs[slot].invm);
            const double oz = PT_KD(&a.hoist[slot])[2];
            const double tmax = INFINITY;
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
              const double dz = dir[k].x * m[8] + dir[k].y * m[9] + dir[k].z * m[10];
              const bool active = act[k];
              double &best_t = h4.best_t[k];
              int &best = h4.best[k];
              PT_PLANE_HIT(slot);
            }
          }
          if (ANYHIT) {  // OnOff: leave as soon as every active pixel has some hit (render.py:52-53 asks no more)
            const bool open = (act[0] && h4.best[0] < 0) || (act[1] && h4.best[1] < 0) || (act[2] && h4.best[2] < 0) ||
                              (act[3] && h4.best[3] < 0);
            all_hit = __ballot(open) == 0ULL;
          }
        }
      }
      PT_T4(5);
      // ---- shade + store ----
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        V3 c = {bgx, bgy, bgz};
        const int hit = h4.best[k];
        if (RENDERER == PT_RENDERER_ONOFF) {  // render.py:52-53
          if (hit >= 0) {
            pt_kargs ca = cold_args(a);
            c.x = ca->onoff[0];
            c.y = ca->onoff[1];
            c.z = ca->onoff[2];
          }
        } else if (hit >= 0) {  // render.py:65-74
          Hit h;
          h.u = 0.0;
          h.v = 0.0;
          Ray rk;
          rk.o = org;
          rk.d = dir[k];
          rk.tmin = tmin;
          V3 p1, p2;
          if constexpr (SLDS) {
            const pt_lds_rec rec = (pt_lds_rec)(const void *)pt_lds_f64 + hit;
            const pt_lds_aux ax = (pt_lds_aux)(const void *)(pt_lds_f64 + a.n_shapes * 16) + hit;
            if (ax->needs_uv) hit_details(rec, ax, rk, h4.best_t[k], h, true);
            p1 = brdf_pigment(a, ax, h.u, h.v);
            p2 = emitted_pigment(a, ax, h.u, h.v);
          } else {
            const PtShapeAux *ax = cold_args(a)->aux + hit;
            if (ax->needs_uv) hit_details(a.recs + hit, ax, rk, h4.best_t[k], h, true);
            p1 = brdf_pigment(a, ax, h.u, h.v);
            p2 = emitted_pigment(a, ax, h.u, h.v);
          }
          c.x = p1.x + p2.x;
          c.y = p1.y + p2.y;
          c.z = p1.z + p2.z;
        }
        PT_T4(6);
        if (act[k]) {
          store_pixel(a, pix[k], c, out_f32);
          nrays += 1ULL;
        }
        PT_T4(7);
      }
    }
  }
#ifdef PT_DEBUG_TIME
#ifdef PT_DEBUG_HEAVY  // section sums of the HEAVY sampled waves only (more than PT_DEBUG_HEAVY cycles)
  {
    unsigned long long tot_ = 0;
    for (int q = 0; q < 8; ++q) tot_ += tsum[q];
    if ((threadIdx.x & 63) == 0 && ((blockIdx.y * gridDim.x + blockIdx.x) & 3) == 0 && tot_ > PT_DEBUG_HEAVY)
      for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
  }
#else
  if ((threadIdx.x & 63) == 0 && ((blockIdx.y * gridDim.x + blockIdx.x) & 15) == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
#endif
  if ((threadIdx.x & 63) == 0 && ((blockIdx.y * gridDim.x + blockIdx.x) & 15) == 0) {  // (sampled waves) the longest one, and how many took more than 16 / 24 / 32 kcycles
    unsigned long long tot = 0;
    for (int q = 0; q < 8; ++q) tot += tsum[q];
    atomicMax(pt_queue(a) + 12, tot);
    if (tot > 16384ULL) atomicAdd(pt_queue(a) + 13, 1ULL);
    if (tot > 24576ULL) atomicAdd(pt_queue(a) + 14, 1ULL);
    if (tot > 32768ULL) atomicAdd(pt_queue(a) + 15, 1ULL);
  }
#endif
  add_ray_count(a, nrays, 0, nres, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- work units for the path tracer's second pass ---------------------------------------------------------
// The first pass (pt_tile_kernel<PATHTRACER>) leaves, per 8x8 region, the mask of the pixels that need real
// path tracing and their number as a key.  The second pass works in UNITS: a unit is up to `ppu` flagged
// pixels of one region, rendered by one wave whose 64 lanes are shared out L = min(S*S, 64 / pixels) to a
// pixel -- the lanes of a pixel trace different samples of it at the same time (path_trace).  ppu is chosen
// from the frame's total F of flagged pixels: with few of them (a rank's share of a frame, a sparse frame) a
// region is cut into several units so that the whole chip works on samples in parallel instead of a few
// waves walking their pixels' S*S samples one after the other; with many, ppu = 64 (a unit = a region) and
// nothing is spent on idle lanes.  Regions without flagged pixels yield nothing.  The order of the units only
// changes WHEN a pixel is rendered, never its value.
// Units with the most pixels come first (they have the fewest lanes per pixel, hence the longest chains): a
// counting sort by size over any number of workgroups, one region per thread -- the first pass counts the regions
// by their number of flagged pixels, pt_unit_scatter turns the counts into descending offsets of the unit sizes
// (every workgroup for itself: 64 numbers) and places the units.
// queue[0] = queue head, [9] = number of units, [10] = ppu (for the statistics), [11] = F (summed up by the first
// pass), [16 + k] = regions with k flagged pixels (first pass), [96 + s] = units of s pixels placed so far; [PT_QUEUE_HEADS + 32 s] = head of shard s
// of the unit list (the second pass pulls units through PT_UNIT_SHARDS heads, 256 B apart: one word takes ~88
// dequeues/us, and thousands of waves pull); all zeroed before the first pass.
PT_DEV int unit_ppu(const unsigned long long *queue, long long lanes_cap, int nsamp, int min_rounds) {
  // lanes per pixel every unit gets at least: the largest power of two (<= S*S, <= 64) at which all flagged
  // pixels together still fit the lanes the launch keeps resident ...
  const unsigned long long total = queue[11];
  int lg = 1;
  while (lg * 2 <= 64 && lg * 2 <= nsamp && (long long)total * (lg * 2) <= lanes_cap) lg *= 2;
  // ... or more, up to four units per resident wave, as long as a unit keeps `min_rounds` rounds of work: many
  // short units spread over the chip more evenly than few long ones (a unit's time varies a lot with what its
  // pixels see), but every unit costs a fetch and a cull, and lanes beyond what speculation can use are wasted
  // (PT_PCG_PIXEL asks for more rounds per unit than PT_PCG_SAMPLE for that reason).  min_rounds < 0: -min_rounds
  // rounds, and single-round units where even those come to three or more per resident wave (a full frame of
  // PT_PCG_SAMPLE: the fetch is cheap next to what finer balancing saves; with fewer units it is not).
  const int mr = min_rounds < 0 ? -min_rounds : min_rounds;
  while (lg * 2 <= 64 && lg * 2 * mr <= nsamp && (long long)total * (lg * 2) <= 4 * lanes_cap) lg *= 2;
  if (min_rounds < 0 && lg * 2 <= 64 && lg * 2 <= nsamp && (long long)total * (lg * 2) <= 4 * lanes_cap &&
      (long long)total * (lg * 2) >= 3 * lanes_cap)
    lg *= 2;
  return 64 / lg;
}
#ifndef PT_SCATTER_BLOCK
#define PT_SCATTER_BLOCK 256
#endif
__global__ void pt_unit_scatter(const unsigned char *keys, const unsigned long long *masks, int n, int4 *units, int units_cap,
                                unsigned long long *queue, long long lanes_cap, int nsamp, int min_rounds) {
  __shared__ int cnt[65], offs[65];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = i < n ? keys[i] : 0;
  const unsigned long long m = i < n ? masks[i] : 0ULL;  // carried in the unit: one dependent load less when a wave fetches it
  const int h = threadIdx.x < 64 ? (int)queue[16 + threadIdx.x + 1] : 0;  // (requested together with F: one round trip)
  const int ppu = unit_ppu(queue, lanes_cap, nsamp, min_rounds);
  // regions with k flagged pixels (counted by the first pass) -> units of s pixels: a region yields k / ppu units
  // of ppu pixels and one of k % ppu.  The first wave does it, lane k - 1 for the regions of k pixels.
  if (threadIdx.x < 65) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int kk = threadIdx.x + 1;
    if (h) {
      if (kk >= ppu) atomicAdd(&cnt[ppu], h * (kk / ppu));
      if (kk % ppu) atomicAdd(&cnt[kk % ppu], h);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // offs[s] = units of more than s pixels (descending order of size)
    const int sz = 64 - threadIdx.x;  // lane 0 holds the largest size
    const int c = cnt[sz];
    int upto = c;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(upto, off, 64);
      if ((int)threadIdx.x >= off) upto += v;
    }
    offs[sz] = upto - c;
    if (threadIdx.x == 63 && blockIdx.x == 0) {
      queue[9] = (unsigned long long)(upto < units_cap ? upto : units_cap);
      queue[10] = (unsigned long long)ppu;
    }
  }
  __syncthreads();
  const int full = k / ppu, rem = k - full * ppu;
  // the units of ppu pixels (most of them): one returning atomic per wave, the lanes share out what it reserved
  const int lane = threadIdx.x & 63;
  int upto = full;  // inclusive prefix sum over the wave
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(upto, off, 64);
    if (lane >= off) upto += v;
  }
  const int wave_total = __shfl(upto, 63, 64);
  int base = 0;
  if (wave_total) {
    if (lane == 63) base = (int)atomicAdd(queue + 96 + ppu, (unsigned long long)wave_total);
    base = __shfl(base, 63, 64);
  }
  if (!k) return;
  const int mlo = (int)(unsigned)m, mhi = (int)(unsigned)(m >> 32);
  if (full) {
    const int at = offs[ppu] + base + upto - full;
    for (int g = 0; g < full; ++g)
      if (at + g < units_cap) units[at + g] = make_int4(i, (g * ppu) | (ppu << 8), mlo, mhi);  // (region, first | count << 8, mask)
  }
  if (rem) {
    const int at = offs[rem] + (int)atomicAdd(queue + 96 + rem, 1ULL);
    if (at < units_cap) units[at] = make_int4(i, (full * ppu) | (rem << 8), mlo, mhi);
  }
}

// position of the n-th (0-based) set bit of m (which has more than n bits set)
PT_DEV int nth_set_bit(unsigned long long m, int n) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll((m >> pos) & ((1ULL << w) - 1ULL));
    if (n >= c) {
      pos += w;
      n -= c;
    }
  }
  return pos & 63;
}

#ifdef PT_DEBUG_TIME
#define PT_TRACE_LEN 8192
__device__ unsigned long long pt_trace[PT_TRACE_LEN + 64 * 80];  // (+ the traced unit's validated draw counts: [pixel][sample], tools/dbgdraws.py)
#define PT_UNITLOG_LEN 16384
// per work unit of the second pass: start tick, end tick, rounds | iterations << 32, pixels | lanes per pixel << 8 |
// workgroup << 16, then the unit's cycles in: scattered-ray queries, shade, sample start + primary query, commit + fetch
__device__ unsigned long long pt_unitlog[PT_UNITLOG_LEN * 8];
#endif

// ---- PathTracer (render.py:99-139) as a per-lane state machine ----------------------------------------
// The reference recursion is depth-first; frame `k` of the explicit stack is the call at depth k.
// Frame fields (in ws, [slot][field][thread] so a wave's accesses are contiguous):
//   0..2 hit_color (after Russian roulette)   3..5 emitted
//   N > 1 only: 6..8 cum_radiance, 9 children done, 10..12 hit point, 13..15 normal,
//               16..18 incoming direction, 19 brdf kind
// Pixels are handed out dynamically (one wave-aggregated atomic per refill): a lane that finishes
// a cheap pixel (sky) immediately takes the next one, so a few expensive pixels (deep recursion,
// num_of_rays > 1) do not hold 63 idle lanes hostage.  Per-pixel seeds make the image independent
// of which lane renders which pixel.
struct PathCtx {
  double *ws;
  size_t stride;  // frame_doubles * nthreads
  size_t nthreads;
  int gtid;
  int lds_base, lds_frame;  // LDS frames: first double of the frame area, doubles per frame
};
// The frame stack lives in LDS whenever (max_depth x frame) x 256 lanes fits beside the survivor masks
// (LDSF): the second pass is a chain of dependent steps per pixel, and a frame access that goes to
// HBM costs more than the step's arithmetic.  Same [slot][field][lane] layout in both homes.
extern __shared__ double pt_lds_f64[];  // the same dynamic LDS block as pt_lds_masks
template <bool LDSF>
PT_DEV double ws_get(const PathCtx &w, int slot, int field) {
  if (LDSF) return pt_lds_f64[w.lds_base + (slot * w.lds_frame + field) * PT_BLOCK + (int)threadIdx.x];
  return w.ws[(size_t)slot * w.stride + (size_t)field * w.nthreads + w.gtid];
}
template <bool LDSF>
PT_DEV void ws_put(const PathCtx &w, int slot, int field, double v) {
  if (LDSF)
    pt_lds_f64[w.lds_base + (slot * w.lds_frame + field) * PT_BLOCK + (int)threadIdx.x] = v;
  else
    w.ws[(size_t)slot * w.stride + (size_t)field * w.nthreads + w.gtid] = v;
}

// next pixel for every lane with `need` set; returns -1 when the frame is exhausted
PT_DEV long long next_pixel(const PtKArgs &a, bool need, long long npix) {
  const unsigned long long mask = __ballot(need);
  long long pix = -1;
  if (need) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    const int rank = __popcll(mask & ((1ULL << lane) - 1ULL));
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(pt_queue(a), (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
    const long long p = (long long)(base + rank);
    pix = p < npix ? p : -1;
  }
  return pix;
}

// Two kinds of work alternate inside a wave, each executed only by the lanes that need it and only
// when enough of them do (the bodies are skipped wave-wide otherwise):
//   P  lanes starting a sample (mode 0): jitter draws, primary ray, query against the survivors.
//   S  lanes holding a scattered ray (mode 1): query against ALL shapes.
// Both kinds then share one shade + unwind block (deliver radiance up the frame stack, scatter the
// next child) which leaves each lane with a ray to query (mode 1), a finished sample (mode 0 / 3) or a
// finished pixel (mode 2).  S queries are batched until >= 16 lanes wait, so the 32..10k-shape loop
// does not run for one or two lanes at a time; regions of pure background never run it.
//
// !TILED (orthogonal camera): 1 lane = 1 pixel, pixels come from one global queue, a lane walks its pixel's
// samples one after the other and P-steps run the full shape loop.
//
// TILED (perspective camera, second pass): a wave works through UNITS (pt_unit_scatter): up to 64 flagged pixels
// of one 8x8 region.  The unit's P-steps use the hoisted, culled tile query against the region's survivor
// masks.  The wave's lanes are shared out L = min(S*S, 64 / pixels) to a pixel, and the L lanes of a pixel
// trace L consecutive samples of it AT THE SAME TIME (a "round"):
//   PT_PCG_SAMPLE  every sample owns its generator: the L samples are independent, all of them count.
//   PT_PCG_PIXEL   the samples of a pixel share ONE generator, consumed in program order: where sample k+1
//     starts in the stream depends on how many numbers sample k drew, which is only known once its path has
//     ended.  Lane j therefore SPECULATES: it guesses what each of the j samples before it draws -- what the last
//     validated sample of the pixel drew, or, where that has been the better guess for this pixel so far, what each
//     sample's upper neighbour in the S x S grid of strata drew (a pixel across an edge repeats its row of short and
//     long paths; `hist`, `pscore`) -- and starts from the state that many draws ahead (pcg_advance).  After the round the samples are validated in order: sample j counts if and only
//     if the state it started from IS the state sample j-1 ended with -- then everything it computed is what
//     the sequential program computes -- and the first one that started elsewhere is thrown away together
//     with everything behind it and repeated in the next round, now from the right state.  The first lane
//     always starts from the validated state, so every round completes at least one sample.
// A round's radiances are added to the pixel's sum in sample order (imagetracer.py:97: cum_color += ...), one
// lane after the other through wave shuffles, so the sum is the sequential one bit for bit; rays are counted
// for validated samples only.  Per-pixel / per-sample seeds depend on the global pixel index alone: the image
// does not depend on how regions are cut into units or how many lanes a pixel gets.
// LAT: the second pass is built for few waves per SIMD; its time is set by chains of dependent steps:
// everything inline, registers no object.
#ifndef PT_REGIONS_INLINE
#define PT_REGIONS_INLINE 1  // second pass: HitRecord / scatter / transcendental code inline (1) or behind calls (0)
#endif
template <bool TILED, bool LDSF, bool LAT, bool SLDS = false, int LEAN = 0>
PT_DEV void path_trace(const PtKArgs &a) {
  constexpr bool INL = LAT && PT_REGIONS_INLINE;
  static_assert(!SLDS || INL, "the scene is staged in LDS for the second pass only");
  PathCtx w;
  int S, nsamp, N, W = 0, rows_local = 0, npass = 0, D = 0, rr = 0, diag_lds = -1, pcg_mode = PT_PCG_PIXEL;
  bool ortho = false;
  {
    pt_kargs c = cold_args(a);
    w.ws = c->ws;
    w.nthreads = (size_t)c->nthreads;
    w.stride = (size_t)c->frame_doubles * w.nthreads;
    w.gtid = blockIdx.x * PT_BLOCK + threadIdx.x;
    w.lds_frame = c->frame_doubles;
    w.lds_base = TILED ? 4 * c->npass : 0;  // behind the four waves' survivor masks (8-byte units)
    ortho = c->cam_kind != PT_CAMERA_PERSPECTIVE;
    diag_lds = c->diag_lds;
    pcg_mode = c->pcg_mode;
    S = c->S;
    N = c->N;
    W = c->W;
    rows_local = c->rows_local;
    npass = c->npass;
    D = c->D;
    rr = c->rr;
  }
  if (blockIdx.x == gridDim.x - 1) {  // the next frame's queue block (nothing of this frame reads it)
    unsigned long long *qn = pt_queue_next(a);
    for (int k = threadIdx.x; k < PT_QUEUE_WORDS; k += PT_BLOCK) qn[k] = 0ULL;
  }
  if (LAT && diag_lds >= 0) {
    // scale+translate records into LDS: world_query_lanes fetches them by lane-private index
    const unsigned long long *src = (const unsigned long long *)a.diag;
    for (int k = threadIdx.x; k < a.n_diag * 8; k += PT_BLOCK) pt_lds_masks[diag_lds + k] = src[k];
    __syncthreads();
  }
  int scene_lds = 0;
  if (SLDS) {
    // the shapes' records (128 B + 256 B each) into LDS: shading gathers ~20 values of the hit shape per lane, and a
    // gather from LDS costs a fraction of one through the vector memory path (8 waves of a CU share one of those)
    scene_lds = cold_args(a)->scene_lds;
    const unsigned long long *src = (const unsigned long long *)a.recs;
    for (int k = threadIdx.x; k < a.n_shapes * 16; k += PT_BLOCK) pt_lds_masks[scene_lds + k] = src[k];
    src = (const unsigned long long *)a.aux;
    for (int k = threadIdx.x; k < a.n_shapes * 32; k += PT_BLOCK) pt_lds_masks[scene_lds + a.n_shapes * 16 + k] = src[k];
    __syncthreads();
  }
  if (LAT) {  // the grid's occupancy bits into LDS: the cell walk of world_query_lanes reads one per step
    pt_kargs c = cold_args(a);
    const int occ_lds = c->grid_occ_lds;
    if (occ_lds >= 0) {
      const int nwords = (c->grid_res[0] * c->grid_res[1] * c->grid_res[2] + 31) / 32;
      unsigned *dst = (unsigned *)pt_lds_masks;
      for (int k = threadIdx.x; k < nwords; k += PT_BLOCK) dst[occ_lds + k] = c->grid_occ[k];
      __syncthreads();
    }
  }
  nsamp = S > 0 ? S * S : 1;
  const double invN = 1.0 / (double)N;
  const int lane = threadIdx.x & 63;
  const int mbase = (threadIdx.x >> 6) * npass;
  const int regions_x = (W + PT_REGION - 1) / PT_REGION;
  bool exhausted = false;           // !TILED: the global queue is empty
  bool first_unit = true;           // TILED (wave-uniform)
  // TILED: units of the frame (written by pt_unit_scatter before this kernel started; read once -- not from the heads' lines)
  const int n_units = TILED ? (int)pt_queue(a)[9] : 0;
  unsigned long long nrays = 0;

  // lane state.  mode 0: starts a sample at the next P-step; 1: inside a path (S-steps); 2: nothing to do;
  // 3 (TILED): sample finished, waits for the end of the round
  int mode = 2;
  long long pix = -1;
  Pcg pcg;
  pcg.state = 0;
  pcg.inc = 1;
  pcg.n = 0;
  int samp = 0, sp = 0, col = 0, grow = 0;
  V3 cum = {0.0, 0.0, 0.0};
  Ray ray;
  ray.o = {0.0, 0.0, 0.0};
  ray.d = {1.0, 0.0, 0.0};
  ray.tmin = 1e-5;
  // what shade() hands to the unwind loop of the same step: a value to deliver, or a child to spawn
  V3 ret = {0.0, 0.0, 0.0};
  bool spawn = false;
  V3 f_wp = {0.0, 0.0, 0.0}, f_n = {0.0, 0.0, 1.0}, f_in = {1.0, 0.0, 0.0};
  int f_brdf = 0;
  // TILED: the unit (wave-uniform) and this lane's place in it
  int L = 1;                        // lanes per pixel
  int leader = lane, jlane = 0;     // first lane of this lane's pixel; this lane's sample slot in a round
  bool in_unit = false;             // the lane belongs to a pixel of the unit
  int vbase = 0;                    // samples of the pixel validated so far (same in all lanes of the pixel)
  uint64_t vstate = 0;              // PT_PCG_PIXEL: generator state behind the last validated sample
  // PT_PCG_PIXEL: what the pixel's last eight validated samples drew, a byte each, the latest in the low byte.  The guess
  // for sample k is what sample k - S drew -- its neighbour one row up in the S x S grid of strata (imagetracer.py:86-93):
  // a pixel across an edge repeats its pattern of short and long paths row after row, where "what the last sample drew"
  // is wrong twice per row.  (S > 8: the sample before it.)
  uint64_t hist = 0;
  int pscore = 0;                   // ... how much more often the upper neighbour was the better guess than the predecessor (per pixel)
  const int hperiod = (S >= 1 && S <= 8) ? S : 1;
  uint64_t st_start = 0;            // state this lane's sample started from
  unsigned srays = 0, prays = 0;    // rays of the current sample; of the pixel's validated samples
  unsigned long long gpix = 0;      // global pixel index (seeds)

  // pixel coordinates + seeds + the sample's primary ray (imagetracer.py:86-97)
  auto start_sample = [&]() {
    pt_kargs c = cold_args(a);
    if (!TILED) {
      if (samp == 0) {
        pixel_coords(a, pix, col, grow);
        if (c->pcg_mode == PT_PCG_PIXEL) pcg_seed(pcg, c->s0, c->q0 + ((unsigned long long)grow * c->W + col));
      }
      if (c->pcg_mode == PT_PCG_SAMPLE)
        pcg_seed(pcg, c->s0, c->q0 + ((unsigned long long)grow * c->W + col) * (unsigned)nsamp + (unsigned)samp);
    }
    double up = 0.5, vp = 0.5;
    if (S > 0) {
      const int sr = samp / S, sc = samp - sr * S;
      up = ((double)sc + pcg_float(pcg)) / (double)S;
      vp = ((double)sr + pcg_float(pcg)) / (double)S;
    }
    ray = primary_ray(a, col, grow, up, vp);
  };

  // TILED: the generator a lane's next sample starts from, `samp` = vbase + jlane
  auto seed_round = [&]() {
    pt_kargs c = cold_args(a);
    if (pcg_mode == PT_PCG_SAMPLE)
      pcg_seed(pcg, c->s0, c->q0 + gpix * (unsigned)nsamp + (unsigned)samp);
    else
    {
      // (sample vbase + i: what its upper neighbour vbase + i - period drew, if the pixel has got that far and that guess
      //  has been the better one so far; else what the last validated sample drew)
      const int period = hperiod;
      unsigned ahead = (unsigned)jlane * ((unsigned)hist & 0xffu);
      if (pscore > 0) {
        ahead = 0;
        for (int i = 0; i < jlane; ++i)
          ahead += (unsigned)(hist >> (vbase + (i % period) >= period ? 8 * (period - 1 - (i % period)) : 0)) & 0xffu;
      }
      pcg.state = pcg_advance(vstate, pcg.inc, ahead);
    }
    pcg.n = 0;
    st_start = pcg.state;
    srays = 0;
  };

  // render.py:103-139 up to (not including) the recursion: sets `ret`, or pushes frame `sp` and asks
  // for child 0 (`spawn`).  `ray` is the ray that was queried, at depth `sp`.
  auto shade_hit = [&](auto rec, auto ax, double best_t) {
    V3 hc, em;
    double lum;
    Hit h;
    h.u = 0.0;
    h.v = 0.0;
    const bool uv = ax->needs_uv != 0;
    bool details = false;
    if (uv) {
      if constexpr (INL)
        hit_details<true>(rec, ax, ray, best_t, h, true);
      else
        hit_details_call(rec, ax, &ray, best_t, &h, true);
      details = true;
    }
    hc = brdf_pigment(a, ax, h.u, h.v);
    em = emitted_pigment(a, ax, h.u, h.v);
    lum = max2(max2(hc.x, hc.y), hc.z);
    if (sp >= rr) {  // render.py:116-123
      const double q = max2(0.05, 1.0 - lum);
      if (pcg_float(pcg) > q) {
        const double k = 1.0 / (1.0 - q);
        hc.x = hc.x * k;
        hc.y = hc.y * k;
        hc.z = hc.z * k;
      } else {
        ret = em;
        return;
      }
    }
    if (!(lum > 0.0)) {  // render.py:139 with cum_radiance = 0
      ret.x = em.x + 0.0 * invN;
      ret.y = em.y + 0.0 * invN;
      ret.z = em.z + 0.0 * invN;
      return;
    }
    if (sp + 1 > D) {
      // Every child of this hit would be beyond max_depth: the reference still calls scatter_ray for each
      // (consuming its draws: 2 for a diffuse BRDF, none for a mirror) and each child returns black at
      // render.py:100-101 without a world query.  No ray, no frame, no geometry is needed: advance the
      // generator and accumulate hit_color * 0 exactly as render.py:135-139 does.
      const bool diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
      V3 fc = {0.0, 0.0, 0.0};
      for (int i = 0; i < N; ++i) {
        if (diffuse) {
          pcg_next(pcg);
          pcg_next(pcg);
        }
        fc.x = fc.x + hc.x * 0.0;
        fc.y = fc.y + hc.y * 0.0;
        fc.z = fc.z + hc.z * 0.0;
      }
      ret.x = em.x + fc.x * invN;
      ret.y = em.y + fc.y * invN;
      ret.z = em.z + fc.z * invN;
      return;
    }
    // render.py:126-137: push the frame, child 0 is scattered at the next S-step
    if (!details) {
      if constexpr (INL)
        hit_details<true>(rec, ax, ray, best_t, h, false);
      else
        hit_details_call(rec, ax, &ray, best_t, &h, false);
    }
    ws_put<LDSF>(w, sp, 0, hc.x);
    ws_put<LDSF>(w, sp, 1, hc.y);
    ws_put<LDSF>(w, sp, 2, hc.z);
    ws_put<LDSF>(w, sp, 3, em.x);
    ws_put<LDSF>(w, sp, 4, em.y);
    ws_put<LDSF>(w, sp, 5, em.z);
    if (N > 1) {
      ws_put<LDSF>(w, sp, 6, 0.0);
      ws_put<LDSF>(w, sp, 7, 0.0);
      ws_put<LDSF>(w, sp, 8, 0.0);
      ws_put<LDSF>(w, sp, 9, 0.0);
      ws_put<LDSF>(w, sp, 10, h.wp.x);
      ws_put<LDSF>(w, sp, 11, h.wp.y);
      ws_put<LDSF>(w, sp, 12, h.wp.z);
      ws_put<LDSF>(w, sp, 13, h.n.x);
      ws_put<LDSF>(w, sp, 14, h.n.y);
      ws_put<LDSF>(w, sp, 15, h.n.z);
      ws_put<LDSF>(w, sp, 16, ray.d.x);
      ws_put<LDSF>(w, sp, 17, ray.d.y);
      ws_put<LDSF>(w, sp, 18, ray.d.z);
      ws_put<LDSF>(w, sp, 19, (double)ax->brdf_kind);
    }
    f_wp = h.wp;
    f_n = h.n;
    f_in = ray.d;
    f_brdf = ax->brdf_kind;
    sp++;
    spawn = true;
  };
  auto shade = [&](int hit, double best_t) {
    spawn = false;
    if (hit < 0) {  // render.py:103-105
      pt_kargs c = cold_args(a);
      ret.x = c->bg[0];
      ret.y = c->bg[1];
      ret.z = c->bg[2];
      return;
    }
    if constexpr (SLDS)
      shade_hit((pt_lds_rec)(const void *)(pt_lds_f64 + scene_lds) + hit,
                (pt_lds_aux)(const void *)(pt_lds_f64 + scene_lds + a.n_shapes * 16) + hit, best_t);
    else
      shade_hit(a.recs + hit, cold_args(a)->aux + hit, best_t);
  };

  // the primary call returned `ret`: one sample done (imagetracer.py:94-104)
  auto finish_sample = [&]() {
    if (TILED) {  // the radiance stays in `ret` until the round is validated
      mode = 3;
      return;
    }
    if (S > 0) {
      cum.x = cum.x + ret.x;
      cum.y = cum.y + ret.y;
      cum.z = cum.z + ret.z;
    } else {
      cum = ret;
    }
    mode = 0;
    if (++samp == nsamp) {
      if (S > 0) {
        const double k = 1.0 / (double)(S * S);
        cum.x = cum.x * k;
        cum.y = cum.y * k;
        cum.z = cum.z * k;
      }
      store_pixel(a, pix, cum);
      cum.x = 0.0;
      cum.y = 0.0;
      cum.z = 0.0;
      samp = 0;
      mode = 2;
    }
  };

#ifdef PT_DEBUG_TIME
  // section sums for every wave, plus a step-by-step trace of the wave that drew the first (fullest) unit
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
  bool tracing = false;
  int trace_n = 0;
  int ulog_seq = -1, ulog_rounds = 0, ulog_iters = 0;
  unsigned long long ulog_t[4] = {0, 0, 0, 0};
  unsigned long long dbg_q[3] = {0, 0, 0};
#define PT_STAMP(k)                                                                        \
  do {                                                                                     \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                            \
    tsum[k] += tn - tprev;                                                                 \
    const unsigned long long np_ = (unsigned long long)__popcll(__ballot(mode == 1));     \
    if (tracing && lane == 0 && trace_n < PT_TRACE_LEN)                                    \
      pt_trace[trace_n] = ((tn - tprev) << 16) | (np_ << 8) | (k);                         \
    if (tracing) trace_n++;                                                                \
    tprev = tn;                                                                            \
  } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif
  for (;;) {
    PT_STAMP(7);
    // (values that never flow from one iteration into the next: said explicitly, so that they hold no
    //  registers across the queries)
    spawn = false;
    f_wp = {0.0, 0.0, 0.0};
    f_n = {0.0, 0.0, 1.0};
    f_in = {1.0, 0.0, 0.0};
    f_brdf = 0;
    // ---- work for idle lanes ----
    if (TILED) {
      if (!__any(mode == 0 || mode == 1)) {
        if (__any(mode == 3)) {
          // ---- end of a round: validate the pixel's samples in order, add them up in order ----
          // (every lane of a pixel runs the same loop over the pixel's L lanes and ends with the same
          //  vbase / vstate / hist; only the values in the leader are used for the pixel's result)
          const bool fin = mode == 3;
          bool chain = true;
#ifdef PT_DEBUG_TIME
          ulog_rounds++;
          const int dbg_vbase0 = vbase;
          int dbg_fin = 0;
#endif
          for (int jj = 0; jj < L; ++jj) {
            const int src = (leader + jj) & 63;
            uint64_t s_from = 0, s_to = 0;  // (PT_PCG_SAMPLE validates nothing: five cross-lane reads less per turn)
            unsigned s_draws = 0;
            if (pcg_mode != PT_PCG_SAMPLE) {
              s_from = __shfl((unsigned long long)st_start, src, 64);
              s_to = __shfl((unsigned long long)pcg.state, src, 64);
              s_draws = (unsigned)__shfl((int)pcg.n, src, 64);
            }
            const int s_fin = __shfl((int)fin, src, 64);
            const unsigned s_rays = (unsigned)__shfl((int)srays, src, 64);
            const double rx_ = __shfl(ret.x, src, 64), ry_ = __shfl(ret.y, src, 64), rz_ = __shfl(ret.z, src, 64);
            chain = chain && s_fin != 0 && (pcg_mode == PT_PCG_SAMPLE || s_from == vstate);
#ifdef PT_DEBUG_TIME
            dbg_fin += s_fin;
#endif
            if (chain) {
              if (S > 0) {  // imagetracer.py:97
                cum.x = cum.x + rx_;
                cum.y = cum.y + ry_;
                cum.z = cum.z + rz_;
              } else {
                cum.x = rx_;
                cum.y = ry_;
                cum.z = rz_;
              }
              vstate = s_to;
              if (vbase >= hperiod)  // which guess would have been right for this sample: its upper neighbour's draws, or its predecessor's?
                pscore += (int)(((unsigned)(hist >> (8 * (hperiod - 1))) & 0xffu) == (s_draws & 0xffu)) - (int)(((unsigned)hist & 0xffu) == (s_draws & 0xffu));
              hist = (hist << 8) | (uint64_t)(s_draws & 0xffu);
              prays += s_rays;
              vbase++;
#ifdef PT_DEBUG_TIME
              if (tracing && in_unit && lane == leader && (leader / L) < 64 && vbase <= 80)
                pt_trace[PT_TRACE_LEN + (leader / L) * 80 + (vbase - 1)] = ((unsigned long long)ulog_rounds << 32) | ((unsigned long long)s_draws << 16) | ((unsigned long long)(leader / L) << 8) | 0xEEULL;
#endif
            }
          }
          mode = 2;
#ifdef PT_DEBUG_TIME
          {  // speculation statistics: pixel-rounds, samples traced, samples kept
            const bool lead = in_unit && lane == leader && pix >= 0;
            unsigned long long r4 = lead ? 1ULL : 0ULL, r5 = lead ? (unsigned long long)dbg_fin : 0ULL,
                               r6 = lead ? (unsigned long long)(vbase - dbg_vbase0) : 0ULL;
            for (int off = 32; off > 0; off >>= 1) {
              r4 += __shfl_down(r4, off, 64);
              r5 += __shfl_down(r5, off, 64);
              r6 += __shfl_down(r6, off, 64);
            }
            if (lane == 0) {
              unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
              wv[4] += r4;
              wv[5] += r5;
              wv[6] += r6;
            }
          }
#endif
          if (in_unit) {
            if (vbase >= nsamp) {
              if (lane == leader && pix >= 0) {  // imagetracer.py:99-104
                if (S > 0) {
                  const double k = 1.0 / (double)(S * S);
                  cum.x = cum.x * k;
                  cum.y = cum.y * k;
                  cum.z = cum.z * k;
                }
                store_pixel(a, pix, cum);
                nrays += prays;
              }
              pix = -1;  // this pixel is done (in every lane of it)
            } else {
              samp = vbase + jlane;
              if (samp < nsamp) {
                seed_round();
                mode = 0;
              }
            }
          }
        }
        PT_STAMP(3);
        if (!__any(mode == 0)) {
          // next unit for this wave, then its region's cone and survivor masks.  The sorted unit list is dealt out
          // to PT_UNIT_SHARDS shards (unit u belongs to shard u % shards: every shard the same mix of sizes) and a
          // workgroup pulls from shard blockIdx % shards only: a returning atomic on ONE head word saturates near 88
          // dequeues/us -- with thousands of waves pulling, queueing at the head costs more than a unit's work.  A
          // wave's first unit is its own rank in the shard (no atomic at all), later ones come from the shard's head,
          // one atomic by lane 0.
          unsigned uid = 0;
          const unsigned nsh = gridDim.x < PT_UNIT_SHARDS ? gridDim.x : PT_UNIT_SHARDS;  // (every shard needs a puller)
          const unsigned shard = blockIdx.x % nsh;
          if (first_unit) {
            uid = (blockIdx.x / nsh) * (PT_BLOCK / 64) + (threadIdx.x >> 6);
            first_unit = false;
          } else {
            const unsigned pullers = (gridDim.x - shard + nsh - 1) / nsh * (PT_BLOCK / 64);
#ifdef PT_DEBUG_TIME
            PT_VM_DRAIN();
            const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
#endif
            if (lane == 0) uid = pullers + (unsigned)atomicAdd(pt_queue(a) + PT_QUEUE_HEADS + 32 * shard, 1ULL);
#ifdef PT_DEBUG_TIME
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(uid) : "memory");
            lat_note(2, __builtin_amdgcn_s_memtime() - lt0);
#endif
          }
          uid = uid * nsh + shard;
          const int seq = (int)__builtin_amdgcn_readfirstlane((int)uid);
          PT_STAMP(6);
#ifdef PT_DEBUG_TIME
          tracing = seq == cold_args(a)->dbg_trace_unit;  // (or, below, the unit that starts at a given flagged pixel of a given region)
          if (LAT && tracing && lane == 0) {
            const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
            for (int q = 0; q < 3; ++q) dbg_q[q] = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane == 0 && ulog_seq >= 0 && ulog_seq < PT_UNITLOG_LEN) {  // close the log entry of the unit just finished
            pt_unitlog[ulog_seq * 8 + 1] = __builtin_amdgcn_s_memtime();
            pt_unitlog[ulog_seq * 8 + 2] = (unsigned long long)ulog_rounds | ((unsigned long long)ulog_iters << 32);
            pt_unitlog[ulog_seq * 8 + 4] = tsum[4] - ulog_t[0];
            pt_unitlog[ulog_seq * 8 + 5] = tsum[5] - ulog_t[1];
            pt_unitlog[ulog_seq * 8 + 6] = tsum[1] + tsum[2] - ulog_t[2];
            pt_unitlog[ulog_seq * 8 + 7] = tsum[0] - ulog_t[3];
          }
          ulog_t[0] = tsum[4];
          ulog_t[1] = tsum[5];
          ulog_t[2] = tsum[1] + tsum[2];
          ulog_t[3] = tsum[0];
          ulog_seq = seq;
          ulog_rounds = 0;
          ulog_iters = 0;
#endif
          pt_kargs ca = cold_args(a);
          if (seq >= n_units) break;
#ifdef PT_DEBUG_TIME
          if (lane == 0) pt_dbg_wave[(size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8 + 7] += 1ULL;
#endif
#ifdef PT_DEBUG_TIME
          const unsigned long long lt0 = __builtin_amdgcn_s_memtime();
          PT_VM_DRAIN();
          const unsigned long long lt1 = __builtin_amdgcn_s_memtime();
          const int4 unit = ca->units[seq];
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : "v"(unit.x) : "memory");
          lat_note(0, __builtin_amdgcn_s_memtime() - lt1);
          lat_note(1, lt1 - lt0);
#else
          const int4 unit = ca->units[seq];
#endif
          const int region = unit.x, first = unit.y & 0xff, count = (unit.y >> 8) & 0xff;
#ifdef PT_DEBUG_TIME
          if (cold_args(a)->dbg_trace_unit <= -2) tracing = (-2 - cold_args(a)->dbg_trace_unit) == region * 64 + first;  // (unit numbers vary from frame to frame)
#endif
          const unsigned long long todo = (unsigned long long)(unsigned)unit.z | ((unsigned long long)(unsigned)unit.w << 32);  // the region's flagged pixels
          const int ry = region / regions_x, rx = region - ry * regions_x;
          const int gr0 = global_row(a, ry * PT_REGION);
          const int gr1 = global_row(a, (ry * PT_REGION + PT_REGION - 1 < rows_local) ? ry * PT_REGION + PT_REGION - 1 : rows_local - 1);
          const TileCone tc = tile_cone(a, rx * PT_REGION, (rx * PT_REGION + PT_REGION < W) ? rx * PT_REGION + PT_REGION : W, gr0, gr1);
          __builtin_amdgcn_wave_barrier();
          for (int p = 0; p < npass; ++p) {
            const int slot = p * 64 + lane;
            bool keep = false;
            if (slot < a.n_shapes) keep = slot >= a.n_spheres || cone_keeps(tc, a.bounds[slot]);  // planes: always
            const unsigned long long m = __ballot(keep);
            if (lane == 0) pt_lds_masks[mbase + p] = m;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // lanes [p * L, (p + 1) * L) take the unit's p-th pixel = flagged pixel `first + p` of the region
          L = 64 / count;
          if (L > nsamp) L = nsamp;
#ifdef PT_DEBUG_TIME
          if (lane == 0 && seq < PT_UNITLOG_LEN) {
            pt_unitlog[seq * 8 + 0] = __builtin_amdgcn_s_memtime();
            pt_unitlog[seq * 8 + 3] = (unsigned long long)count | ((unsigned long long)L << 8) | ((unsigned long long)(blockIdx.x & 0x3ff) << 16) |
                                      ((unsigned long long)first << 26) | ((unsigned long long)region << 32);
          }
#endif
          const int pidx = lane / L;
          in_unit = pidx < count;
          leader = in_unit ? pidx * L : lane;
          jlane = lane - leader;
          mode = 2;
          pix = -1;
          if (in_unit) {
            const int bit = nth_set_bit(todo, first + pidx);
            pix = (long long)(ry * PT_REGION + (bit >> 3)) * W + (rx * PT_REGION + (bit & 7));
            pixel_coords(a, pix, col, grow);
            gpix = (unsigned long long)grow * ca->W + col;
            if (pcg_mode != PT_PCG_SAMPLE) {
              pcg_seed(pcg, ca->s0, ca->q0 + gpix);
              vstate = pcg.state;
            }
            hist = 0x0101010101010101ULL * (uint64_t)(ca->spec_draws & 0xff);
            pscore = 0;
            vbase = 0;
            prays = 0;
            cum.x = 0.0;
            cum.y = 0.0;
            cum.z = 0.0;
            samp = jlane;
            if (samp < nsamp) {
              seed_round();
              mode = 0;
            }
          }
        }
      }
    } else {
      const bool need = mode == 2 && !exhausted;
      if (__any(need)) {
        const long long np = next_pixel(a, need, a.npix);
        if (need) {
          if (np >= 0) {
            pix = np;
            mode = 0;
          }
        }
        exhausted = __any(need && np < 0);
      }
      if (!__any(mode != 2)) break;
    }

    PT_STAMP(0);
#ifdef PT_DEBUG_TIME
    ulog_iters++;
#endif
    const int n_start = __popcll(__ballot(mode == 0));
    const int n_path = __popcll(__ballot(mode == 1));
    if (n_start == 0 && n_path == 0) continue;  // TILED: nothing in flight, the round / unit logic above decides
    const bool do_p = n_start > 0 && n_path < cold_args(a)->p_max_path;
    const bool do_s = n_path >= cold_args(a)->s_min_path || (n_path > 0 && !do_p);

    // ---- queries: primary rays against the region's survivors, scattered rays against everything ----
    const bool prim = do_p && mode == 0;
    const bool scat = do_s && mode == 1;
    double best_t = INFINITY;
    int hit = -1;
    if (do_p) {
      if (prim) start_sample();
      PT_STAMP(1);
      double tp = INFINITY;
      int hp;
      if (TILED)
        hp = ortho ? world_query_tile<false, false, false>(a, ray, mbase, npass, tp, prim)
                   : world_query_tile<false, false, true>(a, ray, mbase, npass, tp, prim);
      else
        hp = world_query<false, false>(a, ray, INFINITY, tp, prim);
      if (prim) {
        hit = hp;
        best_t = tp;
      }
      PT_STAMP(2);
    }
    if (do_s) {
      double ts;
      const int hs = LAT ? world_query_lanes<false, LEAN>(a, ray, INFINITY, ts, scat, diag_lds) : world_query<false, false>(a, ray, INFINITY, ts, scat);
      if (scat) {
        hit = hs;
        best_t = ts;
      }
      PT_STAMP(4);
#ifdef PT_DEBUG_TIME
      if (LAT && tracing && lane == 0) {  // the traced unit: this query's prefilter cycles (8), walk cycles (9), walk turns (10)
        const unsigned long long *wv = pt_dbg_wave + (size_t)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6) * 8;
        for (int q = 0; q < 3; ++q) {
          const unsigned long long now = __hip_atomic_load(wv + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (trace_n < PT_TRACE_LEN) pt_trace[trace_n] = ((now - dbg_q[q]) << 16) | (unsigned long long)(8 + q);
          dbg_q[q] = now;
          trace_n++;
        }
      } else if (LAT && tracing) {
        trace_n += 3;
      }
#endif
    }

    // ---- shade the hit, then unwind: deliver radiance up the stack / scatter the next child, until
    //      this lane has a ray that needs a query (mode 1) or its sample is complete (mode 0 / 2 / 3) ----
    const bool work = prim || scat;
    if (work) {
      if (TILED)
        srays++;
      else
        nrays++;
      shade(hit, best_t);
      mode = 1;
    }
    bool unwinding = work;
    while (unwinding) {
      if (spawn) {
        // scatter_ray consumes its draws even when the child is beyond max_depth
        if (INL)
          ray = scatter_ray<true>(f_brdf, pcg, f_in, f_wp, f_n);
        else
          scatter_ray_call(f_brdf, &pcg, &f_in, &f_wp, &f_n, &ray);
        spawn = false;
        if (sp > D) {  // render.py:100-101: the child returns black without a world query
          ret.x = 0.0;
          ret.y = 0.0;
          ret.z = 0.0;
          continue;
        }
        break;  // mode 1: queried at the next S-step
      }
      if (sp == 0) {
        finish_sample();  // mode 0 (next sample), 2 (pixel done) or 3 (TILED: wait for the round's end)
        break;
      }
      // a child of frame sp-1 returned `ret` (render.py:135-137)
      const int fs = sp - 1;
      const V3 hc = {ws_get<LDSF>(w, fs, 0), ws_get<LDSF>(w, fs, 1), ws_get<LDSF>(w, fs, 2)};
      V3 fc = {0.0, 0.0, 0.0};
      int done = 0;
      if (N > 1) {
        fc.x = ws_get<LDSF>(w, fs, 6);
        fc.y = ws_get<LDSF>(w, fs, 7);
        fc.z = ws_get<LDSF>(w, fs, 8);
        done = (int)ws_get<LDSF>(w, fs, 9);
      }
      fc.x = fc.x + hc.x * ret.x;
      fc.y = fc.y + hc.y * ret.y;
      fc.z = fc.z + hc.z * ret.z;
      done++;
      if (done < N) {
        ws_put<LDSF>(w, fs, 6, fc.x);
        ws_put<LDSF>(w, fs, 7, fc.y);
        ws_put<LDSF>(w, fs, 8, fc.z);
        ws_put<LDSF>(w, fs, 9, (double)done);
        f_wp = {ws_get<LDSF>(w, fs, 10), ws_get<LDSF>(w, fs, 11), ws_get<LDSF>(w, fs, 12)};
        f_n = {ws_get<LDSF>(w, fs, 13), ws_get<LDSF>(w, fs, 14), ws_get<LDSF>(w, fs, 15)};
        f_in = {ws_get<LDSF>(w, fs, 16), ws_get<LDSF>(w, fs, 17), ws_get<LDSF>(w, fs, 18)};
        f_brdf = (int)ws_get<LDSF>(w, fs, 19);
        spawn = true;
        continue;
      }
      // render.py:139
      ret.x = ws_get<LDSF>(w, fs, 3) + fc.x * invN;
      ret.y = ws_get<LDSF>(w, fs, 4) + fc.y * invN;
      ret.z = ws_get<LDSF>(w, fs, 5) + fc.z * invN;
      sp = fs;
    }
    PT_STAMP(5);
  }
#ifdef PT_DEBUG_TIME
  if ((threadIdx.x & 63) == 0)
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
  pt_dbg_flush();
#endif
  add_ray_count(a, nrays);
}

// every pixel of the frame, pixels from one queue (orthogonal camera): throughput matters
template <bool LDSF>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_WAVES_PATH, 8))) void pt_path_kernel(const PtKArgs a) {
  path_trace<false, LDSF, false>(a);
}
// second pass behind pt_tile_kernel<PATHTRACER> (perspective camera): the flagged pixels, by region
#ifndef PT_WAVES_REGIONS
#define PT_WAVES_REGIONS 2
#endif
template <bool LDSF, bool SLDS = false, int LEAN = 0>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_WAVES_REGIONS, 8))) void pt_path_regions_kernel(const PtKArgs a) {
  path_trace<true, LDSF, true, SLDS, LEAN>(a);
}

// ---- PathTracer with num_of_rays > 1 (second pass behind pt_tile_kernel<PATHTRACER>): ONE pixel per wave, a node's children on lanes --
// render.py:126-139 runs the N children of a hit one after the other, each with its whole subtree, all drawing from one
// generator: where child k starts in the stream is known only when child k-1 has returned.  path_trace gives such a
// pixel one lane, which walks the tree ray by ray: up to sum N^d dependent steps (1 111 for the CLI's N = 10, D = 3)
// while a frame's worst pixel sets the launch time (profiles/r03_units_n10_before.log: 8.3 ms, 1 111 iterations).
// Here a wave owns a pixel and works on one NODE at a time (explicit stack of nodes, depth first, so the order of
// draws is the reference's): the node's next children are scattered and traced AT THE SAME TIME on different lanes, each
// from a SPECULATED generator state, and then committed in child order by comparing states -- child k counts iff the
// state it started from is the state child k-1 ended with, in which case everything it computed is what the sequential
// program computes; the first child that started elsewhere (and everything behind it) is simply done again in the next
// round from the right state.  Child 0 always starts right, so a round commits at least one child.
//   * A child that needs children of its own (hit, lum > 0, survived roulette, depth < D) is committed by pushing its
//     node; its later siblings wait for the state its subtree leaves behind.
//   * Ordinary families speculate a chain: child r starts r * cpred draws ahead, cpred = what the last committed child
//     without a subtree drew (initially: its scatter draws, plus the roulette draw where depth >= rr).
//   * LEAF families (children at depth D: traced, but THEIR children are beyond max_depth and only consume draws,
//     render.py:100-101) have few outcomes: c0 draws (hit and killed, black or specular surface) or c0 + 2N (a diffuse hit
//     that survives).  So child r is traced for EVERY start state it can have, r * c0 + b * 2N for b = 0..r: 55 lanes
//     settle ten leaves in one round whatever mix of outcomes they have.  (A miss draws c0 - 1: the chain then breaks
//     there and resumes next round -- slower, never wrong.)
// The sum a node keeps (cum_radiance += hit_color * child, render.py:137) is formed in child order, so the frame is
// the sequential one bit for bit; rays are counted for committed children only.
// LDS: per wave max(D, 1) node records of PT_TREE_FRAME doubles; of the innermost node hit_color, the running sum, the child
// counter and the BRDF kind are also kept in registers (wave-uniform).
#define PT_TREE_FRAME 20  // hc 0..2, em 3..5, cum 6..8, wp 9..11, n 12..14, in 15..17, brdf 18, next child 19
PT_DEV double rl_f64(double v, int lane) {  // v_readlane of a double (lane wave-uniform)
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
PT_DEV unsigned long long rl_u64(unsigned long long u, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
PT_DEV V3 rl_v3(V3 v, int lane) {
  V3 r = {rl_f64(v.x, lane), rl_f64(v.y, lane), rl_f64(v.z, lane)};
  return r;
}

template <bool SMALL>
PT_DEV void path_tree(const PtKArgs &a) {
  int S, nsamp, N, W, rows_local, npass, D, rr, diag_lds, pcg_mode, frames_lds;
  bool ortho;
  {
    pt_kargs c = cold_args(a);
    ortho = c->cam_kind != PT_CAMERA_PERSPECTIVE;
    diag_lds = c->diag_lds;
    pcg_mode = c->pcg_mode;
    S = c->S;
    N = c->N;
    W = c->W;
    rows_local = c->rows_local;
    npass = c->npass;
    D = c->D;
    rr = c->rr;
    frames_lds = 4 * c->npass + (int)(threadIdx.x >> 6) * (c->D > 1 ? c->D : 1) * PT_TREE_FRAME;  // (doubles)
  }
  if (blockIdx.x == gridDim.x - 1) {  // the next frame's queue block (nothing of this frame reads it)
    unsigned long long *qn = pt_queue_next(a);
    for (int k = threadIdx.x; k < PT_QUEUE_WORDS; k += PT_BLOCK) qn[k] = 0ULL;
  }
  if (diag_lds >= 0) {  // scale+translate records into LDS: world_query_lanes fetches them by lane-private index
    const unsigned long long *src = (const unsigned long long *)a.diag;
    for (int k = threadIdx.x; k < a.n_diag * 8; k += PT_BLOCK) pt_lds_masks[diag_lds + k] = src[k];
    __syncthreads();
  }
  {  // the grid's occupancy bits into LDS: the cell walk of world_query_lanes reads one per step
    pt_kargs c = cold_args(a);
    const int occ_lds = c->grid_occ_lds;
    if (occ_lds >= 0) {
      const int nwords = (c->grid_res[0] * c->grid_res[1] * c->grid_res[2] + 31) / 32;
      unsigned *dst = (unsigned *)pt_lds_masks;
      for (int k = threadIdx.x; k < nwords; k += PT_BLOCK) dst[occ_lds + k] = c->grid_occ[k];
      __syncthreads();
    }
  }
  nsamp = S > 0 ? S * S : 1;
  const double invN = 1.0 / (double)N;
  const int lane = threadIdx.x & 63;
  const int mbase = (threadIdx.x >> 6) * npass;
  const int regions_x = (W + PT_REGION - 1) / PT_REGION;
  const int n_units = (int)pt_queue(a)[9];
  bool first_unit = true;
  unsigned long long nrays = 0;  // (wave-uniform: committed rays of this wave's pixels)
  // small worlds: the wave-uniform loop over every shape (records through the scalar cache) has a shorter critical
  // path than per-lane candidate lists -- and a round's latency, not its throughput, is what a pixel's tree waits for
  const bool uniform_loop = a.n_shapes <= cold_args(a)->tree_uniform_max;
  const bool fuse_on = cold_args(a)->tree_fuse != 0;
  int b_last = N / 2;  // survivors of the last complete leaf family (wave-uniform): where the next one's guesses are centred
  // lane r of a leaf round: row = child offset in the round, col = hypothesis b (0..row); rows with row(row+1)/2 + row < 64
  int tri_row = 0;
  while ((tri_row + 1) * (tri_row + 2) / 2 <= lane) ++tri_row;
  const int tri_col = lane - tri_row * (tri_row + 1) / 2;
  int tri_rows = 0;  // rows that fit the wave: 10
  while ((tri_rows + 1) * (tri_rows + 2) / 2 <= 64) ++tri_rows;

  // node records live in LDS (frame d = the node at depth d of the current path through the tree); a wave's DS
  // operations execute in order, so a record written by one lane is what every lane reads afterwards
  auto frame = [&](int d) -> double * { return pt_lds_f64 + frames_lds + d * PT_TREE_FRAME; };
  auto rfl_f64 = [&](double v) -> double { return rl_f64(v, 0); };  // (a broadcast LDS read, made scalar)

  // what a lane found out about the ray it traced (render.py:103-139 up to the recursion)
  bool o_term = true;            // the call returns without children of its own
  V3 o_ret = {0.0, 0.0, 0.0};    // ... this value
  V3 o_hc = {0.0, 0.0, 0.0}, o_em = {0.0, 0.0, 0.0}, o_wp = {0.0, 0.0, 0.0}, o_n = {0.0, 0.0, 1.0};  // else: its node
  int o_brdf = 0;
  Pcg pcg;
  pcg.state = 0;
  pcg.inc = 1;
  pcg.n = 0;
  Ray ray;
  ray.o = {0.0, 0.0, 0.0};
  ray.d = {1.0, 0.0, 0.0};
  ray.tmin = 1e-5;
  auto shade_ray = [&](int hit, double best_t, int depth) {
    o_term = true;
    if (hit < 0) {  // render.py:103-105
      pt_kargs c = cold_args(a);
      o_ret = {c->bg[0], c->bg[1], c->bg[2]};
      return;
    }
    const PtShapeRec *rec = a.recs + hit;
    const PtShapeAux *ax = cold_args(a)->aux + hit;
    Hit h;
    h.u = 0.0;
    h.v = 0.0;
    bool details = false;
    if (ax->needs_uv != 0) {
      hit_details<true>(rec, ax, ray, best_t, h, true);
      details = true;
    }
    V3 hc = brdf_pigment(a, ax, h.u, h.v);
    const V3 em = emitted_pigment(a, ax, h.u, h.v);
    const double lum = max2(max2(hc.x, hc.y), hc.z);
    if (depth >= rr) {  // render.py:116-123
      const double q = max2(0.05, 1.0 - lum);
      if (pcg_float(pcg) > q) {
        const double k = 1.0 / (1.0 - q);
        hc.x = hc.x * k;
        hc.y = hc.y * k;
        hc.z = hc.z * k;
      } else {
        o_ret = em;
        return;
      }
    }
    if (!(lum > 0.0)) {  // render.py:139 with cum_radiance = 0
      o_ret = {em.x + 0.0 * invN, em.y + 0.0 * invN, em.z + 0.0 * invN};
      return;
    }
    if (depth + 1 > D) {  // every child is beyond max_depth: its scatter draws are consumed, it returns black (render.py:100-101)
      const bool diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
      V3 fc = {0.0, 0.0, 0.0};
      for (int i = 0; i < N; ++i) {
        if (diffuse) {
          pcg_next(pcg);
          pcg_next(pcg);
        }
        fc.x = fc.x + hc.x * 0.0;
        fc.y = fc.y + hc.y * 0.0;
        fc.z = fc.z + hc.z * 0.0;
      }
      o_ret = {em.x + fc.x * invN, em.y + fc.y * invN, em.z + fc.z * invN};
      return;
    }
    if (!details) hit_details<true>(rec, ax, ray, best_t, h, false);
    o_term = false;
    o_hc = hc;
    o_em = em;
    o_wp = h.wp;
    o_n = h.n;
    o_brdf = ax->brdf_kind;
  };

#ifdef PT_DEBUG_TIME
  // cycles of this wave in: 0 fetch + cull, 1 primary ray, 2 state jump + scatter, 3 scattered-ray query, 4 shade,
  // 5 commit, 6 node returns; 7: rounds
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
  unsigned long long dbg_leaf_rounds = 0, dbg_committed = 0, dbg_traced = 0, dbg_max_rounds = 0, dbg_fused = 0, dbg_fused_hit = 0;
#define PT_TT(k) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tsum[k] += tn - tprev; tprev = tn; } while (0)
#else
#define PT_TT(k) do { } while (0)
#endif
  for (;;) {
    // ---- next pixel: the unit list, one pixel per unit (see path_trace for the sharded heads) ----
    unsigned uid = 0;
    const unsigned nsh = gridDim.x < PT_UNIT_SHARDS ? gridDim.x : PT_UNIT_SHARDS;
    const unsigned shard = blockIdx.x % nsh;
    if (first_unit) {
      uid = (blockIdx.x / nsh) * (PT_BLOCK / 64) + (threadIdx.x >> 6);
      first_unit = false;
    } else {
      const unsigned pullers = (gridDim.x - shard + nsh - 1) / nsh * (PT_BLOCK / 64);
      if (lane == 0) uid = pullers + (unsigned)atomicAdd(pt_queue(a) + PT_QUEUE_HEADS + 32 * shard, 1ULL);
    }
    uid = uid * nsh + shard;
    const int seq = (int)__builtin_amdgcn_readfirstlane((int)uid);
    if (seq >= n_units) break;
    pt_kargs ca = cold_args(a);
    const int4 unit = ca->units[seq];
    const int region = __builtin_amdgcn_readfirstlane(unit.x), first = __builtin_amdgcn_readfirstlane(unit.y) & 0xff;
    const unsigned long long todo = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(unit.z) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(unit.w) << 32);
    const int ry = region / regions_x, rx = region - ry * regions_x;
    {  // the region's cone and survivor masks, for the primary rays
      const int gr0 = global_row(a, ry * PT_REGION);
      const int gr1 = global_row(a, (ry * PT_REGION + PT_REGION - 1 < rows_local) ? ry * PT_REGION + PT_REGION - 1 : rows_local - 1);
      const TileCone tc = tile_cone(a, rx * PT_REGION, (rx * PT_REGION + PT_REGION < W) ? rx * PT_REGION + PT_REGION : W, gr0, gr1);
      __builtin_amdgcn_wave_barrier();
      for (int p = 0; p < npass; ++p) {
        const int slot = p * 64 + lane;
        bool keep = false;
        if (slot < a.n_shapes) keep = slot >= a.n_spheres || cone_keeps(tc, a.bounds[slot]);  // planes: always
        const unsigned long long m = __ballot(keep);
        if (lane == 0) pt_lds_masks[mbase + p] = m;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int bit = nth_set_bit(todo, first);
    const long long pix = (long long)(ry * PT_REGION + (bit >> 3)) * W + (rx * PT_REGION + (bit & 7));
    int col, grow;
    pixel_coords(a, pix, col, grow);
    const unsigned long long gpix = (unsigned long long)grow * ca->W + col;
    // the pixel's generator (PT_PCG_PIXEL) -- wave-uniform: `gstate` is the state the sequential program is in
    unsigned long long gstate = 0, ginc = 1;
    if (pcg_mode != PT_PCG_SAMPLE) {
      Pcg g;
      pcg_seed(g, ca->s0, ca->q0 + gpix);
      gstate = g.state;
      ginc = g.inc;
    }
    V3 cum_pix = {0.0, 0.0, 0.0};
    unsigned long long prays = 0;
#ifdef PT_DEBUG_TIME
    unsigned long long dbg_rounds_pix = 0;
#endif
    PT_TT(0);
    for (int samp = 0; samp < nsamp; ++samp) {
      if (pcg_mode == PT_PCG_SAMPLE) {
        Pcg g;
        pcg_seed(g, ca->s0, ca->q0 + gpix * (unsigned)nsamp + (unsigned)samp);
        gstate = g.state;
        ginc = g.inc;
      }
      // ---- the sample's primary ray (imagetracer.py:86-97): lane 0 ----
      pcg.state = gstate;
      pcg.inc = ginc;
      pcg.n = 0;
      double up = 0.5, vp = 0.5;
      if (S > 0) {
        const int sr = samp / S, sc = samp - sr * S;
        up = ((double)sc + pcg_float(pcg)) / (double)S;
        vp = ((double)sr + pcg_float(pcg)) / (double)S;
      }
      ray = primary_ray(a, col, grow, up, vp);
      {
        double tp = INFINITY;
        // (an orthogonal camera's rays have no common origin: nothing is hoisted)
        const int hp = ortho ? world_query_tile<false, false, false>(a, ray, mbase, npass, tp, lane == 0)
                             : world_query_tile<false, false, true>(a, ray, mbase, npass, tp, lane == 0);
        shade_ray(hp, tp, 0);
      }
      prays += 1ULL;
      PT_TT(1);
      V3 sample_ret = rl_v3(o_ret, 0);
      gstate = rl_u64(pcg.state, 0);
      int sp = 0;  // nodes on the stack; the innermost one (frame sp - 1) is the node whose children are being traced
      // of that node, in registers (wave-uniform): hit_color, the sum of its children so far, how many are done, its BRDF
      V3 t_hc = {0.0, 0.0, 0.0}, t_cum = {0.0, 0.0, 0.0};
      int t_next = 0, t_brdf = 0;
      unsigned cpred = 0;
      // lane `src` traced a ray that needs children of its own: its node becomes frame sp (written by that lane itself)
      auto push_node = [&](int src, V3 in_dir) {
        if (sp > 0 && lane == 0) {  // the parent's running sum and child counter wait in its record
          double *f = frame(sp - 1);
          f[6] = t_cum.x; f[7] = t_cum.y; f[8] = t_cum.z;
          f[19] = (double)t_next;
        }
        if (lane == src) {
          double *f = frame(sp);
          f[0] = o_hc.x; f[1] = o_hc.y; f[2] = o_hc.z; f[3] = o_em.x; f[4] = o_em.y; f[5] = o_em.z;
          f[6] = 0.0; f[7] = 0.0; f[8] = 0.0; f[9] = o_wp.x; f[10] = o_wp.y; f[11] = o_wp.z;
          f[12] = o_n.x; f[13] = o_n.y; f[14] = o_n.z; f[15] = in_dir.x; f[16] = in_dir.y; f[17] = in_dir.z;
          f[18] = (double)o_brdf; f[19] = 0.0;
        }
        // (the record is read by every lane later on: the compiler may neither move those loads above this store nor
        //  feed them from this lane's registers)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        t_hc = rl_v3(o_hc, src);
        t_cum = {0.0, 0.0, 0.0};
        t_next = 0;
        t_brdf = __builtin_amdgcn_readlane(o_brdf, src);
        sp++;
      };
      // render.py:139 for the innermost node, then render.py:137 in its parent, which becomes the innermost one
      auto pop_node = [&]() -> V3 {
        const double *f = frame(sp - 1);
        const V3 val = {rfl_f64(f[3]) + t_cum.x * invN, rfl_f64(f[4]) + t_cum.y * invN, rfl_f64(f[5]) + t_cum.z * invN};
        sp--;
        if (sp > 0) {
          const double *g = frame(sp - 1);
          t_hc = {rfl_f64(g[0]), rfl_f64(g[1]), rfl_f64(g[2])};
          t_cum = {rfl_f64(g[6]), rfl_f64(g[7]), rfl_f64(g[8])};
          t_brdf = (int)rfl_f64(g[18]);
          t_next = (int)rfl_f64(g[19]);
          t_cum.x = t_cum.x + t_hc.x * val.x;
          t_cum.y = t_cum.y + t_hc.y * val.y;
          t_cum.z = t_cum.z + t_hc.z * val.z;
        }
        return val;
      };
      auto base_draws = [&]() -> unsigned {  // what a child of the innermost node draws when it needs no children: scatter + roulette
        return (t_brdf == PT_BRDF_DIFFUSE ? 2u : 0u) + (sp >= rr ? 1u : 0u);
      };
      if (!__builtin_amdgcn_readfirstlane((int)o_term)) {
        push_node(0, ray.d);
        cpred = base_draws();
      }
      // ---- the tree under the primary hit ----
      while (sp > 0) {
        const int remaining = N - t_next;
        if (remaining <= 0) {
          const V3 val = pop_node();
          if (sp == 0) {
            sample_ret = val;
            break;
          }
          cpred = base_draws();
          PT_TT(6);
          continue;
        }
        // ---- a round: children t_next .. of this node, depth sp, each from a speculated state ----
        const unsigned c0 = base_draws();
        const bool leaf = sp == D;  // (children of the children are beyond max_depth)
        int row, nrows;
        unsigned ahead;
        if (leaf) {
          nrows = remaining < tri_rows ? remaining : tri_rows;
          row = tri_row;
          ahead = (unsigned)tri_row * c0 + (unsigned)tri_col * 2u * (unsigned)N;
        } else {
          nrows = remaining < 64 ? remaining : 64;
          row = lane;
          ahead = (unsigned)lane * cpred;
        }
        bool act = row < nrows;
        // A WHOLE leaf family leaves lanes over (ten leaves: 55 of 64).  Where the family ends is known up to the number
        // b of its members that survive roulette on a diffuse surface -- N * c0 + 2N * b draws -- so the spare lanes
        // trace the NEXT sibling of this node (a child of its parent, one level up) from those states in the same round:
        // when the family commits in full and b is among the guesses, the sibling's ray is already traced when the node
        // returns, and a parent whose children all branch costs one round per child instead of two.
        const int leaf_lanes = nrows * (nrows + 1) / 2;
        bool fused = leaf && sp >= 2 && t_next == 0 && nrows == N && leaf_lanes < 64 && fuse_on;
        if (fused) fused = (int)rfl_f64(frame(sp - 2)[19]) < N;
        // (the spare lanes cover nh consecutive values of b around what the last complete family had)
        const int nh = (64 - leaf_lanes) < (N + 1) ? (64 - leaf_lanes) : (N + 1);
        int bmin = b_last - nh / 2;
        bmin = bmin < 0 ? 0 : (bmin > N + 1 - nh ? N + 1 - nh : bmin);
        const bool sib = fused && lane >= leaf_lanes && lane - leaf_lanes < nh;  // hypothesis b = bmin + lane - leaf_lanes
        if (sib) {
          act = true;
          row = -1;
          ahead = (unsigned)N * c0 + 2u * (unsigned)N * (unsigned)(bmin + lane - leaf_lanes);
        }
        // (what the last round found out is dead: said explicitly, so that it holds no registers across the query)
        o_term = true;
        o_ret = o_hc = o_em = o_wp = {0.0, 0.0, 0.0};
        o_n = {0.0, 0.0, 1.0};
        o_brdf = 0;
        pcg.state = act ? pcg_advance(gstate, ginc, ahead) : gstate;
        pcg.inc = ginc;
        pcg.n = 0;
        const unsigned long long st_start = pcg.state;
        {
          const double *f = frame(sib ? sp - 2 : sp - 1);  // the node the ray leaves from
          const V3 n_wp = {f[9], f[10], f[11]}, n_n = {f[12], f[13], f[14]}, n_in = {f[15], f[16], f[17]};
          ray = scatter_ray<true>((int)f[18], pcg, n_in, n_wp, n_n);  // materials.py:132-152, 175-196
        }
        PT_TT(2);
        double ts = INFINITY;
        int hs;
        if (uniform_loop)
          hs = world_query<false, false>(a, ray, INFINITY, ts, act);
        else
          hs = world_query_lanes<false, SMALL ? 1 : 0>(a, ray, INFINITY, ts, act, diag_lds);
        PT_TT(3);
        if (act) shade_ray(hs, ts, sib ? sp - 1 : sp);
        PT_TT(4);
#ifdef PT_DEBUG_TIME
        tsum[7] += 1;
        dbg_rounds_pix += 1;
        if (leaf) dbg_leaf_rounds += 1;
        dbg_traced += (unsigned long long)__popcll(__ballot(act));
#endif
        // ---- commit in child order ----
        unsigned long long expect = gstate;
        bool pushed = false;
        unsigned fam_draws = 0;
        // child `src` of the innermost node counts: add its value up, or put its node on the stack
        auto commit_child = [&](int src) {
          prays += 1ULL;
          t_next++;
          expect = rl_u64(pcg.state, src);
          if (__builtin_amdgcn_readlane((int)o_term, src)) {
            const V3 val = rl_v3(o_ret, src);
            t_cum.x = t_cum.x + t_hc.x * val.x;  // render.py:137
            t_cum.y = t_cum.y + t_hc.y * val.y;
            t_cum.z = t_cum.z + t_hc.z * val.z;
            cpred = (unsigned)__builtin_amdgcn_readlane((int)pcg.n, src);
            fam_draws += cpred;
          } else {  // the child has children of its own: its node goes on the stack, the siblings wait
            push_node(src, ray.d);
            pushed = true;
          }
        };
        for (int r = 0; r < nrows && !pushed; ++r) {
          const unsigned long long m = __ballot(act && row == r && st_start == expect);
          if (!m) break;  // nobody traced child r from the right state: next round
          commit_child(__ffsll((long long)m) - 1);
        }
        if (leaf && t_next == N && nrows == N && fam_draws >= (unsigned)N * c0)
          b_last = (int)((fam_draws - (unsigned)N * c0) / (2u * (unsigned)N));
#ifdef PT_DEBUG_TIME
        if (fused) dbg_fused += 1;
#endif
        if (fused && t_next == N) {
          // the leaf family is complete: its node returns now, and its parent's next child may be there already
          (void)pop_node();
          cpred = base_draws();
          const unsigned long long m = __ballot(sib && st_start == expect);
          if (m) commit_child(__ffsll((long long)m) - 1);
#ifdef PT_DEBUG_TIME
          if (m) dbg_fused_hit += 1;
#endif
        }
        gstate = expect;
        if (pushed) cpred = base_draws();
        PT_TT(5);
      }
      // imagetracer.py:94-97
      if (S > 0) {
        cum_pix.x = cum_pix.x + sample_ret.x;
        cum_pix.y = cum_pix.y + sample_ret.y;
        cum_pix.z = cum_pix.z + sample_ret.z;
      } else {
        cum_pix = sample_ret;
      }
    }
    if (S > 0) {  // imagetracer.py:99-101
      const double k = 1.0 / (double)(S * S);
      cum_pix.x = cum_pix.x * k;
      cum_pix.y = cum_pix.y * k;
      cum_pix.z = cum_pix.z * k;
    }
    if (lane == 0) store_pixel(a, pix, cum_pix);
    nrays += prays;
#ifdef PT_DEBUG_TIME
    dbg_committed += prays;
    if (dbg_rounds_pix > dbg_max_rounds) dbg_max_rounds = dbg_rounds_pix;
#endif
  }
#ifdef PT_DEBUG_TIME
  if (lane == 0) {
    for (int q = 0; q < 8; ++q) atomicAdd(pt_queue(a) + 1 + q, tsum[q]);
    atomicAdd(pt_queue(a) + 12, dbg_leaf_rounds | (dbg_fused << 24) | (dbg_fused_hit << 44));
    atomicAdd(pt_queue(a) + 13, dbg_committed);
    atomicAdd(pt_queue(a) + 14, dbg_traced);
    atomicMax(pt_queue(a) + 15, dbg_max_rounds);
  }
#endif
  add_ray_count(a, lane == 0 ? nrays : 0ULL);
}

#ifndef PT_TREE_WAVES
#define PT_TREE_WAVES 2
#endif
template <bool SMALL = false>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_TREE_WAVES, 8))) void pt_path_tree_kernel(const PtKArgs a) {
  path_tree<SMALL>(a);
}

// ---- culling probe: cone_keeps / pixel_cone exactly as the render kernels evaluate them, one wave ---------------
__global__ void pt_cull_probe_kernel(const PtKArgs a, int x0, int x1, int row0, int row1, int pixel_x, int pixel_row,
                                     int *keep) {
  const ConeCam cam = cone_cam(a);
  const TileCone tile = tile_cone(cam, x0, x1, row0, row1);
  const TileCone tc = pixel_x >= 0 ? pixel_cone(cam, tile, pixel_x, pixel_row) : tile;
  for (int slot = threadIdx.x; slot < a.n_shapes; slot += 64)
    keep[a.recs[slot].index] = slot >= a.n_spheres ? 1 : (cone_keeps(tc, a.bounds[slot]) ? 1 : 0);
}

// ---- hit-record probe (include/ptrace_debug.h): world_query + hit_details for caller-supplied rays ----------------
// One wave-uniform shape at a time is expressed as a one-record view of the tables (the records are grouped
// [scale+translate spheres | other spheres | planes]); shape_index < 0: the whole world.
__global__ void pt_hit_probe_kernel(const PtKArgs a, int shape_index, const double *rays, int n, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  const double *rp = rays + (size_t)(active ? i : 0) * 8;
  Ray r;
  r.o = {rp[0], rp[1], rp[2]};
  r.d = {rp[3], rp[4], rp[5]};
  r.tmin = rp[6];
  const double tmax = rp[7];
  PtKArgs v = a;
  int slot0 = 0;
  if (shape_index >= 0) {
    for (int s = 0; s < a.n_shapes; ++s)
      if (a.recs[s].index == shape_index) slot0 = s;
    v.recs = a.recs + slot0;
    v.diag = a.diag + (slot0 < a.n_diag ? slot0 : 0);
    v.n_diag = slot0 < a.n_diag ? 1 : 0;
    v.n_spheres = slot0 < a.n_spheres ? 1 : 0;
    v.n_shapes = 1;
  }
  double t = INFINITY;
  const int hit = world_query<false, false>(v, r, tmax, t, active);
  if (!active) return;
  double *o = out + (size_t)i * 12;
  for (int k = 0; k < 12; ++k) o[k] = 0.0;
  if (hit < 0) return;
  const int slot = slot0 + hit;
  Hit h;
  hit_details(a.recs + slot, a.aux + slot, r, t, h, true);
  o[0] = 1.0;
  o[1] = t;
  o[2] = h.wp.x; o[3] = h.wp.y; o[4] = h.wp.z;
  o[5] = h.n.x; o[6] = h.n.y; o[7] = h.n.z;
  o[8] = h.u; o[9] = h.v;
  o[10] = (double)a.recs[slot].index;
}

// ---- the scattered / shadow rays' query on its own: candidates from the conservative fp32 filter (or the grid walk),
// exact visits.  out: n x 4 doubles (hit 0/1, t, World.shapes index, 0); ANYHIT: (blocked 0/1, 0, 0, 0).  The 64 rays of a
// workgroup run as one wave, as in the renderers; a ray with tmin < 0 is an idle lane.
template <bool ANYHIT>
__global__ void pt_lanes_probe_kernel(const PtKArgs a, const double *rays, int n, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double *rp = rays + (size_t)(i < n ? i : 0) * 8;
  Ray r;
  r.o = {rp[0], rp[1], rp[2]};
  r.d = {rp[3], rp[4], rp[5]};
  r.tmin = rp[6];
  const double tmax = rp[7];
  const bool active = i < n && !(r.tmin < 0.0);
  double t = INFINITY;
  const int hit = world_query_lanes<ANYHIT>(a, r, tmax, t, active, -1);
  if (i >= n) return;
  double *o = out + (size_t)i * 4;
  o[0] = (active && hit >= 0) ? 1.0 : 0.0;
  o[1] = (!ANYHIT && active && hit >= 0) ? t : 0.0;
  o[2] = (!ANYHIT && active && hit >= 0) ? (double)a.recs[hit].index : 0.0;
  o[3] = 0.0;
}

// ---- camera probe: primary_ray for caller-supplied (col, row, u_pixel, v_pixel) -------------------------------------
__global__ void pt_camera_probe_kernel(const PtKArgs a, const double *pix, int n, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Ray r = primary_ray(a, (int)pix[4 * i], (int)pix[4 * i + 1], pix[4 * i + 2], pix[4 * i + 3]);
  double *o = out + (size_t)i * 7;
  o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.tmin;
}

// ---- scatter probe: scatter_ray (both forms the kernels use: behind a call, and inline) ----------------------------
__global__ void pt_scatter_probe_kernel(const double *in, int n, double *out, unsigned long long *state_after) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *q = in + (size_t)i * 12;
  Pcg p, p2;
  pcg_seed(p, (uint64_t)q[1], (uint64_t)q[2]);
  p2 = p;
  const V3 nrm = {q[3], q[4], q[5]}, inc = {q[6], q[7], q[8]}, pt = {q[9], q[10], q[11]};
  Ray r;
  scatter_ray_call((int)q[0], &p, &inc, &pt, &nrm, &r);
  const Ray r2 = scatter_ray<true>((int)q[0], p2, inc, pt, nrm);
  double *o = out + (size_t)i * 7;
  o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.tmin;
  // (the inline form must agree with the out-of-line one to the bit: same source, same flags)
  const bool same = r2.o.x == r.o.x && r2.o.y == r.o.y && r2.o.z == r.o.z && r2.d.x == r.d.x && r2.d.y == r.d.y && r2.d.z == r.d.z &&
                    r2.tmin == r.tmin && p2.state == p.state;
  state_after[i] = same ? p.state : ~0ULL;
}

// ---- primitive probe: lets the tests check IEEE exactness of device sqrt / div and measure the ulp
//      distance of ocml's transcendental functions from glibc's ----------------------------
__global__ void pt_probe_kernel(int op, const double *x, const double *y, double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (op) {
    case 0: r = sqrt(x[i]); break;
    case 1: r = x[i] / y[i]; break;
    case 2: r = sin(x[i]); break;
    case 3: r = cos(x[i]); break;
    case 4: r = atan2(x[i], y[i]); break;
    case 5: r = acos(x[i]); break;
    case 6: r = floor(x[i]); break;
    case 7: r = x[i] * y[i] + x[i]; break;  // must NOT be fused (-ffp-contract=off)
    case 8:    // pcg.py:23-62: the (int)y[i]-th output of PCG(init_state = 45, init_seq = x[i]), as a double
    case 9: {  // ... and the matching random_float()
      Pcg p;
      pcg_seed(p, 45ULL, (uint64_t)x[i]);
      uint32_t v = 0;
      double f = 0.0;
      for (int k = 0; k <= (int)y[i]; ++k) {
        if (op == 8)
          v = pcg_next(p);
        else
          f = pcg_float(p);
      }
      r = op == 8 ? (double)v : f;
      break;
    }
    case 10: {  // pcg_advance(state, inc, n) == n calls of pcg_next: 1.0 when the states agree (n = y[i])
      Pcg p, q;
      pcg_seed(p, 45ULL, (uint64_t)x[i]);
      q = p;
      const unsigned nsteps = (unsigned)y[i];
      for (unsigned k = 0; k < nsteps; ++k) pcg_next(p);
      r = (pcg_advance(q.state, q.inc, nsteps) == p.state) ? 1.0 : 0.0;
      break;
    }
    default: break;
  }
  out[i] = r;
}
