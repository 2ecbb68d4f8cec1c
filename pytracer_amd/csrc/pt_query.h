// pt_query.h -- World.ray_intersection: the wave-uniform shape loop and the per-lane candidate lists of scattered / shadow rays.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
#define PT_UNITLOG_LEN 16384
// per work unit of the path tracer's second pass: start tick, end tick, rounds | iterations << 32, pixels | lanes per pixel << 8 |
// workgroup << 16, then the unit's cycles in: scattered-ray queries, shade, sample start + primary query, commit + fetch
// (pt_tile4_kernel: per wave its cycles, the per-tile part of them, its end tick)
__device__ unsigned long long pt_unitlog[PT_UNITLOG_LEN * 8];
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
