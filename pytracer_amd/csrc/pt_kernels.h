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
// Execution model (MI355X): 1 lane = 1 pixel (grid-stride), wave64, 256-thread workgroups.  The
// shape loop index is wave-uniform, so shape records are fetched through the scalar cache into
// SGPRs (one s_load per record per wave) and every v_mul_f64 takes its matrix element as an SGPR
// operand: VGPRs hold only the ray.  The path tracer is a per-lane state machine whose ONLY
// convergent hot loop is the shape loop: a lane that finishes a path immediately starts its next
// sample, so lanes stay busy until their pixel's S*S samples are done (no per-bounce tail).
// MFMA is not used (no dense contraction on this path); the bound is fp64 VALU issue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptrace.h"
#include "pt_layout.h"

#define PT_DEV static __device__ __forceinline__
#define PT_PI 3.141592653589793
#define PT_BLOCK 256

// Uniform (wave-invariant) reads go through the constant address space so the backend emits
// s_load_* (scalar cache -> SGPRs) instead of per-lane global loads.
typedef const __attribute__((address_space(4))) double *pt_kdouble;
typedef const __attribute__((address_space(4))) int32_t *pt_kint;
#define PT_KD(p) ((pt_kdouble)(const void *)(p))
#define PT_KI(p) ((pt_kint)(const void *)(p))

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
};
PT_DEV uint32_t pcg_next(Pcg &p) {
  const uint64_t old = p.state;
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
PT_DEV double max2(double a, double b) { return (b > a) ? b : a; }  // Python max(a, b)

// ---- the shape loop: World.ray_intersection (world.py:51-69) ---------------------------------------
// Returns the record slot of the closest shape hit in (r.tmin, best_t) or -1; best_t is updated.
// Records are grouped (spheres first, then planes) so each loop body is branch-free on the shape
// kind; a tie in t between a plane and an earlier winner is resolved by the original list index,
// which reproduces "first shape in list order wins" (world.py:62, strict <).
// ANYHIT: leave as soon as every active lane has some hit (OnOff, shadow rays) — the hit/miss
// answer is identical, only `which` shape is unspecified.
// HOIST: primary rays of a perspective camera share their origin, so invm*origin and c=|o'|^2-1
// are per-shape constants (a.hoist), computed in the same operation order by pt_prep_hoist.
template <bool ANYHIT, bool HOIST>
PT_DEV int world_query(const PtKArgs &a, const Ray &r, double &best_t, bool active) {
  int best = -1;
  const double tmin = r.tmin;
  const int ns = a.n_spheres;
  const int n = a.n_shapes;
  // ---- spheres: shapes.py:102-121 (54 flop generic, 30 hoisted, + sqrt and 1-2 div on a hit) ----
  for (int i = 0; i < ns; ++i) {
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
    const double bb = 2.0 * (ox * dx + oy * dy + oz * dz);
    const double delta = bb * bb - 4.0 * aa * cc;
    if (active && delta > 0.0) {
      // first root inside (tmin, tmax); the roots are ordered (a > 0), so using the running
      // best_t as the upper limit selects the same winner as world.py:62
      const double sd = sqrt(delta);
      const double den = 2.0 * aa;
      double t = (-bb - sd) / den;
      bool ok = (t > tmin) && (t < best_t);
      if (!ok) {
        t = (-bb + sd) / den;
        ok = (t > tmin) && (t < best_t);
      }
      if (ok) {
        best_t = t;
        best = i;
      }
    }
    if (ANYHIT) {
      if (__ballot(active && best < 0) == 0ULL) return best;
    }
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
    if (active && !(fabs(dz) < 1e-5)) {
      const double t = -oz / dz;
      if (!(t <= tmin)) {
        bool take = t < best_t;
        if (!ANYHIT && t == best_t && best >= 0) {
          // exact tie with an earlier winner: the lower World.shapes index wins
          take = *PT_KI(&a.recs[i].index) < a.recs[best].index;
        }
        if (take) {
          best_t = t;
          best = i;
        }
      }
    }
    if (ANYHIT) {
      if (__ballot(active && best < 0) == 0ULL) return best;
    }
  }
  return best;
}

// ---- the closest hit's HitRecord (shapes.py:123-131, 177-189; world.py:66-67) ----------------------
// Computed once per ray for the winner only; every value is a pure function of (ray, shape, t), so
// it equals what the reference computed for that candidate.
PT_DEV void hit_details(const PtKArgs &a, const Ray &r, double t, int i, Hit &h, bool need_uv) {
  const PtShapeRec *rec = a.recs + i;
  const PtShapeAux *ax = a.aux + i;  // same (grouped) slot order as recs
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
      const double uu = atan2(hp.y, hp.x) / (2.0 * PT_PI);
      h.u = (uu >= 0.0) ? uu : uu + 1.0;
      double z = hp.z;  // the reference raises ValueError outside [-1, 1]: clamp
      z = (z > 1.0) ? 1.0 : ((z < -1.0) ? -1.0 : z);
      h.v = acos(z) / PT_PI;
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

// ---- pigments (materials.py:50-100) --------------------------------------------------------------------
PT_DEV V3 pigment_color(const PtKArgs &a, int kind, const double *c1, const double *c2, double steps,
                        int tex, double u, double v) {
  const double *c = c1;
  if (kind == PT_PIGMENT_CHECKERED) {
    const long long iu = (long long)floor(u * steps);
    const long long iv = (long long)floor(v * steps);
    // Python's % 2 is non-negative; (x & 1) is the same parity for negative x in two's complement
    c = ((iu & 1LL) == (iv & 1LL)) ? c1 : c2;
  } else if (kind == PT_PIGMENT_IMAGE) {
    const int w = a.tex[tex].w, hh = a.tex[tex].h;
    long long col = (long long)(u * (double)w);  // int() truncates toward zero
    long long row = (long long)(v * (double)hh);
    if (col >= w) col = w - 1;
    if (row >= hh) row = hh - 1;
    c = a.tex_data + a.tex[tex].offset + (row * w + col) * 3;
  }
  V3 r = {c[0], c[1], c[2]};
  return r;
}
PT_DEV V3 brdf_pigment(const PtKArgs &a, const PtShapeAux *ax, double u, double v) {
  return pigment_color(a, ax->pig_kind, ax->pig_c1, ax->pig_c2, ax->pig_steps, ax->pig_tex, u, v);
}
PT_DEV V3 emitted_pigment(const PtKArgs &a, const PtShapeAux *ax, double u, double v) {
  return pigment_color(a, ax->emi_kind, ax->emi_c1, ax->emi_c2, ax->emi_steps, ax->emi_tex, u, v);
}

// ---- BRDF.scatter_ray (materials.py:132-152, 175-196; geometry.py:247-262) -------------------------
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
    const double cp = cos(phi), sp = sin(phi);
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

// ---- ImageTracer.fire_ray + Camera.fire_ray (imagetracer.py:48-58; camera.py:59-78, 103-124) -----
PT_DEV Ray primary_ray(const PtKArgs &a, int col, int row, double up, double vp) {
  const double u = ((double)col + up) / (double)a.W;
  const double v = 1.0 - ((double)row + vp) / (double)a.H;
  V3 o, d;
  if (a.cam_kind == PT_CAMERA_PERSPECTIVE) {
    o.x = -a.cam_dist;
    o.y = 0.0;
    o.z = 0.0;
    d.x = a.cam_dist;
    d.y = (1.0 - 2.0 * u) * a.cam_aspect;
    d.z = 2.0 * v - 1.0;
  } else {
    o.x = -1.0;
    o.y = (1.0 - 2.0 * u) * a.cam_aspect;
    o.z = 2.0 * v - 1.0;
    d.x = 1.0;
    d.y = 0.0;
    d.z = 0.0;
  }
  Ray r;
  r.o = xf_point(a.cam_m, o);
  r.d = xf_vec(a.cam_m, d);
  r.tmin = 1.0e-5;
  return r;
}

// local (rank-compact) pixel index -> column and GLOBAL row (pt_params partition)
PT_DEV void pixel_coords(const PtKArgs &a, long long pix, int &col, int &grow) {
  const int lr = (int)(pix / a.W);
  col = (int)(pix - (long long)lr * a.W);
  const int blk = lr / a.row_block;
  grow = (blk * a.n_ranks + a.rank) * a.row_block + (lr - blk * a.row_block);
}

PT_DEV void store_pixel(const PtKArgs &a, long long pix, V3 c) {
  if (a.out_f32) {
    float *o = (float *)a.out + pix * 3;
    o[0] = (float)c.x;
    o[1] = (float)c.y;
    o[2] = (float)c.z;
  } else {
    double *o = (double *)a.out + pix * 3;
    o[0] = c.x;
    o[1] = c.y;
    o[2] = c.z;
  }
}

PT_DEV void add_ray_count(const PtKArgs &a, unsigned long long n) {
  if (a.ray_counter) {
    // wave reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.ray_counter, n);
  }
}

// ---- pt_prep_hoist: per-shape constants of the primary rays (perspective camera) ----------------------
__global__ void pt_prep_hoist(const PtShapeRec *recs, PtHoist *hoist, int n, V3 origin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 o = xf_point(recs[i].invm, origin);
  PtHoist h;
  h.ox = o.x;
  h.oy = o.y;
  h.oz = o.z;
  h.c = (o.x * o.x + o.y * o.y + o.z * o.z) - 1.0;
  hoist[i] = h;
}

// ---- OnOff / Flat / PointLight: one world query per sample (+ shadow rays) ----------------------------
template <int RENDERER, bool HOIST>
__global__ __launch_bounds__(PT_BLOCK) void pt_simple_kernel(const PtKArgs a) {
  const int S = a.S;
  const int nsamp = S > 0 ? S * S : 1;
  const V3 bg = {a.bg[0], a.bg[1], a.bg[2]};
  unsigned long long nrays = 0;
  for (long long base = (long long)blockIdx.x * PT_BLOCK; base < a.npix; base += a.nthreads) {
    const long long pix = base + threadIdx.x;
    const bool active = pix < a.npix;
    int col = 0, grow = 0;
    if (active) pixel_coords(a, pix, col, grow);
    const unsigned long long gpix = (unsigned long long)grow * a.W + col;
    Pcg pcg;
    if (S > 0 && a.pcg_mode == PT_PCG_PIXEL) pcg_seed(pcg, a.s0, a.q0 + gpix);
    V3 cum = {0.0, 0.0, 0.0};
    for (int s = 0; s < nsamp; ++s) {
      double up = 0.5, vp = 0.5;
      if (S > 0) {  // imagetracer.py:86-93: u drawn first, then v; sub_row outer, sub_col inner
        if (a.pcg_mode == PT_PCG_SAMPLE) pcg_seed(pcg, a.s0, a.q0 + gpix * (unsigned)nsamp + (unsigned)s);
        const int sr = s / S, sc = s - sr * S;
        up = ((double)sc + pcg_float(pcg)) / (double)S;
        vp = ((double)sr + pcg_float(pcg)) / (double)S;
      }
      const Ray ray = primary_ray(a, col, grow, up, vp);
      double best_t = INFINITY;
      const int hit = world_query<RENDERER == PT_RENDERER_ONOFF, HOIST>(a, ray, best_t, active);
      if (active) nrays++;
      V3 c = bg;
      if (RENDERER == PT_RENDERER_ONOFF) {  // render.py:52-53
        if (hit >= 0) {
          c.x = a.onoff[0];
          c.y = a.onoff[1];
          c.z = a.onoff[2];
        }
      } else if (RENDERER == PT_RENDERER_FLAT) {  // render.py:65-74
        if (hit >= 0) {
          const PtShapeAux *ax = a.aux + hit;
          Hit h;
          h.u = 0.0;
          h.v = 0.0;
          // Flat needs only (u, v); skip the whole HitRecord when both pigments are uniform
          if (ax->needs_uv) hit_details(a, ray, best_t, hit, h, true);
          const V3 p1 = brdf_pigment(a, ax, h.u, h.v);
          const V3 p2 = emitted_pigment(a, ax, h.u, h.v);
          c.x = p1.x + p2.x;
          c.y = p1.y + p2.y;
          c.z = p1.z + p2.z;
        }
      } else {  // PointLight, render.py:157-193
        const bool lit = active && hit >= 0;
        Hit h;
        const PtShapeAux *ax = a.aux + (hit >= 0 ? hit : 0);
        V3 res = bg;
        if (lit) {
          hit_details(a, ray, best_t, hit, h, ax->needs_uv != 0);
          const V3 em = emitted_pigment(a, ax, h.u, h.v);
          res.x = a.ambient[0] + em.x;
          res.y = a.ambient[1] + em.y;
          res.z = a.ambient[2] + em.z;
        }
        for (int l = 0; l < a.n_lights; ++l) {
          pt_kdouble L = PT_KD(&a.lights[l]);
          const V3 lp = {L[0], L[1], L[2]};
          // world.py:71-80: shadow ray from the hit point towards the light, any-hit in (1e-2/|d|, 1)
          Ray sh;
          sh.o = lit ? h.wp : lp;
          sh.d.x = lp.x - sh.o.x;
          sh.d.y = lp.y - sh.o.y;
          sh.d.z = lp.z - sh.o.z;
          const double dn = sqrt(sh.d.x * sh.d.x + sh.d.y * sh.d.y + sh.d.z * sh.d.z);
          sh.tmin = 1e-2 / dn;
          double tlim = 1.0;
          const int blocked = world_query<true, false>(a, sh, tlim, lit);
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
              const double th_in = acos(dot3(normalize3(h.n), normalize3(in_dir)));
              const double th_out = acos(dot3(normalize3(h.n), normalize3(out_dir)));
              if (fabs(th_in - th_out) < ax->brdf_param) bc = brdf_pigment(a, ax, h.u, h.v);
            }
            res.x = res.x + bc.x * L[3] * cos_theta * df;
            res.y = res.y + bc.y * L[4] * cos_theta * df;
            res.z = res.z + bc.z * L[5] * cos_theta * df;
          }
        }
        c = res;
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
  add_ray_count(a, nrays);
}

// ---- PathTracer (render.py:99-139) as a per-lane state machine ----------------------------------------
// The reference recursion is depth-first; frame `k` of the explicit stack is the call at depth k.
// Frame fields (in a.ws, [slot][field][thread] so a wave's accesses are contiguous):
//   0..2 hit_color (after Russian roulette)   3..5 emitted
//   N > 1 only: 6..8 cum_radiance, 9 children done, 10..12 hit point, 13..15 normal,
//               16..18 incoming direction, 19 brdf kind
PT_DEV double &ws_at(const PtKArgs &a, int slot, int field, int gtid) {
  return a.ws[((size_t)slot * a.frame_doubles + field) * (size_t)a.nthreads + gtid];
}

__global__ __launch_bounds__(PT_BLOCK) void pt_path_kernel(const PtKArgs a) {
  const int gtid = blockIdx.x * PT_BLOCK + threadIdx.x;
  const int S = a.S;
  const int nsamp = S > 0 ? S * S : 1;
  const int N = a.N;
  const double invN = 1.0 / (double)N;
  const V3 bg = {a.bg[0], a.bg[1], a.bg[2]};
  long long pix = gtid;
  bool alive = pix < a.npix;
  unsigned long long nrays = 0;

  Pcg pcg;
  pcg.state = 0;
  pcg.inc = 1;
  int samp = 0, sp = 0, col = 0, grow = 0;
  V3 cum = {0.0, 0.0, 0.0};
  Ray ray;
  ray.o = {0.0, 0.0, 0.0};
  ray.d = {1.0, 0.0, 0.0};
  ray.tmin = 1e-5;
  bool skip_query = false;  // max_depth < 0: the primary call returns black without a query

  // (re)start: pixel coordinates + seeds + the sample's primary ray
  auto start_sample = [&]() {
    if (samp == 0) {
      pixel_coords(a, pix, col, grow);
      if (a.pcg_mode == PT_PCG_PIXEL)
        pcg_seed(pcg, a.s0, a.q0 + ((unsigned long long)grow * a.W + col));
    }
    if (a.pcg_mode == PT_PCG_SAMPLE)
      pcg_seed(pcg, a.s0, a.q0 + ((unsigned long long)grow * a.W + col) * (unsigned)nsamp + (unsigned)samp);
    double up = 0.5, vp = 0.5;
    if (S > 0) {
      const int sr = samp / S, sc = samp - sr * S;
      up = ((double)sc + pcg_float(pcg)) / (double)S;
      vp = ((double)sr + pcg_float(pcg)) / (double)S;
    }
    ray = primary_ray(a, col, grow, up, vp);
    skip_query = a.D < 0;
  };
  if (alive) start_sample();

  while (alive) {
    V3 ret = {0.0, 0.0, 0.0};
    bool spawn = false;
    // registers describing the frame just pushed (child 0 is spawned from them)
    V3 f_wp = {0.0, 0.0, 0.0}, f_n = {0.0, 0.0, 1.0}, f_in = ray.d;
    int f_brdf = 0;

    if (!skip_query) {
      // ---- the convergent hot loop: one world query for this lane's current ray (depth = sp) ----
      double best_t = INFINITY;
      const int hit = world_query<false, false>(a, ray, best_t, true);
      nrays++;
      if (hit < 0) {
        ret = bg;  // render.py:103-105
      } else {
        const PtShapeAux *ax = a.aux + hit;
        Hit h;
        hit_details(a, ray, best_t, hit, h, ax->needs_uv != 0);
        V3 hc = brdf_pigment(a, ax, h.u, h.v);
        const V3 em = emitted_pigment(a, ax, h.u, h.v);
        const double lum = max2(max2(hc.x, hc.y), hc.z);
        bool go_on = true;
        if (sp >= a.rr) {  // render.py:116-123
          const double q = max2(0.05, 1.0 - lum);
          if (pcg_float(pcg) > q) {
            const double k = 1.0 / (1.0 - q);
            hc.x = hc.x * k;
            hc.y = hc.y * k;
            hc.z = hc.z * k;
          } else {
            ret = em;
            go_on = false;
          }
        }
        if (go_on) {
          if (lum > 0.0) {  // render.py:126-137: push the frame, spawn child 0
            ws_at(a, sp, 0, gtid) = hc.x;
            ws_at(a, sp, 1, gtid) = hc.y;
            ws_at(a, sp, 2, gtid) = hc.z;
            ws_at(a, sp, 3, gtid) = em.x;
            ws_at(a, sp, 4, gtid) = em.y;
            ws_at(a, sp, 5, gtid) = em.z;
            if (N > 1) {
              ws_at(a, sp, 6, gtid) = 0.0;
              ws_at(a, sp, 7, gtid) = 0.0;
              ws_at(a, sp, 8, gtid) = 0.0;
              ws_at(a, sp, 9, gtid) = 0.0;
              ws_at(a, sp, 10, gtid) = h.wp.x;
              ws_at(a, sp, 11, gtid) = h.wp.y;
              ws_at(a, sp, 12, gtid) = h.wp.z;
              ws_at(a, sp, 13, gtid) = h.n.x;
              ws_at(a, sp, 14, gtid) = h.n.y;
              ws_at(a, sp, 15, gtid) = h.n.z;
              ws_at(a, sp, 16, gtid) = ray.d.x;
              ws_at(a, sp, 17, gtid) = ray.d.y;
              ws_at(a, sp, 18, gtid) = ray.d.z;
              ws_at(a, sp, 19, gtid) = (double)ax->brdf_kind;
            }
            f_wp = h.wp;
            f_n = h.n;
            f_in = ray.d;
            f_brdf = ax->brdf_kind;
            sp++;
            spawn = true;
          } else {  // render.py:139 with cum_radiance = 0
            ret.x = em.x + 0.0 * invN;
            ret.y = em.y + 0.0 * invN;
            ret.z = em.z + 0.0 * invN;
          }
        }
      }
    }
    skip_query = false;

    // ---- unwind: deliver `ret` up the stack / spawn the next child, until a ray needs a query ----
    for (;;) {
      if (spawn) {
        // scatter_ray consumes its draws even when the child is beyond max_depth
        ray = scatter_ray(f_brdf, pcg, f_in, f_wp, f_n);
        spawn = false;
        if (sp > a.D) {  // render.py:100-101: the child returns black without a world query
          ret.x = 0.0;
          ret.y = 0.0;
          ret.z = 0.0;
          continue;
        }
        break;
      }
      if (sp == 0) {  // the primary call returned: one sample done (imagetracer.py:94-104)
        if (S > 0) {
          cum.x = cum.x + ret.x;
          cum.y = cum.y + ret.y;
          cum.z = cum.z + ret.z;
        } else {
          cum = ret;
        }
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
          pix += a.nthreads;
          if (pix >= a.npix) {
            alive = false;
            break;
          }
        }
        start_sample();
        if (skip_query) {
          skip_query = false;
          ret.x = 0.0;
          ret.y = 0.0;
          ret.z = 0.0;
          continue;
        }
        break;
      }
      // a child of frame sp-1 returned `ret` (render.py:135-137)
      const int fs = sp - 1;
      const V3 hc = {ws_at(a, fs, 0, gtid), ws_at(a, fs, 1, gtid), ws_at(a, fs, 2, gtid)};
      V3 fc = {0.0, 0.0, 0.0};
      int done = 0;
      if (N > 1) {
        fc.x = ws_at(a, fs, 6, gtid);
        fc.y = ws_at(a, fs, 7, gtid);
        fc.z = ws_at(a, fs, 8, gtid);
        done = (int)ws_at(a, fs, 9, gtid);
      }
      fc.x = fc.x + hc.x * ret.x;
      fc.y = fc.y + hc.y * ret.y;
      fc.z = fc.z + hc.z * ret.z;
      done++;
      if (done < N) {
        ws_at(a, fs, 6, gtid) = fc.x;
        ws_at(a, fs, 7, gtid) = fc.y;
        ws_at(a, fs, 8, gtid) = fc.z;
        ws_at(a, fs, 9, gtid) = (double)done;
        f_wp = {ws_at(a, fs, 10, gtid), ws_at(a, fs, 11, gtid), ws_at(a, fs, 12, gtid)};
        f_n = {ws_at(a, fs, 13, gtid), ws_at(a, fs, 14, gtid), ws_at(a, fs, 15, gtid)};
        f_in = {ws_at(a, fs, 16, gtid), ws_at(a, fs, 17, gtid), ws_at(a, fs, 18, gtid)};
        f_brdf = (int)ws_at(a, fs, 19, gtid);
        spawn = true;
        continue;
      }
      // render.py:139
      ret.x = ws_at(a, fs, 3, gtid) + fc.x * invN;
      ret.y = ws_at(a, fs, 4, gtid) + fc.y * invN;
      ret.z = ws_at(a, fs, 5, gtid) + fc.z * invN;
      sp = fs;
    }
  }
  add_ray_count(a, nrays);
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
    default: break;
  }
  out[i] = r;
}
