/* pt_oracle.c — CPU restatement of pytracer's per-pixel ray-trace/shade path.
 *
 * *** TEST INFRASTRUCTURE — NOT PRODUCT CODE ***
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and only as the checker / reported CPU baseline.  The product path (libptrace.so, HIP) never
 * links, imports or falls back to it.
 *
 * Parity status: PINNED.  the npz files under tests/golden/ hold outputs of the reference itself (imported in
 * the build container by tests/golden/make_golden.py); tests/test_oracle_golden.py checks this
 * file against every one of them bit-for-bit (sqr mode 0), plus the known-answer vectors of the
 * reference's own tests/test_all.py.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/pytracer/).  All arithmetic is IEEE fp64 in the reference's operation
 * order; build with -ffp-contract=off (Python never fuses a*b+c) and glibc libm (the reference
 * uses math.sqrt/sin/cos/atan2/acos/floor).
 *
 * sqr mode: Vec.squared_norm (geometry.py:114-118) is written `x**2`, which CPython evaluates
 * with libm pow(x, 2.0); glibc's pow is not correctly rounded and differs from x*x by 1 ulp for
 * ~0.08 % of inputs (SURVEY.md hazard H2).  Mode 0 (default) calls pow — bit-exact with the
 * reference; mode 1 uses x*x — the arithmetic the device kernel uses, so that the device can be
 * compared bit-for-bit with this file wherever no libm transcendental is involved.
 */
#include "../include/ptrace.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_PI 3.141592653589793 /* math.pi */

static int g_sqr_mode = 0;
/* volatile pointer: keeps gcc from folding pow(x, 2.0) into x*x */
static double (*volatile g_pow)(double, double) = pow;

void pto_set_sqr_mode(int mode) { g_sqr_mode = mode; }
int pto_get_sqr_mode(void) { return g_sqr_mode; }

static inline double sq(double x) { return g_sqr_mode ? x * x : g_pow(x, 2.0); }

/* ---- PCG (pcg.py:23-62) --------------------------------------------------------------------*/
typedef struct {
  uint64_t state, inc;
} pcg_t;

/* pcg.py:43-58 */
static inline uint32_t pcg_random(pcg_t *p) {
  uint64_t old = p->state;
  p->state = old * 6364136223846793005ULL + p->inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
  uint32_t rot = (uint32_t)(old >> 59);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}

/* pcg.py:29-41 */
static inline void pcg_init(pcg_t *p, uint64_t init_state, uint64_t init_seq) {
  p->state = 0;
  p->inc = (init_seq << 1) | 1;
  pcg_random(p);
  p->state += init_state;
  pcg_random(p);
}

/* pcg.py:60-62: int / int true division == fp64 division of the two exactly-representable ints */
static inline double pcg_random_float(pcg_t *p) { return (double)pcg_random(p) / 4294967295.0; }

void pto_pcg_init(uint64_t st[2], uint64_t init_state, uint64_t init_seq) {
  pcg_t p;
  pcg_init(&p, init_state, init_seq);
  st[0] = p.state;
  st[1] = p.inc;
}
uint32_t pto_pcg_random(uint64_t st[2]) {
  pcg_t p = {st[0], st[1]};
  uint32_t r = pcg_random(&p);
  st[0] = p.state;
  return r;
}
double pto_pcg_random_float(uint64_t st[2]) {
  pcg_t p = {st[0], st[1]};
  double r = pcg_random_float(&p);
  st[0] = p.state;
  return r;
}

/* ---- geometry / transformations ----------------------------------------------------------- */
typedef struct {
  double x, y, z;
} v3;

typedef struct {
  v3 o, d;
  double tmin, tmax;
  int depth;
} ray_t; /* ray.py:29-44 */

/* transformations.py:67-78 (Point; the w row of an affine matrix gives exactly 1.0) */
static inline v3 xf_point(const double m[12], v3 p) {
  v3 r;
  r.x = p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3];
  r.y = p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7];
  r.z = p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11];
  return r;
}
/* transformations.py:59-66 (Vec) */
static inline v3 xf_vec(const double m[12], v3 v) {
  v3 r;
  r.x = v.x * m[0] + v.y * m[1] + v.z * m[2];
  r.y = v.x * m[4] + v.y * m[5] + v.z * m[6];
  r.z = v.x * m[8] + v.y * m[9] + v.z * m[10];
  return r;
}
/* transformations.py:79-86 (Normal: transpose of the inverse) */
static inline v3 xf_normal(const double invm[12], v3 n) {
  v3 r;
  r.x = n.x * invm[0] + n.y * invm[4] + n.z * invm[8];
  r.y = n.x * invm[1] + n.y * invm[5] + n.z * invm[9];
  r.z = n.x * invm[2] + n.y * invm[6] + n.z * invm[10];
  return r;
}
/* geometry.py:110-112 */
static inline double dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* geometry.py:114-118 (Vec.squared_norm: ** 2) */
static inline double vec_sqnorm(v3 a) { return sq(a.x) + sq(a.y) + sq(a.z); }
/* geometry.py:130-136 (Vec.normalize) */
static inline v3 vec_normalize(v3 a) {
  double n = sqrt(vec_sqnorm(a));
  v3 r = {a.x / n, a.y / n, a.z / n};
  return r;
}
/* geometry.py:213-225 (Normal.squared_norm uses x*x, normalize divides) */
static inline v3 normal_normalize(v3 a) {
  double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  v3 r = {a.x / n, a.y / n, a.z / n};
  return r;
}
/* ray.py:52-57 */
static inline v3 ray_at(const ray_t *r, double t) {
  v3 p = {r->o.x + t * r->d.x, r->o.y + t * r->d.y, r->o.z + t * r->d.z};
  return p;
}
/* ray.py:59-69 */
static inline ray_t ray_transform(const ray_t *r, const double m[12]) {
  ray_t q;
  q.o = xf_point(m, r->o);
  q.d = xf_vec(m, r->d);
  q.tmin = r->tmin;
  q.tmax = r->tmax;
  q.depth = r->depth;
  return q;
}
/* geometry.py:265-276 */
static inline double normalized_dot(v3 a, v3 b) { return dot3(vec_normalize(a), vec_normalize(b)); }

/* ---- scene access --------------------------------------------------------------------------*/
static inline void load12(const double *a, int n, int i, double out[12]) {
  for (int k = 0; k < 12; ++k) out[k] = a[(size_t)k * n + i];
}
static inline void load3(const double *a, int n, int i, double out[3]) {
  for (int k = 0; k < 3; ++k) out[k] = a[(size_t)k * n + i];
}

typedef struct {
  double t;
  v3 world_point;
  v3 normal;
  double u, v;
  int shape;
} hit_t; /* hitrecord.py:27-46 (ray and material are implied by the caller / shape index) */

static int g_acos_domain_errors = 0; /* H4: the reference raises ValueError here; we clamp */
int pto_acos_domain_errors(void) { return g_acos_domain_errors; }

/* shapes.py:36-42 */
static inline void sphere_uv(v3 p, double *u, double *v) {
  double uu = atan2(p.y, p.x) / (2.0 * PTO_PI);
  *u = (uu >= 0.0) ? uu : uu + 1.0;
  double z = p.z;
  if (z > 1.0) {
    z = 1.0;
    g_acos_domain_errors++;
  } else if (z < -1.0) {
    z = -1.0;
    g_acos_domain_errors++;
  }
  *v = acos(z) / PTO_PI;
}

/* shapes.py:97-131 */
static int sphere_intersect(const pt_scene_desc *s, int i, const ray_t *ray, hit_t *h) {
  double invm[12], m[12];
  load12(s->invm, s->n_shapes, i, invm);
  ray_t inv = ray_transform(ray, invm);
  double a = vec_sqnorm(inv.d);
  double b = 2.0 * dot3(inv.o, inv.d);
  double c = vec_sqnorm(inv.o) - 1.0;
  double delta = b * b - 4.0 * a * c;
  if (delta <= 0.0) return 0;
  double sd = sqrt(delta);
  double t1 = (-b - sd) / (2.0 * a);
  double t2 = (-b + sd) / (2.0 * a);
  double t;
  if (t1 > inv.tmin && t1 < inv.tmax)
    t = t1;
  else if (t2 > inv.tmin && t2 < inv.tmax)
    t = t2;
  else
    return 0;
  v3 hp = ray_at(&inv, t);
  load12(s->m, s->n_shapes, i, m);
  h->world_point = xf_point(m, hp);
  /* shapes.py:45-54 */
  v3 n = hp;
  if (!(dot3(hp, inv.d) < 0.0)) {
    n.x = -hp.x;
    n.y = -hp.y;
    n.z = -hp.z;
  }
  h->normal = xf_normal(invm, n);
  sphere_uv(hp, &h->u, &h->v);
  h->t = t;
  h->shape = i;
  return 1;
}

/* shapes.py:163-189 */
static int plane_intersect(const pt_scene_desc *s, int i, const ray_t *ray, hit_t *h) {
  double invm[12], m[12];
  load12(s->invm, s->n_shapes, i, invm);
  ray_t inv = ray_transform(ray, invm);
  if (fabs(inv.d.z) < 1e-5) return 0;
  double t = -inv.o.z / inv.d.z;
  if (t <= inv.tmin || t >= inv.tmax) return 0;
  v3 hp = ray_at(&inv, t);
  load12(s->m, s->n_shapes, i, m);
  h->world_point = xf_point(m, hp);
  v3 n = {0.0, 0.0, (inv.d.z < 0.0) ? 1.0 : -1.0};
  h->normal = xf_normal(invm, n);
  h->u = hp.x - floor(hp.x);
  h->v = hp.y - floor(hp.y);
  h->t = t;
  h->shape = i;
  return 1;
}

static inline int shape_intersect(const pt_scene_desc *s, int i, const ray_t *ray, hit_t *h) {
  return s->kind[i] == PT_SHAPE_SPHERE ? sphere_intersect(s, i, ray, h)
                                       : plane_intersect(s, i, ray, h);
}

/* shapes.py:133-151 / 191-198 */
static int shape_quick_intersect(const pt_scene_desc *s, int i, const ray_t *ray) {
  double invm[12];
  load12(s->invm, s->n_shapes, i, invm);
  ray_t inv = ray_transform(ray, invm);
  if (s->kind[i] == PT_SHAPE_SPHERE) {
    double a = vec_sqnorm(inv.d);
    double b = 2.0 * dot3(inv.o, inv.d);
    double c = vec_sqnorm(inv.o) - 1.0;
    double delta = b * b - 4.0 * a * c;
    if (delta <= 0.0) return 0;
    double sd = sqrt(delta);
    double t1 = (-b - sd) / (2.0 * a);
    double t2 = (-b + sd) / (2.0 * a);
    return (inv.tmin < t1 && t1 < inv.tmax) || (inv.tmin < t2 && t2 < inv.tmax);
  }
  if (fabs(inv.d.z) < 1e-5) return 0;
  double t = -inv.o.z / inv.d.z;
  return inv.tmin < t && t < inv.tmax;
}

/* world.py:51-69; *n_rays counts world queries */
static int world_intersect(const pt_scene_desc *s, const ray_t *ray, hit_t *closest) {
  int found = 0;
  hit_t h;
  for (int i = 0; i < s->n_shapes; ++i) {
    if (!shape_intersect(s, i, ray, &h)) continue;
    if (!found || h.t < closest->t) {
      *closest = h;
      found = 1;
    }
  }
  if (found) closest->normal = normal_normalize(closest->normal);
  return found;
}

/* world.py:71-80 */
static int is_point_visible(const pt_scene_desc *s, v3 point, v3 observer) {
  v3 dir = {point.x - observer.x, point.y - observer.y, point.z - observer.z};
  double dn = sqrt(vec_sqnorm(dir));
  ray_t r;
  r.o = observer;
  r.d = dir;
  r.tmin = 1e-2 / dn;
  r.tmax = 1.0;
  r.depth = 0;
  for (int i = 0; i < s->n_shapes; ++i)
    if (shape_quick_intersect(s, i, &r)) return 0;
  return 1;
}

/* ---- pigments (materials.py:50-100) --------------------------------------------------------*/
static inline int64_t py_mod2(int64_t a) { return ((a % 2) + 2) % 2; } /* Python's a % 2 */

static v3 pigment_color(const pt_scene_desc *s, int i, int emitted, double u, double v) {
  const int n = s->n_shapes;
  int kind = emitted ? s->emi_kind[i] : s->pig_kind[i];
  const double *c1 = emitted ? s->emi_c1 : s->pig_c1;
  const double *c2 = emitted ? s->emi_c2 : s->pig_c2;
  double col[3];
  if (kind == PT_PIGMENT_UNIFORM) { /* materials.py:58-59 */
    load3(c1, n, i, col);
  } else if (kind == PT_PIGMENT_CHECKERED) { /* materials.py:96-100 */
    double steps = emitted ? s->emi_steps[i] : s->pig_steps[i];
    int64_t iu = (int64_t)floor(u * steps);
    int64_t iv = (int64_t)floor(v * steps);
    load3((py_mod2(iu) == py_mod2(iv)) ? c1 : c2, n, i, col);
  } else { /* materials.py:69-82 */
    int t = emitted ? s->emi_tex[i] : s->pig_tex[i];
    int w = s->tex_w[t], hgt = s->tex_h[t];
    int64_t c = (int64_t)(u * w); /* int() truncates toward zero */
    int64_t r = (int64_t)(v * hgt);
    if (c >= w) c = w - 1;
    if (r >= hgt) r = hgt - 1;
    const double *px = s->tex_data + s->tex_offset[t] + (r * w + c) * 3;
    col[0] = px[0];
    col[1] = px[1];
    col[2] = px[2];
  }
  v3 r = {col[0], col[1], col[2]};
  return r;
}

/* ---- ONB and BRDF scattering ----------------------------------------------------------------*/
/* geometry.py:247-262 */
static inline void onb_from_z(v3 n, v3 *e1, v3 *e2, v3 *e3) {
  double sign = (n.z > 0.0) ? 1.0 : -1.0;
  double a = -1.0 / (sign + n.z);
  double b = n.x * n.y * a;
  e1->x = 1.0 + sign * n.x * n.x * a;
  e1->y = sign * b;
  e1->z = -sign * n.x;
  e2->x = b;
  e2->y = sign + n.y * n.y * a;
  e2->z = -n.y;
  *e3 = n;
}

/* materials.py:132-152 (diffuse), :175-196 (specular) */
static ray_t scatter_ray(int brdf_kind, pcg_t *pcg, v3 incoming, v3 point, v3 normal, int depth) {
  ray_t r;
  r.o = point;
  r.tmax = INFINITY;
  r.depth = depth;
  if (brdf_kind == PT_BRDF_DIFFUSE) {
    v3 e1, e2, e3;
    onb_from_z(normal, &e1, &e2, &e3);
    double cts = pcg_random_float(pcg);
    double ct = sqrt(cts), st = sqrt(1.0 - cts);
    double phi = 2.0 * PTO_PI * pcg_random_float(pcg);
    double cp = cos(phi), sp = sin(phi);
    r.d.x = ct * (cp * e1.x) + ct * (sp * e2.x) + st * e3.x;
    r.d.y = ct * (cp * e1.y) + ct * (sp * e2.y) + st * e3.y;
    r.d.z = ct * (cp * e1.z) + ct * (sp * e2.z) + st * e3.z;
    r.tmin = 1.0e-3;
  } else {
    v3 rd = vec_normalize(incoming);
    v3 nn = vec_normalize(normal);
    double dp = dot3(nn, rd);
    r.d.x = rd.x - dp * (2.0 * nn.x);
    r.d.y = rd.y - dp * (2.0 * nn.y);
    r.d.z = rd.z - dp * (2.0 * nn.z);
    r.tmin = 1e-5;
  }
  return r;
}

/* materials.py:129-130 (diffuse eval), :164-173 (specular eval) */
static v3 brdf_eval(const pt_scene_desc *s, int i, v3 normal, v3 in_dir, v3 out_dir, double u,
                    double v) {
  if (s->brdf_kind[i] == PT_BRDF_DIFFUSE) {
    v3 c = pigment_color(s, i, 0, u, v);
    double k = 1.0 / PTO_PI;
    v3 r = {c.x * k, c.y * k, c.z * k};
    return r;
  }
  double theta_in = acos(normalized_dot(normal, in_dir));
  double theta_out = acos(normalized_dot(normal, out_dir));
  if (fabs(theta_in - theta_out) < s->brdf_param[i]) return pigment_color(s, i, 0, u, v);
  v3 z = {0.0, 0.0, 0.0};
  return z;
}

/* ---- renderers (render.py) -------------------------------------------------------------------*/
typedef struct {
  const pt_scene_desc *s;
  const pt_params *p;
  pcg_t *path_pcg;
  uint64_t n_rays;
} rctx_t;

static inline double max2(double a, double b) { return (b > a) ? b : a; } /* Python max(a, b) */

/* render.py:99-139 */
static v3 pathtracer(rctx_t *c, const ray_t *ray) {
  const pt_params *p = c->p;
  v3 zero = {0.0, 0.0, 0.0};
  if (ray->depth > p->max_depth) return zero;
  hit_t h;
  c->n_rays++;
  if (!world_intersect(c->s, ray, &h)) {
    v3 bg = {p->background[0], p->background[1], p->background[2]};
    return bg;
  }
  v3 hc = pigment_color(c->s, h.shape, 0, h.u, h.v);
  v3 em = pigment_color(c->s, h.shape, 1, h.u, h.v);
  double lum = max2(max2(hc.x, hc.y), hc.z);
  if (ray->depth >= p->rr_limit) {
    double q = max2(0.05, 1 - lum);
    if (pcg_random_float(c->path_pcg) > q) {
      double k = 1.0 / (1.0 - q);
      hc.x = hc.x * k;
      hc.y = hc.y * k;
      hc.z = hc.z * k;
    } else {
      return em;
    }
  }
  v3 cum = zero;
  if (lum > 0.0) {
    for (int i = 0; i < p->num_of_rays; ++i) {
      ray_t nr = scatter_ray(c->s->brdf_kind[h.shape], c->path_pcg, ray->d, h.world_point, h.normal,
                             ray->depth + 1);
      v3 rad = pathtracer(c, &nr);
      cum.x = cum.x + hc.x * rad.x;
      cum.y = cum.y + hc.y * rad.y;
      cum.z = cum.z + hc.z * rad.z;
    }
  }
  double k = 1.0 / p->num_of_rays;
  v3 r = {em.x + cum.x * k, em.y + cum.y * k, em.z + cum.z * k};
  return r;
}

/* render.py:157-193 */
static v3 pointlight(rctx_t *c, const ray_t *ray) {
  const pt_params *p = c->p;
  const pt_scene_desc *s = c->s;
  hit_t h;
  c->n_rays++;
  if (!world_intersect(s, ray, &h)) {
    v3 bg = {p->background[0], p->background[1], p->background[2]};
    return bg;
  }
  v3 em = pigment_color(s, h.shape, 1, h.u, h.v);
  v3 res = {p->ambient[0] + em.x, p->ambient[1] + em.y, p->ambient[2] + em.z};
  for (int l = 0; l < s->n_lights; ++l) {
    double lp[3], lc[3];
    load3(s->light_pos, s->n_lights, l, lp);
    load3(s->light_color, s->n_lights, l, lc);
    v3 pos = {lp[0], lp[1], lp[2]};
    c->n_rays++; /* shadow ray */
    if (!is_point_visible(s, pos, h.world_point)) continue;
    v3 dv = {h.world_point.x - pos.x, h.world_point.y - pos.y, h.world_point.z - pos.z};
    double dist = sqrt(vec_sqnorm(dv));
    double inv = 1.0 / dist;
    v3 in_dir = {inv * dv.x, inv * dv.y, inv * dv.z};
    v3 neg_in = {-in_dir.x, -in_dir.y, -in_dir.z};
    double cos_theta = max2(0.0, normalized_dot(neg_in, h.normal));
    double lr = s->light_radius[l];
    double df = (lr > 0) ? sq(lr / dist) : 1.0; /* render.py:176-180: (...) ** 2 */
    v3 out_dir = {-ray->d.x, -ray->d.y, -ray->d.z};
    v3 bc = brdf_eval(s, h.shape, h.normal, in_dir, out_dir, h.u, h.v);
    res.x = res.x + bc.x * lc[0] * cos_theta * df;
    res.y = res.y + bc.y * lc[1] * cos_theta * df;
    res.z = res.z + bc.z * lc[2] * cos_theta * df;
  }
  return res;
}

static v3 radiance(rctx_t *c, const ray_t *ray) {
  const pt_params *p = c->p;
  switch (p->renderer) {
    case PT_RENDERER_ONOFF: { /* render.py:52-53 */
      hit_t h;
      c->n_rays++;
      int hit = world_intersect(c->s, ray, &h);
      v3 r = {hit ? p->onoff_color[0] : p->background[0], hit ? p->onoff_color[1] : p->background[1],
              hit ? p->onoff_color[2] : p->background[2]};
      return r;
    }
    case PT_RENDERER_FLAT: { /* render.py:65-74 */
      hit_t h;
      c->n_rays++;
      if (!world_intersect(c->s, ray, &h)) {
        v3 bg = {p->background[0], p->background[1], p->background[2]};
        return bg;
      }
      v3 a = pigment_color(c->s, h.shape, 0, h.u, h.v);
      v3 b = pigment_color(c->s, h.shape, 1, h.u, h.v);
      v3 r = {a.x + b.x, a.y + b.y, a.z + b.z};
      return r;
    }
    case PT_RENDERER_PATHTRACER:
      return pathtracer(c, ray);
    default:
      return pointlight(c, ray);
  }
}

/* ---- camera + per-pixel driver ---------------------------------------------------------------*/
/* camera.py:59-78, 103-124 */
static ray_t camera_fire_ray(const pt_camera *cam, double u, double v) {
  ray_t r;
  if (cam->kind == PT_CAMERA_PERSPECTIVE) {
    r.o.x = -cam->screen_distance;
    r.o.y = 0.0;
    r.o.z = 0.0;
    r.d.x = cam->screen_distance;
    r.d.y = (1.0 - 2 * u) * cam->aspect_ratio;
    r.d.z = 2 * v - 1;
  } else {
    r.o.x = -1.0;
    r.o.y = (1.0 - 2 * u) * cam->aspect_ratio;
    r.o.z = 2 * v - 1;
    r.d.x = 1.0;
    r.d.y = 0.0;
    r.d.z = 0.0;
  }
  r.tmin = 1.0e-5;
  r.tmax = INFINITY;
  r.depth = 0;
  return ray_transform(&r, cam->m);
}

/* imagetracer.py:48-58 */
static ray_t tracer_fire_ray(const pt_camera *cam, int W, int H, int col, int row, double up,
                             double vp) {
  double u = (col + up) / W;
  double v = 1.0 - (row + vp) / H;
  return camera_fire_ray(cam, u, v);
}

int pt_rows_for_rank_oracle(const pt_params *p) {
  int rb = p->row_block > 0 ? p->row_block : 1;
  int nr = p->n_ranks > 0 ? p->n_ranks : 1;
  int rows = 0;
  for (int b = 0; b * rb < p->height; ++b)
    if (b % nr == p->rank) {
      int r0 = b * rb, r1 = r0 + rb;
      if (r1 > p->height) r1 = p->height;
      rows += r1 - r0;
    }
  return rows;
}

static inline void store_px(void *out, int fmt, size_t idx, v3 c) {
  if (fmt == PT_OUT_F64) {
    double *o = (double *)out + idx * 3;
    o[0] = c.x;
    o[1] = c.y;
    o[2] = c.z;
  } else {
    float *o = (float *)out + idx * 3;
    o[0] = (float)c.x;
    o[1] = (float)c.y;
    o[2] = (float)c.z;
  }
}

/* imagetracer.py:80-104 for one pixel; jitter draws from jp, scattering from ctx->path_pcg */
static v3 trace_pixel(rctx_t *ctx, const pt_camera *cam, int col, int row, pcg_t *jp,
                      uint64_t pixel_index) {
  const pt_params *p = ctx->p;
  const int S = p->samples_per_side;
  v3 cum = {0.0, 0.0, 0.0};
  if (S > 0) {
    for (int sr = 0; sr < S; ++sr)
      for (int sc = 0; sc < S; ++sc) {
        if (p->pcg_mode == PT_PCG_SAMPLE) {
          pcg_init(jp, p->path_state, p->path_seq + pixel_index * (uint64_t)(S * S) + (uint64_t)(sr * S + sc));
        }
        double up = (sc + pcg_random_float(jp)) / S;
        double vp = (sr + pcg_random_float(jp)) / S;
        ray_t ray = tracer_fire_ray(cam, p->width, p->height, col, row, up, vp);
        v3 c = radiance(ctx, &ray);
        cum.x = cum.x + c.x;
        cum.y = cum.y + c.y;
        cum.z = cum.z + c.z;
      }
    double k = 1.0 / (double)(S * S); /* imagetracer.py:100: 1 / S**2 (int / int) */
    v3 r = {cum.x * k, cum.y * k, cum.z * k};
    return r;
  }
  ray_t ray = tracer_fire_ray(cam, p->width, p->height, col, row, 0.5, 0.5);
  return radiance(ctx, &ray);
}

/* diagnostics (tools/ray_histogram.py): when set, pto_render also writes every pixel's ray count (PT_PCG_PIXEL / SAMPLE
 * frames, local row-major like `out`) */
static uint32_t *g_ray_image = 0;
void pto_set_ray_image(uint32_t *buf) { g_ray_image = buf; }

/* imagetracer.py:60-110.  out holds this rank's rows compactly. */
int pto_render(const pt_scene_desc *s, const pt_camera *cam, const pt_params *p, void *out,
               size_t out_bytes, int n_threads, uint64_t *n_rays_out) {
  if (!s || !cam || !p || !out) return PT_ERR_INVALID;
  if (p->width <= 0 || p->height <= 0 || p->samples_per_side < 0) return PT_ERR_INVALID;
  const int W = p->width, H = p->height;
  const int rb = p->row_block > 0 ? p->row_block : 1;
  const int nr = p->n_ranks > 0 ? p->n_ranks : 1;
  if (p->rank < 0 || p->rank >= nr) return PT_ERR_INVALID;
  const int rows = pt_rows_for_rank_oracle(p);
  const size_t esz = p->out_format == PT_OUT_F64 ? 8 : 4;
  if (out_bytes < (size_t)rows * W * 3 * esz) return PT_ERR_SIZE;
  /* local row -> global row table */
  int *grow = (int *)malloc(sizeof(int) * (rows > 0 ? rows : 1));
  if (!grow) return PT_ERR_NOMEM;
  int k = 0;
  for (int r = 0; r < H; ++r)
    if ((r / rb) % nr == p->rank) grow[k++] = r;
  uint64_t total_rays = 0;

  if (p->pcg_mode == PT_PCG_SEQ) {
    /* the reference's own semantics: two global streams, strictly serial */
    pcg_t jitter, path;
    pcg_init(&jitter, p->jitter_state, p->jitter_seq);
    pcg_init(&path, p->path_state, p->path_seq);
    rctx_t ctx = {s, p, &path, 0};
    for (int lr = 0; lr < rows; ++lr)
      for (int col = 0; col < W; ++col) {
        int row = grow[lr];
        v3 c = trace_pixel(&ctx, cam, col, row, &jitter, (uint64_t)row * W + col);
        store_px(out, p->out_format, (size_t)lr * W + col, c);
      }
    total_rays = ctx.n_rays;
  } else {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total_rays)
#endif
    for (int lr = 0; lr < rows; ++lr) {
      int row = grow[lr];
      for (int col = 0; col < W; ++col) {
        uint64_t pix = (uint64_t)row * W + col;
        pcg_t pcg;
        pcg_init(&pcg, p->path_state, p->path_seq + pix);
        rctx_t ctx = {s, p, &pcg, 0};
        v3 c = trace_pixel(&ctx, cam, col, row, &pcg, pix);
        store_px(out, p->out_format, (size_t)lr * W + col, c);
        if (g_ray_image) g_ray_image[(size_t)lr * W + col] = (uint32_t)ctx.n_rays;
        total_rays += ctx.n_rays;
      }
    }
  }
  free(grow);
  if (n_rays_out) *n_rays_out = total_rays;
  return PT_OK;
}

int pto_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- unit-level entry points used by the golden tests ------------------------------------------*/
static ray_t ray_from8(const double r[8], int depth) {
  ray_t q;
  q.o.x = r[0];
  q.o.y = r[1];
  q.o.z = r[2];
  q.d.x = r[3];
  q.d.y = r[4];
  q.d.z = r[5];
  q.tmin = r[6];
  q.tmax = r[7];
  q.depth = depth;
  return q;
}
static void ray_to8(const ray_t *q, double r[8]) {
  r[0] = q->o.x;
  r[1] = q->o.y;
  r[2] = q->o.z;
  r[3] = q->d.x;
  r[4] = q->d.y;
  r[5] = q->d.z;
  r[6] = q->tmin;
  r[7] = q->tmax;
}
static void hit_to10(const hit_t *h, double o[10]) {
  o[0] = h->t;
  o[1] = h->world_point.x;
  o[2] = h->world_point.y;
  o[3] = h->world_point.z;
  o[4] = h->normal.x;
  o[5] = h->normal.y;
  o[6] = h->normal.z;
  o[7] = h->u;
  o[8] = h->v;
  o[9] = (double)h->shape;
}

/* what: 0 = m*Point, 1 = m*Vec, 2 = Normal (pass invm) */
void pto_xform(const double m[12], int what, const double in[3], double out[3]) {
  v3 a = {in[0], in[1], in[2]}, r;
  r = what == 0 ? xf_point(m, a) : what == 1 ? xf_vec(m, a) : xf_normal(m, a);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
}
/* Shape.ray_intersection: normal NOT normalised (as the per-shape HitRecord) */
int pto_shape_intersect(const pt_scene_desc *s, int i, const double ray[8], double out[10]) {
  ray_t r = ray_from8(ray, 0);
  hit_t h;
  memset(&h, 0, sizeof h);
  int hit = shape_intersect(s, i, &r, &h);
  if (hit) hit_to10(&h, out);
  return hit;
}
int pto_shape_quick_intersect(const pt_scene_desc *s, int i, const double ray[8]) {
  ray_t r = ray_from8(ray, 0);
  return shape_quick_intersect(s, i, &r);
}
/* World.ray_intersection: normal normalised */
int pto_world_intersect(const pt_scene_desc *s, const double ray[8], double out[10]) {
  ray_t r = ray_from8(ray, 0);
  hit_t h;
  memset(&h, 0, sizeof h);
  int hit = world_intersect(s, &r, &h);
  if (hit) hit_to10(&h, out);
  return hit;
}
int pto_is_point_visible(const pt_scene_desc *s, const double point[3], const double obs[3]) {
  v3 a = {point[0], point[1], point[2]}, b = {obs[0], obs[1], obs[2]};
  return is_point_visible(s, a, b);
}
void pto_camera_fire_ray(const pt_camera *cam, double u, double v, double out[8]) {
  ray_t r = camera_fire_ray(cam, u, v);
  ray_to8(&r, out);
}
void pto_tracer_fire_ray(const pt_camera *cam, int W, int H, int col, int row, double up, double vp,
                         double out[8]) {
  ray_t r = tracer_fire_ray(cam, W, H, col, row, up, vp);
  ray_to8(&r, out);
}
void pto_onb(const double n[3], double out[9]) {
  v3 a = {n[0], n[1], n[2]}, e1, e2, e3;
  onb_from_z(a, &e1, &e2, &e3);
  out[0] = e1.x;
  out[1] = e1.y;
  out[2] = e1.z;
  out[3] = e2.x;
  out[4] = e2.y;
  out[5] = e2.z;
  out[6] = e3.x;
  out[7] = e3.y;
  out[8] = e3.z;
}
void pto_scatter(int brdf_kind, uint64_t pcg_state[2], const double in_dir[3], const double point[3],
                 const double normal[3], int depth, double out[8]) {
  pcg_t p = {pcg_state[0], pcg_state[1]};
  v3 a = {in_dir[0], in_dir[1], in_dir[2]}, b = {point[0], point[1], point[2]},
     c = {normal[0], normal[1], normal[2]};
  ray_t r = scatter_ray(brdf_kind, &p, a, b, c, depth);
  pcg_state[0] = p.state;
  ray_to8(&r, out);
}
void pto_pigment(const pt_scene_desc *s, int i, int emitted, double u, double v, double out[3]) {
  v3 c = pigment_color(s, i, emitted, u, v);
  out[0] = c.x;
  out[1] = c.y;
  out[2] = c.z;
}
/* Renderer.__call__(ray) for p->renderer; scattering draws come from pcg_state */
void pto_radiance(const pt_scene_desc *s, const pt_params *p, uint64_t pcg_state[2],
                  const double ray[8], int depth, double out[3], uint64_t *n_rays) {
  pcg_t pc = {pcg_state[0], pcg_state[1]};
  rctx_t ctx = {s, p, &pc, 0};
  ray_t r = ray_from8(ray, depth);
  v3 c = radiance(&ctx, &r);
  pcg_state[0] = pc.state;
  out[0] = c.x;
  out[1] = c.y;
  out[2] = c.z;
  if (n_rays) *n_rays = ctx.n_rays;
}

/* ---- HdrImage post-processing ----------------------------------------------*/
/* hdrimages.py:113-118: float32 payload, bottom row first; struct.pack("<f" / ">f") */
void pto_pack_pfm(const double *img, int W, int H, int big_endian, unsigned char *out) {
  size_t k = 0;
  for (int y = H - 1; y >= 0; --y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        float f = (float)img[((size_t)y * W + x) * 3 + c];
        unsigned char b[4];
        memcpy(b, &f, 4); /* host is little endian */
        if (big_endian) {
          out[k++] = b[3];
          out[k++] = b[2];
          out[k++] = b[1];
          out[k++] = b[0];
        } else {
          out[k++] = b[0];
          out[k++] = b[1];
          out[k++] = b[2];
          out[k++] = b[3];
        }
      }
}

/* hdrimages.py:120-128 with colors.py:59-63; sequential sum exactly as the reference loops */
double pto_average_luminosity(const double *img, long long npix, double delta) {
  double cumsum = 0.0;
  for (long long i = 0; i < npix; ++i) {
    const double r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
    double mx = r, mn = r; /* Python max/min keep the first of equal values; the value is the same */
    if (g > mx) mx = g;
    if (b > mx) mx = b;
    if (g < mn) mn = g;
    if (b < mn) mn = b;
    cumsum += log10(delta + (mx + mn) / 2);
  }
  return pow(10, cumsum / (double)npix);
}

/* hdrimages.py:130-146 (normalize with scale = factor / luminosity, clamp) and :160-166 (LDR bytes) */
void pto_tonemap(double *img, long long n, double scale, int clamp, double gamma, unsigned char *rgb8,
                 int write_back) {
  for (long long i = 0; i < n; ++i) {
    double x = img[i] * scale;
    if (clamp) x = x / (1 + x);
    if (write_back) img[i] = x;
    if (rgb8) {
      int b = (int)(255 * pow(x, 1 / gamma));
      rgb8[i] = (unsigned char)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
  }
}
