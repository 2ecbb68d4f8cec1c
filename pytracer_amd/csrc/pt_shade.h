// pt_shade.h -- HitRecord, pigments, BRDF.scatter_ray.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
      // (atan2 / acos ALWAYS behind a call, also in the inlined second-pass kernels: a sphere's (u, v) is needed only under a
      //  checkered or image pigment, and inlined their ~35 polynomial constants were hoisted to the kernels' entry and spilled --
      //  152 - 328 B of scratch re-read one dependent load per coefficient, tools/scratch_sites.py; the values are ocml's either way)
      const double uu = pt_atan2(hp.y, hp.x) / (2.0 * PT_PI);
      h.u = (uu >= 0.0) ? uu : uu + 1.0;
      double z = hp.z;  // the reference raises ValueError outside [-1, 1]: clamp
      z = (z > 1.0) ? 1.0 : ((z < -1.0) ? -1.0 : z);
      h.v = pt_acos(z) / PT_PI;
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
    // materials.py:96-100: int(floor(u * n)) % 2 == int(floor(v * n)) % 2.  The parity of an integer-valued double t without
    // a conversion to a 64-bit integer (a dozen instructions on this hardware, and undefined beyond 2^63): t is odd iff
    // halving it leaves a fraction -- t * 0.5 is exact, so is floor, so is the doubling; beyond 2^53 every double is even,
    // which is what Python's exact int() finds too.  Python's % 2 is non-negative, so -3 is odd like 3.
    const double tu = floor(u * steps), tv = floor(v * steps);
    const bool odd_u = floor(tu * 0.5) * 2.0 != tu, odd_v = floor(tv * 0.5) * 2.0 != tv;
    c = (odd_u == odd_v) ? c1 : c2;
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
    double cp, sp;
    if (INL)
      sincos(phi, &sp, &cp);
    else
      pt_sincos(phi, &sp, &cp);
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
