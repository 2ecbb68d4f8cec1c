// pt_simple.h -- PointLightRenderer shading; the one-lane-one-pixel kernel.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
  // the BRDF's pigment at the hit point (materials.py:129, 172: `self.pigment.get_color(uv)`) does not depend on the light: the
  // reference evaluates it once per visible light, here once per hit -- the same value, and (u, v), the shape's record and
  // the checker's arithmetic no longer live across the shadow rays' queries
  V3 pc = {0.0, 0.0, 0.0};
  bool diffuse = true;
  double spec_threshold = 0.0;
  if (lit) {
    pc = brdf_pigment(a, ax, h.u, h.v);
    diffuse = ax->brdf_kind == PT_BRDF_DIFFUSE;
    spec_threshold = ax->brdf_param;
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
      if (diffuse) {  // materials.py:129-130
        const double k = 1.0 / PT_PI;
        bc.x = pc.x * k;
        bc.y = pc.y * k;
        bc.z = pc.z * k;
      } else {  // materials.py:164-173
        const V3 out_dir = {-ray.d.x, -ray.d.y, -ray.d.z};
        const double th_in = pt_acos(dot3(normalize3(h.n), normalize3(in_dir)));
        const double th_out = pt_acos(dot3(normalize3(h.n), normalize3(out_dir)));
        if (fabs(th_in - th_out) < spec_threshold) bc = pc;
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
      pcg_seed_pixel(pcg, c->pcg_mode, c->s0, c->q0, gpix, nsamp);
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
