// pt_probes.h -- diagnostic kernels behind include/ptrace_debug.h.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
    case 11: {  // sincos(phi) against sin(phi), cos(phi) for phi = 2 pi pcg_unit(v), v in [x[i], x[i] + y[i]): the mismatches
      const uint64_t v0 = (uint64_t)x[i], cnt = (uint64_t)y[i];
      uint64_t bad = 0;
      for (uint64_t v = v0; v < v0 + cnt && v <= 0xFFFFFFFFULL; ++v) {
        const double phi = 2.0 * PT_PI * pcg_unit((uint32_t)v);
        double s1, c1;
        sincos(phi, &s1, &c1);
        const double s2 = sin(phi), c2 = cos(phi);
        bad += (__double_as_longlong(s1) != __double_as_longlong(s2)) || (__double_as_longlong(c1) != __double_as_longlong(c2));
      }
      r = (double)bad;
      break;
    }
    default: break;
  }
  out[i] = r;
}
