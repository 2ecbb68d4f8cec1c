// pt_tile.h -- tile culling, the cell pre-pass, the 8x8-tile and the 16x16-tile kernels.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
      pcg_seed_pixel(pcg, c->pcg_mode, c->s0, c->q0, gpix, nsamp);
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
// (A variant with 16x8 tiles and two pixels per lane -- twice the waves, half the pixels each, for frames whose 16x16 tiles do
//  not fill the chip -- was measured slower on every frame and deleted in round 6: profiles/DROPPED_VARIANTS.md.)
#ifndef PT_TILE4_WAVES
#define PT_TILE4_WAVES 4  // waves per SIMD the register allocation aims at (5 / 6 measured: profiles/DROPPED_VARIANTS.md)
#endif
template <int RENDERER, bool SLDS = false>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_TILE4_WAVES, 8))) void pt_tile4_kernel(const PtKArgs a) {
  constexpr int NPX = 4;  // pixels per lane: the four 8x8 quadrants of the tile
  constexpr int TH = 16;  // tile height
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
    const bool okcA = colA < W, okcB = colB < W, okrA = lrowA < rows_local, okrB = lrowB < rows_local;
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
          // (a shape whose two pigments are uniform carries pigment + emitted, added at upload with the same fp64
          //  addition, in pig_c2: pt_layout.h)
          if constexpr (SLDS) {
            const pt_lds_rec rec = (pt_lds_rec)(const void *)pt_lds_f64 + hit;
            const pt_lds_aux ax = (pt_lds_aux)(const void *)(pt_lds_f64 + a.n_shapes * 16) + hit;
            if (ax->needs_uv) {
              hit_details(rec, ax, rk, h4.best_t[k], h, true);
              p1 = brdf_pigment(a, ax, h.u, h.v);
              p2 = emitted_pigment(a, ax, h.u, h.v);
              c.x = p1.x + p2.x;
              c.y = p1.y + p2.y;
              c.z = p1.z + p2.z;
            } else {
              c.x = ax->pig_c2[0];
              c.y = ax->pig_c2[1];
              c.z = ax->pig_c2[2];
            }
          } else {
            const PtShapeAux *ax = cold_args(a)->aux + hit;
            if (ax->needs_uv) {
              hit_details(a.recs + hit, ax, rk, h4.best_t[k], h, true);
              p1 = brdf_pigment(a, ax, h.u, h.v);
              p2 = emitted_pigment(a, ax, h.u, h.v);
              c.x = p1.x + p2.x;
              c.y = p1.y + p2.y;
              c.z = p1.z + p2.z;
            } else {
              c.x = ax->pig_c2[0];
              c.y = ax->pig_c2[1];
              c.z = ax->pig_c2[2];
            }
          }
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
  if ((threadIdx.x & 63) == 0) {  // every wave's cycles, by workgroup and wave (tools/dbgtile4_waves.py: how uneven are a workgroup's four tiles?)
    const int wv = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (int)(threadIdx.x >> 6);
    if (wv < PT_UNITLOG_LEN) {
      unsigned long long tot_ = 0;
      for (int q = 0; q < 8; ++q) tot_ += tsum[q];
      pt_unitlog[wv * 8] = tot_;
      pt_unitlog[wv * 8 + 1] = tsum[0] + tsum[1] + tsum[2] + tsum[3];  // prologue, cone, cull, dome check: per tile
      pt_unitlog[wv * 8 + 2] = __builtin_amdgcn_s_memtime();
    }
  }
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
