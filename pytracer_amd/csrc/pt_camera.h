// pt_camera.h -- primary rays (ImageTracer.fire_ray, Camera.fire_ray), ray counters, per-camera constants.
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
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
