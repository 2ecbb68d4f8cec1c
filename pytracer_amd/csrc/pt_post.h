// pt_post.h — HdrImage post-processing on the device.
//
// What main.py:203-213 does after fire_all_rays, restated for a frame that already sits in HBM:
//   HdrImage.write_pfm          hdrimages.py:96-118   float32 payload, bottom row first
//   HdrImage.average_luminosity hdrimages.py:120-128  10^(mean(log10(delta + (max+min)/2)))
//   HdrImage.normalize_image    hdrimages.py:130-140  pixel * (factor / luminosity)
//   HdrImage.clamp_image        hdrimages.py:142-146  x / (1 + x)
//   HdrImage.write_ldr_image    hdrimages.py:148-171  int(255 * pow(x, 1/gamma)) per channel
// All HBM-bound byte/float shuffles: one coalesced pass each, nothing to tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static __device__ __forceinline__ double pt_px(const void *img, int f32, long long i) {
  return f32 ? (double)((const float *)img)[i] : ((const double *)img)[i];
}

// out[((H-1-y)*W + x)*3 + k] = float32(img[(y*W + x)*3 + k]), optionally byte-swapped (big endian)
__global__ void pt_post_pfm_kernel(const void *img, int f32, int W, int H, int big_endian, uint32_t *out) {
  const long long n = (long long)W * H * 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / ((long long)W * 3), rem = i - row * (long long)W * 3;
    const float v = (float)pt_px(img, f32, i);
    uint32_t b = __float_as_uint(v);
    if (big_endian) b = __builtin_bswap32(b);
    out[(long long)(H - 1 - row) * W * 3 + rem] = b;
  }
}

// Deterministic two-level sum of log10(delta + luminosity): every block owns a fixed contiguous
// chunk of pixels, threads stride through it, a fixed-shape tree adds the 256 partials.
#define PT_POST_CHUNK 8192
__global__ void pt_post_loglum_kernel(const void *img, int f32, long long npix, double delta, double *partials) {
  __shared__ double acc[256];
  const long long base = (long long)blockIdx.x * PT_POST_CHUNK;
  double t = 0.0;
  for (int k = threadIdx.x; k < PT_POST_CHUNK; k += 256) {
    const long long p = base + k;
    if (p < npix) {
      const double r = pt_px(img, f32, p * 3), g = pt_px(img, f32, p * 3 + 1), b = pt_px(img, f32, p * 3 + 2);
      const double mx = fmax(fmax(r, g), b), mn = fmin(fmin(r, g), b);
      t += log10(delta + (mx + mn) / 2);  // colors.py:59-63
    }
  }
  acc[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) acc[threadIdx.x] += acc[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}
__global__ void pt_post_sum_kernel(const double *partials, int n, double *out) {
  __shared__ double acc[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) t += partials[i];
  acc[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) acc[threadIdx.x] += acc[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = acc[0];
}

// normalize + clamp (+ LDR bytes): x' = x*scale; x'' = x'/(1+x'); byte = int(255 * pow(x'', 1/gamma))
__global__ void pt_post_tonemap_kernel(void *img, int f32, long long n, double scale, int do_clamp, double inv_gamma,
                                       unsigned char *rgb8, int write_back) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double x = pt_px(img, f32, i) * scale;
    if (do_clamp) x = x / (1 + x);
    if (write_back) {
      if (f32)
        ((float *)img)[i] = (float)x;
      else
        ((double *)img)[i] = x;
    }
    if (rgb8) {
      const double v = 255 * pow(x, inv_gamma);
      int b = (int)v;  // int() truncates toward zero
      b = b < 0 ? 0 : (b > 255 ? 255 : b);
      rgb8[i] = (unsigned char)b;
    }
  }
}

// ---- the sparse form of a rank's shard for the multi-GPU gather (pytracer_amd/dist.py) -------------------
// A shard of n pixels is cut into runs of PT_SPARSE_RUN consecutive pixels.  A run whose pixels all equal its first one
// BIT FOR BIT travels as that one pixel; the others travel whole, in order.  Lossless, and most of a frame is sky.
//   fixed   = [int64 count of runs that are not constant][int32 per run: its place among those, -1 = constant; padded to
//              a multiple of 8 bytes][first pixel of every run]
//   payload = [count][PT_SPARSE_RUN][3], the last run of the shard filled up with the shard's last pixel
// Encoding is three passes over run-sized pieces, all HBM-bound: classify (a wave per run), scan (one workgroup; the runs
// keep their order), move (a wave per run).  Decoding is ONE pass that also drops the rows into the frame's row-block
// order.  T = uint32_t / uint64_t: bit patterns, never floating-point compares.
#define PT_SPARSE_RUN 128

template <typename T>
__global__ __launch_bounds__(256) void pt_sparse_classify_kernel(const T *px, long long npix, long long nt, int *place, T *firsts) {
  const long long run = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= nt) return;
  const int lane = threadIdx.x & 63;
  const long long base = run * PT_SPARSE_RUN;
  const T f0 = px[base * 3], f1 = px[base * 3 + 1], f2 = px[base * 3 + 2];
  bool same = true;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long p = base + lane + 64 * k;
    if (p < npix) same = same && px[p * 3] == f0 && px[p * 3 + 1] == f1 && px[p * 3 + 2] == f2;
  }
  const bool all_same = __ballot(!same) == 0ULL;
  if (lane == 0) {
    place[run] = all_same ? -1 : 0;
    firsts[run * 3] = f0;
    firsts[run * 3 + 1] = f1;
    firsts[run * 3 + 2] = f2;
  }
}

// place[run] (0 where the run is not constant, -1 where it is) -> its rank among the runs that are not; *count = their number
__global__ __launch_bounds__(1024) void pt_sparse_scan_kernel(int *place, long long nt, long long nt_padded, long long *count) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const long long per = (nt + 1023) / 1024, a = (long long)t * per, b = (a + per < nt) ? a + per : nt;
  int mine = 0;
  for (long long r = a; r < b; ++r) mine += place[r] < 0 ? 0 : 1;
  part[t] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // (inclusive scan, Hillis-Steele: 10 steps on 1024 values)
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run_off = part[t] - mine;
  for (long long r = a; r < b; ++r)
    if (place[r] >= 0) place[r] = run_off++;
  if (t == 1023) *count = (long long)part[1023];
  if (t == 0)
    for (long long r = nt; r < nt_padded; ++r) place[r] = -1;
}

template <typename T>
__global__ __launch_bounds__(256) void pt_sparse_pack_kernel(const T *px, long long npix, long long nt, const int *place, T *payload) {
  const long long run = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= nt || place[run] < 0) return;
  const int lane = threadIdx.x & 63;
  T *dst = payload + (long long)place[run] * PT_SPARSE_RUN * 3;
  const long long base = run * PT_SPARSE_RUN;
  for (int e = lane; e < PT_SPARSE_RUN * 3; e += 64) {  // (consecutive lanes, consecutive words)
    long long p = base + e / 3;
    if (p >= npix) p = npix - 1;
    dst[e] = px[p * 3 + e % 3];
  }
}

// out: the shard itself (n_ranks <= 1), or the frame it belongs to: local row lr of rank `rank` is row
// ((lr / row_block) * n_ranks + rank) * row_block + lr % row_block (abi.rows_for_rank: block b belongs to rank b % n_ranks)
template <typename T>
__global__ __launch_bounds__(256) void pt_sparse_unpack_kernel(const int *place, const T *firsts, const T *payload, long long npix, long long nt,
                                                              T *out, int width, int row_block, int n_ranks, int rank) {
  const long long run = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= nt) return;
  const int lane = threadIdx.x & 63;
  const int pl = place[run];
  const T *src = payload + (long long)(pl < 0 ? 0 : pl) * PT_SPARSE_RUN * 3;
  const T f0 = firsts[run * 3], f1 = firsts[run * 3 + 1], f2 = firsts[run * 3 + 2];
  const long long base = run * PT_SPARSE_RUN;
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // a lane: pixels lane and lane + 64 of the run
    const int j = lane + 64 * k;
    const long long p = base + j;
    if (p >= npix) continue;
    long long q = p;
    if (n_ranks > 1) {  // (the host checked that the shard has fewer than 2^31 pixels: 32-bit divisions)
      const unsigned pu = (unsigned)p, w = (unsigned)width, rb = (unsigned)row_block;
      const unsigned lr = pu / w, col = pu - lr * w, blk = lr / rb;
      q = ((long long)(blk * (unsigned)n_ranks + (unsigned)rank) * rb + (lr - blk * rb)) * w + col;
    }
    T *o = out + q * 3;
    if (pl < 0) {
      o[0] = f0;
      o[1] = f1;
      o[2] = f2;
    } else {
      o[0] = src[j * 3];
      o[1] = src[j * 3 + 1];
      o[2] = src[j * 3 + 2];
    }
  }
}

// ... the shards of several ranks in one launch (blockIdx.y = which): rank 0 of an 8-rank job decodes seven per frame
#define PT_SPARSE_MANY 64
struct PtSparseMany {
  const void *fixed[PT_SPARSE_MANY];
  const void *payload[PT_SPARSE_MANY];
  long long npix[PT_SPARSE_MANY];
  int rank[PT_SPARSE_MANY];
};

template <typename T>
__global__ __launch_bounds__(256) void pt_sparse_unpack_many_kernel(const PtSparseMany m, T *out, int width, int row_block, int n_ranks) {
  const int k = blockIdx.y;
  const long long npix = m.npix[k], nt = (npix + PT_SPARSE_RUN - 1) / PT_SPARSE_RUN, ntp = (nt + 1) / 2 * 2;
  const long long run = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= nt) return;
  const unsigned char *fx = (const unsigned char *)m.fixed[k];
  const int *place = (const int *)(fx + 8);
  const T *firsts = (const T *)(fx + 8 + ntp * 4), *payload = (const T *)m.payload[k];
  const int lane = threadIdx.x & 63, rank = m.rank[k];
  const int pl = place[run];
  const T *src = payload + (long long)(pl < 0 ? 0 : pl) * PT_SPARSE_RUN * 3;
  const T f0 = firsts[run * 3], f1 = firsts[run * 3 + 1], f2 = firsts[run * 3 + 2];
  const long long base = run * PT_SPARSE_RUN;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int j = lane + 64 * kk;
    const long long p = base + j;
    if (p >= npix) continue;
    const unsigned pu = (unsigned)p, w = (unsigned)width, rb = (unsigned)row_block;
    const unsigned lr = pu / w, col = pu - lr * w, blk = lr / rb;
    const long long q = ((long long)(blk * (unsigned)n_ranks + (unsigned)rank) * rb + (lr - blk * rb)) * w + col;
    T *o = out + q * 3;
    if (pl < 0) {
      o[0] = f0;
      o[1] = f1;
      o[2] = f2;
    } else {
      o[0] = src[j * 3];
      o[1] = src[j * 3 + 1];
      o[2] = src[j * 3 + 2];
    }
  }
}
