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
