// What do s_memtime / s_memrealtime tick at, and how long is a dependent fp64 op, on an idle chip
// (one wave) and on a busy one (every SIMD occupied)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void chain(double *out, const double *in, int n) {
  double x = in[threadIdx.x & 63], y = in[64 + (threadIdx.x & 63)];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) x = x * y + 1e-9;  // 2 dependent fp64 ops (contract off)
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (x == 12345.0) out[0] = x;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[64] = (double)(t1 - t0); out[65] = (double)(r1 - r0); }
}
int main() {
  double *in, *out; double h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.3 + 0.001 * i;
  CHK(hipMalloc(&in, sizeof h)); CHK(hipMalloc(&out, 66 * 8));
  CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int n = 200000; double r[66];
  const int grids[] = {1, 1, 1024, 1024, 4096, 1};
  const int blocks[] = {64, 64, 256, 256, 256, 64};
  for (int k = 0; k < 6; ++k) {
    CHK(hipEventRecord(e0)); chain<<<grids[k], blocks[k]>>>(out, in, n); CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost));
    printf("grid %5d x %3d: %.3f ms; memtime %.0f ticks (%.1f MHz), realtime %.0f ticks (%.1f MHz); %.2f ns per dependent fp64 op\n",
           grids[k], blocks[k], ms, r[64], r[64] / ms / 1e3, r[65], r[65] / ms / 1e3, ms * 1e6 / (2.0 * n));
  }
  return 0;
}
