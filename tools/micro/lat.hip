// Dependent-chain latency of fp64 VALU ops, fp64 div/sqrt, scalar loads, for ONE wave on an idle chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ void chain(double *out, const double *in, int n, const double *tab) {
  double x = in[threadIdx.x], y = in[64 + threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (OP == 0) x = x * y + 1e-9;             // mul + add (2 dependent ops, no fma: contract off)
    else if (OP == 1) x = 1.0 / (x + 1.5);     // add + div
    else if (OP == 2) x = sqrt(x + 2.0);       // add + sqrt
    else if (OP == 3) x = sin(x) + 0.5;        // add + sin
    else if (OP == 4) { const __attribute__((address_space(4))) double *p = (const __attribute__((address_space(4))) double *)(const void *)(tab + ((i * 37) & 1023) * 8); x = x * p[0] + p[1]; }  // scalar load per iter
    else if (OP == 5) { x = x * tab[(((int)x) & 1023) * 8 + (threadIdx.x & 7)] + 1.0; }  // dependent vector load
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0);
}
int main() {
  double *in, *out, *tab; double h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.3 + 0.001 * i;
  CHK(hipMalloc(&in, sizeof h)); CHK(hipMalloc(&out, 65 * 8)); CHK(hipMalloc(&tab, 1024 * 64));
  CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice)); CHK(hipMemset(tab, 0, 1024 * 64));
  const int n = 2000; double r[65];
  const char *names[] = {"mul+add", "add+div", "add+sqrt", "add+sin", "s_load+mul+add", "dep vload+mul+add"};
  for (int op = 0; op < 6; ++op) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (op) { case 0: chain<0><<<1, 64>>>(out, in, n, tab); break; case 1: chain<1><<<1, 64>>>(out, in, n, tab); break;
        case 2: chain<2><<<1, 64>>>(out, in, n, tab); break; case 3: chain<3><<<1, 64>>>(out, in, n, tab); break;
        case 4: chain<4><<<1, 64>>>(out, in, n, tab); break; default: chain<5><<<1, 64>>>(out, in, n, tab); }
      CHK(hipDeviceSynchronize());
    }
    CHK(hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost));
    printf("%-20s %8.1f memtime ticks / iteration (100 MHz ticks? see clock)\n", names[op], r[64] / n);
  }
  return 0;
}
