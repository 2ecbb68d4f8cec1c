// How long does a launch of G workgroups take to get going and to drain on MI355X (gfx950)?  Every wave spins for a fixed
// number of cycles (s_memtime); the time per launch (hipEvents around a batch of launches on one stream) minus the spin
// is what dispatch, ramp and drain cost for that launch shape.  (Wave time stamps are not compared across the chip: the
// XCDs' counters do not share an origin.)
//
//   ./ramp            table over launch shapes (workgroups x threads, LDS per workgroup, registers per lane)
// Build: hipcc --offload-arch=gfx950 -O2 -o ramp tools/micro/ramp.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV>
__global__ void __launch_bounds__(256) spin_kernel(unsigned long long spin, unsigned long long *stamps, float *sink) {
  extern __shared__ char lds_claim[];
  // census: waves running at this moment / the most there ever were (stamps[0], stamps[1])
  if (stamps && (threadIdx.x & 63) == 0) {
    const unsigned long long c = atomicAdd(stamps, 1ULL) + 1ULL;
    atomicMax(stamps + 1, c);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float v[NV];  // registers the launch has to hand out
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = (float)(threadIdx.x + k);
  unsigned long long t = t0;
  while (t - t0 < spin) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] * 1.0001f + 0.5f;
    t = __builtin_amdgcn_s_memtime();
  }
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) acc += v[k];
  if (acc == 123.456f) sink[0] = acc;
  if (stamps && (threadIdx.x & 63) == 0) atomicAdd(stamps, ~0ULL);  // (-1)
}

template <int NV>
static int run(int G, int B, int lds, unsigned long long spin, double clock_hz) {
  unsigned long long *stamps = nullptr;
  float *sink = nullptr;
  const size_t nw = (size_t)G * (B / 64);
  CHK(hipMalloc(&stamps, nw * 16));
  CHK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipFuncSetAttribute((const void *)spin_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(spin_kernel<NV>, dim3(G), dim3(B), lds, 0, spin, (unsigned long long *)nullptr, sink);
  CHK(hipDeviceSynchronize());
  CHK(hipMemset(stamps, 0, 16));
  hipLaunchKernelGGL(spin_kernel<NV>, dim3(G), dim3(B), lds, 0, spin, stamps, sink);
  CHK(hipDeviceSynchronize());
  unsigned long long census[2] = {0, 0};
  CHK(hipMemcpy(census, stamps, 16, hipMemcpyDeviceToHost));
  const int K = 200;
  CHK(hipEventRecord(e0));
  for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin_kernel<NV>, dim3(G), dim3(B), lds, 0, spin, (unsigned long long *)nullptr, sink);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0.0f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  // the same K launches captured in a hipGraph and replayed: does the graph shorten the gap between dependent kernels?
  float ms_graph = -1.0f;
  {
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin_kernel<NV>, dim3(G), dim3(B), lds, st, spin, (unsigned long long *)nullptr, sink);
    CHK(hipStreamEndCapture(st, &graph));
    CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(exec, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    CHK(hipGraphLaunch(exec, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms_graph, e0, e1));
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(st);
  }
  int regs = 0, resident = 0;
  {
    hipFuncAttributes fa;
    CHK(hipFuncGetAttributes(&fa, (const void *)spin_kernel<NV>));
    regs = fa.numRegs;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)spin_kernel<NV>, B, lds));
  }
  const double tick_us = 1e6 / clock_hz;
  printf("%5d x %3d threads  LDS %6d B  %3d VGPRs (runtime: %d workgroups fit a CU; most waves running at once: %llu of %llu)  every wave spins %5.2f us | %6.2f us per launch, back to back -> %5.2f us beyond the spin; as a hipGraph of %d nodes: %6.2f us per launch\n", G, B,
         lds, regs, resident, census[1], (unsigned long long)nw, spin * tick_us, ms * 1e3 / K, ms * 1e3 / K - spin * tick_us, K, ms_graph * 1e3 / K);
  (void)hipFree(stamps);
  (void)hipFree(sink);
  return 0;
}

int main() {
  // the clock s_memtime runs at: measured against hipEvents
  double clock_hz = 1e8;
  {
    unsigned long long *stamps = nullptr;
    float *sink = nullptr;
    CHK(hipMalloc(&stamps, 16));
    CHK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const unsigned long long spin = 2000000ULL;
    hipLaunchKernelGGL(spin_kernel<8>, dim3(1), dim3(64), 0, 0, 1000ULL, (unsigned long long *)nullptr, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin_kernel<8>, dim3(1), dim3(64), 0, 0, spin, (unsigned long long *)nullptr, sink);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    clock_hz = (double)spin / (ms * 1e-3);
    printf("s_memtime: %.1f MHz (a spin of %llu ticks took %.3f ms)\n", clock_hz / 1e6, spin, ms);
  }
  const unsigned long long us8 = (unsigned long long)(8e-6 * clock_hz), us0 = 0ULL;
  printf("-- no work: what the launch itself costs\n");
  run<8>(1, 64, 0, us0, clock_hz);
  run<8>(920, 256, 0, us0, clock_hz);
  run<8>(920, 256, 12672, us0, clock_hz);
  run<40>(920, 256, 12672, us0, clock_hz);
  printf("-- every wave spins 8 us (the 16x16-tile kernel's waves live ~8 us)\n");
  run<8>(256, 256, 0, us8, clock_hz);
  run<8>(920, 256, 0, us8, clock_hz);
  run<8>(920, 256, 12672, us8, clock_hz);
  run<40>(920, 256, 12672, us8, clock_hz);
  run<40>(1840, 128, 6336, us8, clock_hz);
  run<40>(3680, 64, 3168, us8, clock_hz);
  run<40>(1024, 256, 12672, us8, clock_hz);
  run<40>(512, 256, 12672, us8, clock_hz);
  run<88>(920, 256, 12672, us8, clock_hz);
  return 0;
}
