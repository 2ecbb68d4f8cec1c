// Issue cost of vector instructions on MI355X (gfx950): SIMD-cycles per wave64 instruction, by instruction class
// and by the number of waves sharing a SIMD, with every CU busy.
//
// Every wave runs CH independent dependency chains of ONE instruction (inline asm, so the stream is exactly what
// the table says), ITER x UNROLL x CH instructions between two s_memtime stamps.  A launch is 256 x W workgroups of
// four waves; each workgroup claims 160 KB / W of LDS, so a CU holds at most W of them = W waves per SIMD when
// the dispatcher spreads them evenly -- which is checked, not assumed: every wave records HW_ID / XCC_ID and the
// host counts the waves that shared a SIMD.
//
//   cycles/instr (wave)  = (t1 - t0) / instructions           what ONE wave sees
//   cycles/instr (SIMD)  = that / waves on its SIMD             what the instruction costs the SIMD: the roofline price
//
// Build: hipcc --offload-arch=gfx950 -O2 -o issue tools/micro/issue.hip ; run: ./issue [json-out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static double g_clock_hz = 2.4e9;
constexpr int CH = 16, UNROLL = 4, ITER = 1500;

enum Op { MUL_F64, ADD_F64, FMA_F64, ADD_F32, MUL_F32, FMA_F32, PK_MUL_F32, PK_FMA_F32, PK_ADD_F32, ADD_U32, AND_B32, MOV_B32,
          LSHL_B64, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, SQRT_F32, RCP_F32, RSQ_F64, RCP_F64, SQRT_F64, CVT_F64_F32, CVT_F32_F64,
          CMP_F64, CMP_F32, CNDMASK, MIX_F64_U32, MIX_F64_F32, MIX_F64_SALU, MIX_U32_SALU, SALU_ONLY, READLANE, MBCNT,
          MUL_F64_SGPR, MIX_F32_SALU_2_1, MIX_F32_SALU_4_1, S_MOV, S_AND_B64, N_OPS };
static const char *NAMES[N_OPS] = {
    "v_mul_f64", "v_add_f64", "v_fma_f64", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32",
    "v_add_u32", "v_and_b32", "v_mov_b32", "v_lshlrev_b64", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_sqrt_f32", "v_rcp_f32",
    "v_rsq_f64", "v_rcp_f64", "v_sqrt_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cmp_lt_f64", "v_cmp_lt_f32", "v_cndmask_b32",
    "v_mul_f64+v_add_u32 (1:1)", "v_mul_f64+v_add_f32 (1:1)", "v_mul_f64+s_add_u32 (1:1)", "v_add_u32+s_add_u32 (1:1)", "s_add_u32",
    "v_readlane_b32", "v_mbcnt_lo_u32_b32", "v_mul_f64 (SGPR operand)", "v_add_f32+s_add_u32 (2:1)", "v_add_f32+s_add_u32 (4:1)", "s_mov_b32",
    "s_and_b64"};
// instructions per chain step (for the mixes: both count)
static const double PER_STEP[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 1, 1, 1, 1, 1.5, 1.25, 1, 1};

template <int OP>
__global__ void __launch_bounds__(256) issue_kernel(unsigned long long *stamps, unsigned *ids, double *sink, const double *in) {
  extern __shared__ char lds_claim[];
  double d[CH];
  float f[CH];
  unsigned u[CH];
  unsigned long long q[CH];
  unsigned s[4] = {1, 2, 3, 4};
  unsigned long long m[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const double one_s = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(unsigned long long, in[0]) >> 32)) * 4294967296ll);
  const double one = in[0];  // 1.0: products and sums stay finite and normal
  const float onef = (float)in[0];
  const float zf = (float)in[1];
  const double z = in[1];    // 0.0
#pragma unroll
  for (int k = 0; k < CH; ++k) { d[k] = in[2 + k]; f[k] = (float)in[2 + k]; u[k] = threadIdx.x + k; q[k] = threadIdx.x * 77ull + k; }
  float2 p[CH / 2];
#pragma unroll
  for (int k = 0; k < CH / 2; ++k) p[k] = make_float2(f[2 * k], f[2 * k + 1]);
  const float2 onep = make_float2(onef, onef), zp = make_float2(zf, zf);
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < UNROLL; ++r) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one));
        else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(z));
        else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(one), "v"(z));
        else if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zf));
        else if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(onef));
        else if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(onef), "v"(zf));
        else if (OP == PK_MUL_F32) { if (k < CH / 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(onep)); else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k - CH / 2]) : "v"(onep)); }
        else if (OP == PK_FMA_F32) { if (k < CH / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(onep), "v"(zp)); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k - CH / 2]) : "v"(onep), "v"(zp)); }
        else if (OP == PK_ADD_F32) { if (k < CH / 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(zp)); else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k - CH / 2]) : "v"(zp)); }
        else if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % CH]));
        else if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(0xffffffffu));
        else if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) % CH]));
        else if (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 0, %0" : "+v"(q[k]));
        else if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(u[k]), "v"(u[(k + 1) % CH]) : "vcc");
        else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % CH]));
        else if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % CH]));
        else if (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
        else if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
        else if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
        else if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
        else if (OP == SQRT_F64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[k]));
        else if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
        else if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
        else if (OP == CMP_F64) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "+s"(m[k & 7]) : "v"(d[k]), "v"(one));  // (e64: any SGPR pair, no WAW on vcc)
        else if (OP == CMP_F32) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "+s"(m[k & 7]) : "v"(f[k]), "v"(onef));
        else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) % CH]), "s"(m[0]));
        else if (OP == MUL_F64_SGPR) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "s"(one_s));
        else if (OP == MIX_F32_SALU_2_1) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zf)); if (k & 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[(k >> 1) & 3]) : : "scc"); }
        else if (OP == MIX_F32_SALU_4_1) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zf)); if ((k & 3) == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[(k >> 2) & 3]) : : "scc"); }
        else if (OP == S_MOV) asm volatile("s_mov_b32 %0, %1" : "=s"(s[k & 3]) : "s"(s[(k + 1) & 3]));
        else if (OP == S_AND_B64) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[k & 7]) : "s"(m[(k + 1) & 7]) : "scc");
        else if (OP == MIX_F64_U32) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % CH])); }
        else if (OP == MIX_F64_F32) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zf)); }
        else if (OP == MIX_F64_SALU) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[k & 3]) : : "scc"); }
        else if (OP == MIX_U32_SALU) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % CH])); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[k & 3]) : : "scc"); }
        else if (OP == SALU_ONLY) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[k & 3]) : : "scc");
        else if (OP == READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[k & 3]) : "v"(u[k]));
        else if (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u[k]) : "s"(s[0]));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double acc = 0;
#pragma unroll
  for (int k = 0; k < CH; ++k) acc += d[k] + (double)f[k] + (double)u[k] + (double)q[k];
#pragma unroll
  for (int k = 0; k < CH / 2; ++k) acc += (double)p[k].x + (double)p[k].y;
  acc += (double)(s[0] + s[1] + s[2] + s[3]);
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += (double)m[k];
  if (acc == 123.456) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[2 * w] = t0; stamps[2 * w + 1] = t1;
    ids[2 * w] = hw; ids[2 * w + 1] = xcc;
  }
  if (lds_claim[0] == 77 && acc == 1.0) sink[1] = 1.0;
}

template <int OP>
static int run_op(int W, int n_cu, unsigned long long *stamps, unsigned *ids, double *sink, const double *in, FILE *js, bool first) {
  const int lds = std::min(65536, (160 * 1024) / W - 512);  // at most W workgroups per CU
  const int blocks = n_cu * W;
  CHK(hipFuncSetAttribute((const void *)issue_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  issue_kernel<OP><<<blocks, 256, lds>>>(stamps, ids, sink, in);  // warm
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  issue_kernel<OP><<<blocks, 256, lds>>>(stamps, ids, sink, in);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = blocks * 4;
  std::vector<unsigned long long> st(2 * waves); std::vector<unsigned> id(2 * waves);
  CHK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(id.data(), ids, id.size() * 4, hipMemcpyDeviceToHost));
  // waves per (xcc, se, sh, cu, simd)
  std::map<unsigned long long, int> per_simd;
  for (int w = 0; w < waves; ++w) {
    const unsigned hw = id[2 * w], xcc = id[2 * w + 1] & 0xf;
    const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xff30u);  // se, sh, cu, simd bits of HW_ID
    per_simd[key]++;
  }
  const double n_instr = (double)ITER * UNROLL * CH * PER_STEP[OP];
  // keep the waves that really had W-1 neighbours on their SIMD
  double sum = 0; int kept = 0; std::vector<double> cyc;
  int hist[20] = {0};
  for (int w = 0; w < waves; ++w) {
    const unsigned hw = id[2 * w], xcc = id[2 * w + 1] & 0xf;
    const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xff30u);
    const int n = per_simd[key];
    hist[std::min(n, 19)]++;
    if (n == W) { const double c = (double)(st[2 * w + 1] - st[2 * w]) / n_instr; sum += c; ++kept; cyc.push_back(c); }
  }
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc.empty() ? 0 : cyc[cyc.size() / 2];
  const double mean = kept ? sum / kept : 0;
  // s_memtime ticks are shader cycles (MI355X_MICROARCH.md, constants table); the launch's event time gives the same
  // figure from the outside: launch_ms x clock / instructions / W (includes ramp and tail)
  const double ev = ms * 1e-3 * g_clock_hz / n_instr / W;
  printf("%-30s W=%d  cycles/instr: wave %.3f  SIMD %.3f  (events: SIMD %.3f)  kept %d/%d waves  SIMDs seen %zu  launch %.3f ms\n", NAMES[OP], W, med, med / W, ev, kept, waves, per_simd.size(), ms);
  if (js) {
    fprintf(js, "%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_wave\": %.4f, \"cycles_per_instr_simd\": %.4f, \"cycles_per_instr_simd_from_events\": %.4f, \"mean_wave\": %.4f, \"waves_kept\": %d, \"waves\": %d, \"simds_seen\": %zu, \"launch_ms\": %.4f, \"instr_per_wave\": %.0f}",
            first ? "" : ",", NAMES[OP], W, med, med / W, ev, mean, kept, waves, per_simd.size(), ms, n_instr);
  }
  return 0;
}

template <int OP>
static int run_all(const std::vector<int> &Ws, int n_cu, unsigned long long *stamps, unsigned *ids, double *sink, const double *in, FILE *js, bool &first) {
  for (int W : Ws) { if (run_op<OP>(W, n_cu, stamps, ids, sink, in, js, first)) return 1; first = false; }
  return 0;
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  int clock_khz = 0; CHK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  g_clock_hz = clock_khz * 1e3;
  printf("%s: %d CUs, clock %d kHz, wall clock %d kHz\n", prop.name, n_cu, clock_khz, wall_khz);
  FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"wall_clock_khz\": %d, \"chains\": %d, \"rows\": [", prop.name, n_cu, clock_khz, wall_khz, CH);
  unsigned long long *stamps; unsigned *ids; double *sink, *in;
  const int max_waves = n_cu * 8 * 4;
  CHK(hipMalloc(&stamps, max_waves * 16)); CHK(hipMalloc(&ids, max_waves * 8)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&in, 64 * 8));
  double h[64]; h[0] = 1.0; h[1] = 0.0; for (int i = 2; i < 64; ++i) h[i] = 1.0 + 0.01 * i;
  CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
  std::vector<int> Ws = {1, 2, 3, 4, 5, 8};
  if (const char *e = getenv("ISSUE_WS")) {  // e.g. ISSUE_WS=4: one residency only (PMC passes over the same streams)
    Ws.clear();
    for (const char *q = e; *q; ++q) if (*q >= '1' && *q <= '8') Ws.push_back(*q - '0');
  }
  bool first = true;
#define RUN(OP) if (run_all<OP>(Ws, n_cu, stamps, ids, sink, in, js, first)) return 1;
  RUN(MUL_F64) RUN(ADD_F64) RUN(FMA_F64) RUN(ADD_F32) RUN(MUL_F32) RUN(FMA_F32) RUN(PK_MUL_F32) RUN(PK_FMA_F32) RUN(PK_ADD_F32)
  RUN(ADD_U32) RUN(AND_B32) RUN(MOV_B32) RUN(LSHL_B64) RUN(MAD_U64_U32) RUN(MUL_LO_U32) RUN(MUL_HI_U32) RUN(SQRT_F32) RUN(RCP_F32)
  RUN(RSQ_F64) RUN(RCP_F64) RUN(SQRT_F64) RUN(CVT_F64_F32) RUN(CVT_F32_F64) RUN(CMP_F64) RUN(CMP_F32) RUN(CNDMASK)
  RUN(MIX_F64_U32) RUN(MIX_F64_F32) RUN(MIX_F64_SALU) RUN(MIX_U32_SALU) RUN(SALU_ONLY) RUN(READLANE) RUN(MBCNT)
  RUN(MUL_F64_SGPR) RUN(MIX_F32_SALU_2_1) RUN(MIX_F32_SALU_4_1) RUN(S_MOV) RUN(S_AND_B64)
  if (js) { fprintf(js, "\n]}\n"); fclose(js); }
  return 0;
}
