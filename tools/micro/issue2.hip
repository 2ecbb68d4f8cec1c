// Issue cost (SIMD-cycles per wave64 instruction) of the opcodes tools/isa_mix.py finds in the kernels' disassembly that
// tools/micro/issue.hip had not measured: division helpers, ldexp / floor / trunc, class and integer compares, 64-bit moves,
// select on VCC, conversions to / from int, lane reads.  Same method as issue.hip: every wave runs CH independent chains of ONE
// instruction (inline asm) between two s_memtime stamps, 256 x W workgroups of four waves, each claiming 160 KB / W of LDS
// (W workgroups per CU = W waves per SIMD; checked through HW_ID).  Build: hipcc --offload-arch=gfx950 -O2 -o issue2
// tools/micro/issue2.hip ; run: ./issue2            (ISSUE_WS=24 picks the residencies, default 2 and 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int CH = 16, UNROLL = 4, ITER = 1500;

#define OPS(X)                                                                                                               \
  X(MOV_B64, "v_mov_b64", asm volatile("v_mov_b64 %0, %1" : "=v"(d[k]) : "v"(d[(k + 1) % CH])))                             \
  X(DIV_SCALE_F64, "v_div_scale_f64", asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[k]) : "v"(one) : "vcc"))  \
  X(DIV_FMAS_F64, "v_div_fmas_f64", asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(one), "v"(z)))          \
  X(DIV_FIXUP_F64, "v_div_fixup_f64", asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(one)))               \
  X(LDEXP_F64, "v_ldexp_f64", asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(zero_u)))                            \
  X(FLOOR_F64, "v_floor_f64", asm volatile("v_floor_f64 %0, %0" : "+v"(d[k])))                                              \
  X(TRUNC_F64, "v_trunc_f64", asm volatile("v_trunc_f64 %0, %0" : "+v"(d[k])))                                              \
  X(FMAC_F64, "v_fmac_f64", asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[k]) : "v"(z), "v"(one)))                          \
  X(CMP_CLASS_F64, "v_cmp_class_f64", asm volatile("v_cmp_class_f64 %0, %1, %2" : "+s"(m[k & 7]) : "v"(d[k]), "v"(u[k])))   \
  X(CMP_LT_I32, "v_cmp_lt_i32 (e64)", asm volatile("v_cmp_lt_i32 %0, %1, %2" : "+s"(m[k & 7]) : "v"(u[k]), "v"(u[(k + 1) % CH]))) \
  X(CMP_LT_F64_VCC, "v_cmp_lt_f64 (e32, vcc)", asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[k]), "v"(one) : "vcc"))    \
  X(CNDMASK_VCC, "v_cndmask_b32 (e32, vcc)", asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % CH]))) \
  X(CNDMASK_VCC_SET, "v_cndmask_b32 (e32, vcc = exec set before the loop)", asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % CH])))   \
  X(CNDMASK_E64_VCC, "v_cndmask_b32_e64 (vcc as the SGPR operand)", asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % CH])))   \
  X(CNDMASK_VCC_NODEP, "v_cndmask_b32 (e32, vcc; independent destinations)", asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(w32), "v"(zero_u)))   \
  X(CNDMASK_SGPR, "v_cndmask_b32 (e64, SGPR pair; issue.hip's row)", asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) % CH]), "s"(m[0])))   \
  X(ADDC_VCC, "v_addc_co_u32 (e32: vcc in and out)", asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) % CH]) : "vcc"))   \
  X(MAX_F32, "v_max_f32", asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(onef)))                                     \
  X(MAX3_F32, "v_max3_f32", asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(onef)))                             \
  X(LDEXP_F32, "v_ldexp_f32", asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zero_u)))                            \
  X(LSHL_ADD_U64, "v_lshl_add_u64", asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k]) : "v"(q[(k + 1) % CH])))       \
  X(CVT_I32_F64, "v_cvt_i32_f64", asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[k]) : "v"(d[k])))                            \
  X(CVT_F64_I32, "v_cvt_f64_i32", asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(u[k])))                            \
  X(CVT_F32_I32, "v_cvt_f32_i32", asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(u[k])))                            \
  X(LSHLREV_B32, "v_lshlrev_b32", asm volatile("v_lshlrev_b32 %0, 0, %0" : "+v"(u[k])))                                     \
  X(BFE_U32, "v_bfe_u32", asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(zero_u), "v"(w32)))                    \
  X(READFIRSTLANE, "v_readfirstlane_b32", asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s[k & 3]) : "v"(u[k])))          \
  X(WRITELANE, "v_writelane_b32", asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(u[k]) : "s"(s[k & 3])))                   \
  X(ADD_F64_REF, "v_add_f64 (reference row)", asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(z)))                   \
  X(ADD_F32_REF, "v_add_f32 (reference row)", asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(zf)))

enum Op {
#define X(ID, NAME, ASM) ID,
  OPS(X)
#undef X
  N_OPS };
static const char *NAMES[N_OPS] = {
#define X(ID, NAME, ASM) NAME,
  OPS(X)
#undef X
};

template <int OP>
__global__ void __launch_bounds__(256) issue_kernel(unsigned long long *stamps, unsigned *ids, double *sink, const double *in) {
  extern __shared__ char lds_claim[];
  double d[CH];
  float f[CH];
  unsigned u[CH];
  unsigned long long q[CH];
  unsigned s[4] = {1, 2, 3, 4};
  unsigned long long m[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const double one = in[0], z = in[1];
  const float onef = (float)in[0], zf = (float)in[1];
  const unsigned zero_u = (unsigned)in[1], w32 = 32u + (unsigned)in[1];
#pragma unroll
  for (int k = 0; k < CH; ++k) { d[k] = in[2 + k]; f[k] = (float)in[2 + k]; u[k] = threadIdx.x + k; q[k] = threadIdx.x * 77ull + k; }
  if (OP == CNDMASK_VCC_SET || OP == CNDMASK_E64_VCC || OP == CNDMASK_VCC_NODEP) asm volatile("s_mov_b64 vcc, exec" : : : "vcc");
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < UNROLL; ++r) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
#define X(ID, NAME, ASM) if (OP == ID) { ASM; }
        OPS(X)
#undef X
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double acc = 0;
#pragma unroll
  for (int k = 0; k < CH; ++k) acc += d[k] + (double)f[k] + (double)u[k] + (double)q[k];
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
static int run_op(int W, int n_cu, unsigned long long *stamps, unsigned *ids, double *sink, const double *in) {
  const int lds = std::min(65536, (160 * 1024) / W - 512);
  const int blocks = n_cu * W;
  CHK(hipFuncSetAttribute((const void *)issue_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  issue_kernel<OP><<<blocks, 256, lds>>>(stamps, ids, sink, in);
  CHK(hipDeviceSynchronize());
  issue_kernel<OP><<<blocks, 256, lds>>>(stamps, ids, sink, in);
  CHK(hipDeviceSynchronize());
  const int waves = blocks * 4;
  std::vector<unsigned long long> st(2 * waves); std::vector<unsigned> id(2 * waves);
  CHK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(id.data(), ids, id.size() * 4, hipMemcpyDeviceToHost));
  std::map<unsigned long long, int> per_simd;
  for (int w = 0; w < waves; ++w) per_simd[((unsigned long long)(id[2 * w + 1] & 0xf) << 32) | (id[2 * w] & 0xff30u)]++;
  const double n_instr = (double)ITER * UNROLL * CH;
  std::vector<double> cyc;
  for (int w = 0; w < waves; ++w)
    if (per_simd[((unsigned long long)(id[2 * w + 1] & 0xf) << 32) | (id[2 * w] & 0xff30u)] == W) cyc.push_back((double)(st[2 * w + 1] - st[2 * w]) / n_instr);
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc.empty() ? 0 : cyc[cyc.size() / 2];
  printf("%-30s W=%d  cycles/instr: wave %.3f  SIMD %.3f  kept %zu/%d waves\n", NAMES[OP], W, med, med / W, cyc.size(), waves);
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s: %d CUs\n", prop.name, n_cu);
  unsigned long long *stamps; unsigned *ids; double *sink, *in;
  const int max_waves = n_cu * 8 * 4;
  CHK(hipMalloc(&stamps, max_waves * 16)); CHK(hipMalloc(&ids, max_waves * 8)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&in, 64 * 8));
  double h[64]; h[0] = 1.0; h[1] = 0.0; for (int i = 2; i < 64; ++i) h[i] = 1.0 + 0.01 * i;
  CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
  std::vector<int> Ws = {2, 4};
  if (const char *e = getenv("ISSUE_WS")) { Ws.clear(); for (const char *q = e; *q; ++q) if (*q >= '1' && *q <= '8') Ws.push_back(*q - '0'); }
#define X(ID, NAME, ASM) for (int W : Ws) if (run_op<ID>(W, n_cu, stamps, ids, sink, in)) return 1;
  OPS(X)
#undef X
  return 0;
}
