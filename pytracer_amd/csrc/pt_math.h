// pt_math.h -- PCG-XSH-RR and Transformation products (pcg.py, transformations.py).
// A part of pt_kernels.h (which includes the parts in order: each relies on the ones before it); not a header of its own.
// ---- PCG-XSH-RR 64/32 (pcg.py:23-62) -----------------------------------------------------------
struct Pcg {
  uint64_t state, inc;
  unsigned n;  // draws since the counter was last cleared (only the path tracer's second pass reads it)
};
PT_DEV uint32_t pcg_next(Pcg &p) {
  const uint64_t old = p.state;
  p.n++;
  p.state = old * 6364136223846793005ULL + p.inc;
  const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
  const uint32_t rot = (uint32_t)(old >> 59);
  return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
PT_DEV void pcg_seed(Pcg &p, uint64_t init_state, uint64_t init_seq) {
  p.state = 0;
  p.inc = (init_seq << 1) | 1ULL;
  pcg_next(p);
  p.state += init_state;
  pcg_next(p);
  p.n = 0;
}
// The state `delta` draws further on: state -> state * M^delta + inc * (M^(delta-1) + ... + 1) mod 2^64, by
// repeated squaring (the generator is a linear congruential one; identical to `delta` calls of pcg_next).
PT_DEV uint64_t pcg_advance(uint64_t state, uint64_t inc, unsigned delta) {
  uint64_t acc_mul = 1ULL, acc_add = 0ULL, cur_mul = 6364136223846793005ULL, cur_add = inc;
  while (delta) {
    if (delta & 1u) {
      acc_mul *= cur_mul;
      acc_add = acc_add * cur_mul + cur_add;
    }
    cur_add = (cur_mul + 1ULL) * cur_add;
    cur_mul *= cur_mul;
    delta >>= 1;
  }
  return acc_mul * state + acc_add;
}
// The same jump with the increment factored out: state -> A * state + inc * G for ANY generator of the family, so the two
// coefficients of a distance that is known in advance are computed once (the 2 N scatter draws of N children beyond
// max_depth, render.py:100-101 / 128: path_trace, path_tree; the leaf rounds' hypotheses: path_tree's table).
struct PcgJump {
  uint64_t A, G;
};
PT_DEV PcgJump pcg_jump_coeffs(unsigned delta) {
  uint64_t acc_mul = 1ULL, acc_g = 0ULL, cur_mul = 6364136223846793005ULL, cur_g = 1ULL;
  while (delta) {
    if (delta & 1u) {
      acc_mul *= cur_mul;
      acc_g = acc_g * cur_mul + cur_g;
    }
    cur_g = (cur_mul + 1ULL) * cur_g;
    cur_mul *= cur_mul;
    delta >>= 1;
  }
  PcgJump j = {acc_mul, acc_g};
  return j;
}
PT_DEV void pcg_jump(Pcg &p, PcgJump j, unsigned delta) {  // == `delta` calls of pcg_next whose outputs nobody reads
  p.state = j.A * p.state + p.inc * j.G;
  p.n += delta;
}
// ... with a 64-bit distance: where sample k of pixel i starts in ONE sequential stream that every sample draws the same
// count from (PT_PCG_SEQ: ImageTracer.pcg, two jitter numbers per sample, imagetracer.py:84-101) -- a 4K frame at 256
// samples per pixel is 4.2e9 draws in.
PT_DEV uint64_t pcg_advance64(uint64_t state, uint64_t inc, uint64_t delta) {
  uint64_t acc_mul = 1ULL, acc_add = 0ULL, cur_mul = 6364136223846793005ULL, cur_add = inc;
  while (delta) {
    if (delta & 1ULL) {
      acc_mul *= cur_mul;
      acc_add = acc_add * cur_mul + cur_add;
    }
    cur_add = (cur_mul + 1ULL) * cur_add;
    cur_mul *= cur_mul;
    delta >>= 1;
  }
  return acc_mul * state + acc_add;
}
// The generator of pixel `gpix` before its first sample (S > 0), by alignment (a.s0 / a.q0 are the
// path seeds, or under PT_PCG_SEQ the jitter seeds): PIXEL -- PCG(S0, Q0 + i); SEQ -- the ONE generator PCG(S0, Q0) of
// the reference's ImageTracer, 2 * S^2 * i draws in (renderers without a scattering stream: every sample before this
// pixel drew exactly its two jitter numbers); SAMPLE seeds per sample instead.
PT_DEV void pcg_seed_pixel(Pcg &p, int pcg_mode, uint64_t s0, uint64_t q0, unsigned long long gpix, int nsamp) {
  if (pcg_mode == PT_PCG_PIXEL) {
    pcg_seed(p, s0, q0 + gpix);
  } else if (pcg_mode == PT_PCG_SEQ) {
    pcg_seed(p, s0, q0);
    p.state = pcg_advance64(p.state, p.inc, 2ULL * (unsigned long long)nsamp * gpix);
  }
}
// pcg.py:60-62: random() / 0xFFFFFFFF, an fp64 division (inclusive 1.0) -- of a 32-bit integer by a CONSTANT: the quotient is
// formed from the constant's correctly rounded reciprocal and one correction step (a multiplication and two explicit fused
// multiply-adds -- not a contraction: the expression is not the reference's, its VALUE is, for every one of the 2^32 possible
// inputs: tests/proofs/pcg_float_div.c checks them all) instead of the dozen dependent instructions of a general division.
PT_DEV double pcg_unit(uint32_t v) {
  const double x = (double)v;
  const double r = 0x1.00000001p-32;  // RN(1 / 4294967295.0)
  const double q0 = x * r;
  const double e = __builtin_fma(-4294967295.0, q0, x);
  return __builtin_fma(e, r, q0);
}
PT_DEV double pcg_float(Pcg &p) { return pcg_unit(pcg_next(p)); }

// ---- transformations.py:58-86 ----------------------------------------------------------------------
template <typename P>
PT_DEV V3 xf_point(P m, V3 p) {
  V3 r;
  r.x = p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3];
  r.y = p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7];
  r.z = p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11];
  return r;
}
template <typename P>
PT_DEV V3 xf_vec(P m, V3 v) {
  V3 r;
  r.x = v.x * m[0] + v.y * m[1] + v.z * m[2];
  r.y = v.x * m[4] + v.y * m[5] + v.z * m[6];
  r.z = v.x * m[8] + v.y * m[9] + v.z * m[10];
  return r;
}
template <typename P>
PT_DEV V3 xf_normal(P im, V3 n) {  // transpose of the inverse
  V3 r;
  r.x = n.x * im[0] + n.y * im[4] + n.z * im[8];
  r.y = n.x * im[1] + n.y * im[5] + n.z * im[9];
  r.z = n.x * im[2] + n.y * im[6] + n.z * im[10];
  return r;
}
PT_DEV double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vec.normalize / Normal.normalize (geometry.py:130-136, 219-225): x*x here
PT_DEV V3 normalize3(V3 a) {
  const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  V3 r = {a.x / n, a.y / n, a.z / n};
  return r;
}
// ocml's fp64 sin/cos/atan2/acos are polynomial kernels with ~25 double constants each.  Inlined, LICM
// hoists those constants out of the pixel loops into VGPRs that stay live for the whole kernel (~50
// registers for code that runs once per bounce at most).  Behind a call they live only in the callee.
#define PT_NOINLINE static __device__ __attribute__((noinline))
PT_NOINLINE double pt_sin(double x) { return sin(x); }
PT_NOINLINE double pt_cos(double x) { return cos(x); }
// sin and cos of ONE angle share their argument reduction and both polynomial kernels in ocml (sincos): 198 instructions
// against 348 for the two calls, whose common part the compiler does not merge.  The VALUES are those of sin() and cos() --
// op 11 of pt_probe_kernel compares them over every angle scatter_ray can form (tests/test_gpu_probes.py).
PT_NOINLINE void pt_sincos(double x, double *s, double *c) { sincos(x, s, c); }
PT_NOINLINE double pt_atan2(double y, double x) { return atan2(y, x); }
PT_NOINLINE double pt_acos(double x) { return acos(x); }

// ---- one lane's value for the whole wave (v_readlane; `lane` wave-uniform) ----
PT_DEV double rl_f64(double v, int lane) {  // v_readlane of a double (lane wave-uniform)
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
PT_DEV unsigned long long rl_u64(unsigned long long u, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
PT_DEV V3 rl_v3(V3 v, int lane) {
  V3 r = {rl_f64(v.x, lane), rl_f64(v.y, lane), rl_f64(v.z, lane)};
  return r;
}

PT_DEV double max2(double a, double b) { return (b > a) ? b : a; }  // Python max(a, b)
