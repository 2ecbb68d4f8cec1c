/* Proof by exhaustion for csrc/pt_math.h: pcg_float.
 *
 * The reference computes random_float() = random() / 0xFFFFFFFF (pcg.py:60-62): an fp64 DIVISION of a 32-bit integer by the
 * constant 4294967295.0.  On the device a division is a dozen dependent instructions (v_div_scale x 2, v_rcp_f64, five
 * fused multiply-adds, v_div_fmas, v_div_fixup) in chains whose latency is the kernels' time; the kernels compute
 *     q0 = x * r;  e = fma(-c, q0, x);  q = fma(e, r, q0)        with r = RN(1 / c) = 0x1.00000001p-32
 * instead (one multiplication, two fused multiply-adds: the classical correction step of a division by a known reciprocal).
 * This program checks ALL 2^32 possible inputs: q equals the IEEE quotient bit for bit, every time (the plain product q0
 * alone is off by an ulp for 5 767 168 of them).  tests/test_pcg_float_division.py builds and runs it (~5 s on 8 cores). */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <omp.h>
int main(void) {
  const double c = 4294967295.0;
  const double r = 1.0 / c;  // correctly rounded reciprocal (IEEE division)
  long long bad = 0, bad0 = 0;
  #pragma omp parallel for reduction(+:bad,bad0) schedule(static)
  for (long long i = 0; i < (1LL << 32); ++i) {
    const double x = (double)(uint32_t)i;
    const double want = x / c;
    const double q0 = x * r;
    const double e = fma(-c, q0, x);
    const double q1 = fma(e, r, q0);
    if (q1 != want) ++bad;
    if (q0 != want) ++bad0;
  }
  printf("r = %a; mismatches of the corrected quotient: %lld; of the plain product: %lld\n", r, bad, bad0);
  return bad != 0;
}
