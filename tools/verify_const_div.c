// tools/verify_const_div.c -- exhaustive CPU proof (all 2^32 fp32 inputs) that the 3-instruction
// constant-division sequence used by the exact collision kernel equals the IEEE quotient.
//   gcc -O2 -march=native -ffp-contract=off -fopenmp tools/verify_const_div.c -lm -o /tmp/v && /tmp/v
// Output recorded in profiles/r01_const_div_proof.txt.
// exhaustive check: for every finite float x, does the 3-op (mul, fma, fma) or 5-op sequence with
// R = RN(1/C) reproduce RN(x / C) bit for bit?
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline float bits(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static inline uint32_t ubits(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
int main(void){
  const float c_sq = 1.f/3.f;
  const float Cs[3] = { c_sq, 2.f*c_sq, 2.f*c_sq*c_sq };
  for (int ci=0; ci<3; ci++){
    const float C = Cs[ci]; const float R = 1.0f / C;
    long bad3=0, bad5=0, bad3_norm=0; uint32_t first3=0;
    #pragma omp parallel for reduction(+:bad3,bad5,bad3_norm) schedule(static)
    for (int64_t i=0;i<(1LL<<32);i++){
      uint32_t u=(uint32_t)i; float x=bits(u);
      if (!isfinite(x)) continue;
      float want = x / C;
      float q0 = x*R; float r0 = fmaf(-C,q0,x); float q1 = fmaf(r0,R,q0);
      float r1 = fmaf(-C,q1,x); float q2 = fmaf(r1,R,q1);
      if (ubits(q1)!=ubits(want)) { bad3++; float ax=fabsf(x); if (ax>1e-30f && ax<1e30f) bad3_norm++; }
      if (ubits(q2)!=ubits(want)) bad5++;
    }
    /* tiny dividends: the fast quotient must stay finite and far below 2^-25 (it is then absorbed
       by the additions to 1.f exactly like the correct quotient) */
    long tiny_bad = 0;
    #pragma omp parallel for reduction(+:tiny_bad) schedule(static)
    for (int64_t i=0;i<(1LL<<32);i++){
      uint32_t u=(uint32_t)i; float x=bits(u);
      if (!isfinite(x) || fabsf(x) > 1e-30f) continue;
      float q0 = x*R; float r0 = fmaf(-C,q0,x); float q1 = fmaf(r0,R,q0);
      if (!isfinite(q1) || fabsf(q1) > 0x1p-90f) tiny_bad++;
    }
    printf("C=%.9g: dividends with |x| <= 1e-30 whose fast quotient is not finite and below 2^-90: %ld\n", C, tiny_bad);
    printf("C=%.9g (0x%08x) R=%.9g: 3-op mismatches %ld (in 1e-30<|x|<1e30: %ld), 5-op mismatches %ld\n", C, ubits(C), R, bad3, bad3_norm, bad5);
  }
  return 0;
}
