// lbm_kernels.hip.h -- CDNA4 (gfx950) device code of the D2Q9-BGK engine.
//
// One fused kernel replaces the reference's five sweeps per timestep
// (accelerate_flow, propagate, rebound, collision, av_velocity;
// /root/reference/SerialCode/d2q9-bgk.c:207-458):
//   pull-stream 9 populations from the neighbours (periodic in x; in y either periodic or fed by
//   halo rows stored around the slab), bounce back on blocked cells, BGK-relax fluid cells, sum
//   |u| of the relaxed cells into one partial per workgroup, and apply the NEXT step's
//   accelerate_flow to the lid row before storing (so no separate pass over that row is needed).
// step_vec4 / step_scalar advance one timestep per pass over memory, step2_stream two, stepk_stream / stepk_pk
// two to four (stepk_pk with the collision on pairs of cells: packed fp32 instructions), step_tile several from LDS.
//
// Layout: structure of arrays interleaved by row -- value (k, y, x) lives at
// base + y*row_pitch + k*plane_stride + x with plane_stride = pitch and row_pitch = 9*pitch, i.e.
// the 9 planes of one row lie next to each other (36*nx bytes), rows follow each other, and four
// halo rows sit below row 0 and above row rows-1.  Every access is a contiguous, 16-byte-aligned
// run along x of ONE speed (fully coalesced), and a workgroup's 9+9 streams fall into a ~1 MB
// window instead of 18 windows 256 MiB apart: measured 10-13 % faster than 9 whole-grid planes on
// MI355X (profiles/r01_tuning.md).
//
// Each value is read exactly once and written exactly once per step: the algorithmic traffic is
// 72 B per lattice update, and there is no reuse to stage in LDS or to feed MFMA.  The one-step
// kernel is bound by HBM bandwidth: each lane owns 4 consecutive cells and moves every plane with
// one 16-byte access; the +-1 column shifts of the six x-moving populations are assembled from
// the lane's own aligned vector plus one neighbour dword.
//
// Speed numbering (SerialCode/d2q9-bgk.c:9-15):   6 2 5
//                                                  3 0 1
//                                                  7 4 8
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace lbm {

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kQ = 9;
constexpr int kNoRow = -1000000;  // "no such row in this slab"

// 1/3 rounded to fp32, and the two constant divisors of the equilibrium, folded in fp32
// exactly as the reference's "2.f * c_sq" and "2.f * c_sq * c_sq" (SerialCode/d2q9-bgk.c:308,367-370)
constexpr float kCsq = 1.f / 3.f;
constexpr float kTwoCsq = 2.f * kCsq;
constexpr float kTwoCsqSq = 2.f * kCsq * kCsq;
constexpr float kW0 = 4.f / 9.f;
constexpr float kW1 = 1.f / 9.f;
constexpr float kW2 = 1.f / 36.f;

struct StepArgs {
  const float* src;           // plane 0 of the source lattice
  float* dst;                 // plane 0 of the destination lattice
  const unsigned char* mask;  // rows x pitch, 1 = blocked
  long plane_stride;          // floats between planes
  int pitch;                  // bytes between rows of the uint8 mask
  long row_pitch;             // floats between lattice rows of one plane
  int nx;                     // cells per row
  int rows;                   // rows owned by this slab
  int row_first;              // first slab row this launch advances
  int row_stride;             // distance between the rows this launch advances (1 = contiguous)
  int n_rows;                 // number of rows this launch advances
  int accel_row;              // slab row that receives next step's acceleration, or kNoRow
  float omega;
  float a1, a2;               // density*accel/9, density*accel/36 (SerialCode/d2q9-bgk.c:219-220)
  float* partials;            // one fp32 partial sum of |u| per workgroup of this launch
  int reverse;                // 1: workgroup b handles tile (n_tiles-1-b): rows are swept top-down
  int wrap;                   // 1: the slab is the whole grid, rows wrap periodically;
                              // 0: rows -1 and `rows` are halo rows stored around the slab
};

// ---------------------------------------------------------------------------------------------
// per-cell arithmetic
// ---------------------------------------------------------------------------------------------

// EXACT: the reference's expression trees, IEEE division and sqrt, no contraction
// (the library is compiled with -ffp-contract=off).  SerialCode/d2q9-bgk.c:325-401, 426-450.
__device__ __forceinline__ void moments_exact(const float (&f)[kQ], float& rho, float& ux, float& uy) {
  float d = f[0];
#pragma unroll
  for (int k = 1; k < kQ; k++) d += f[k];
  rho = d;
  ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / d;
  uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / d;
}

// ---- exact division by the three constant divisors ------------------------------------------
// x / C with C constant is the bulk of the reference's 27 divides per cell.  With R = RN(1/C):
//     q = x*R;  r = fma(-C, q, x);  q' = fma(r, R, q)
// q' equals the correctly rounded x / C -- verified EXHAUSTIVELY on the CPU for every fp32 x with
// 1e-30 < |x| < 1e30 and each of the three constants (tools/verify_const_div.c; the only
// mismatches lie where the residual r underflows or x*R overflows); x = 0 gives 0.
// Below 1e-30 the fast quotient may be a few ulps off, but it cannot change the result: every
// quotient is only ever ADDED to a value that is exactly 1 at that point --
//   eq_k = w*rho * (((1 + u_k/c^2) + u_k^2/(2c^4)) - |u|^2/(2c^2)):
//   |u_k^2| < 1e-30 means |u_k| < 1e-15, so |u_k/c^2| < 2^-25 and 1 + u_k/c^2 == 1, then
//   1 + (anything below 2^-25) == 1;  |u|^2 < 1e-30 means every |u_k| < 2e-15, so the minuend is 1
//   and 1 - (anything below 2^-25) == 1
// -- and a wrong-by-ulps number below 5e-30 is absorbed exactly like the right one (also checked
// exhaustively: the fast quotient of every |x| <= 1e-30 is finite and below 2^-90).
// Only huge dividends need care: a cell whose |u|^2 is not below 5e28 (or is NaN) takes the IEEE
// divides.  So the result is bit-identical to the reference's "x / c_sq" for ALL inputs at
// 3 instructions per divide instead of ~11 plus a quarter-rate v_rcp_f32.
__device__ __forceinline__ float div_const_fast(float x, float C, float R) {
  const float q = x * R;
  const float r = __fmaf_rn(-C, q, x);
  return __fmaf_rn(r, R, q);
}

struct EqTerms {
  float q1;  // u / c_sq
  float q2;  // (u*u) / (2 c_sq^2)
};
template <bool GUARDED_FAST>
__device__ __forceinline__ EqTerms eq_terms(float u) {
  constexpr float kInvCsq = 1.0f / kCsq;            // RN(1/C), folded by the compiler in fp32
  constexpr float kInvTwoCsqSq = 1.0f / kTwoCsqSq;
  EqTerms e;
  const float sq = u * u;
  if constexpr (GUARDED_FAST) {
    e.q1 = div_const_fast(u, kCsq, kInvCsq);
    e.q2 = div_const_fast(sq, kTwoCsqSq, kInvTwoCsqSq);
  } else {
    e.q1 = u / kCsq;
    e.q2 = sq / kTwoCsqSq;
  }
  return e;
}

__device__ __forceinline__ float equilibrium_from_terms(float w_rho, float q1, float q2, float usq_term) {
  return w_rho * (1.f + q1 + q2 - usq_term);
}

template <bool GUARDED_FAST>
__device__ __forceinline__ void collide_exact_body(const float (&t)[kQ], float omega, float rho, float ux,
                                                   float uy, float (&r)[kQ]) {
  constexpr float kInvTwoCsq = 1.0f / kTwoCsq;
  const float u_sq = ux * ux + uy * uy;
  const float usq_term = GUARDED_FAST ? div_const_fast(u_sq, kTwoCsq, kInvTwoCsq) : u_sq / kTwoCsq;
  const float w1r = kW1 * rho, w2r = kW2 * rho;
  // u[3] = -u[1], u[4] = -u[2], u[7] = -u[5], u[8] = -u[6] (exact negations), so the quotients of
  // the four opposite directions are the negated / identical quotients of the first four
  const EqTerms ex = eq_terms<GUARDED_FAST>(ux);
  const EqTerms ey = eq_terms<GUARDED_FAST>(uy);
  const EqTerms es = eq_terms<GUARDED_FAST>(ux + uy);
  const EqTerms ed = eq_terms<GUARDED_FAST>(-ux + uy);
  float eq[kQ];
  eq[0] = kW0 * rho * (1.f - usq_term);
  eq[1] = equilibrium_from_terms(w1r, ex.q1, ex.q2, usq_term);
  eq[2] = equilibrium_from_terms(w1r, ey.q1, ey.q2, usq_term);
  eq[3] = equilibrium_from_terms(w1r, -ex.q1, ex.q2, usq_term);
  eq[4] = equilibrium_from_terms(w1r, -ey.q1, ey.q2, usq_term);
  eq[5] = equilibrium_from_terms(w2r, es.q1, es.q2, usq_term);
  eq[6] = equilibrium_from_terms(w2r, ed.q1, ed.q2, usq_term);
  eq[7] = equilibrium_from_terms(w2r, -es.q1, es.q2, usq_term);
  eq[8] = equilibrium_from_terms(w2r, -ed.q1, ed.q2, usq_term);
#pragma unroll
  for (int k = 0; k < kQ; k++) r[k] = t[k] + omega * (eq[k] - t[k]);
}

// want_speed (wave-uniform): also return |u| of the relaxed cell for the av_velocity sum; the warm-up rows
// of a two-step band relax cells whose |u| belongs to another band's sum
template <bool EXACT>
__device__ __forceinline__ void collide(const float (&t)[kQ], float omega, float (&r)[kQ], float& speed,
                                        bool want_speed = true);

// ---- the two divides by the density, sharing one reciprocal -----------------------------------
// hipcc expands the IEEE fp32 divide a / b (AMDGPU LowerFDIV32) into
//     r0 = v_rcp_f32(b);  e = fma(-b, r0, 1);  r = fma(e, r0, r0);            (refined reciprocal)
//     q = a*r;  e = fma(-b, q, a);  q = fma(e, r, q);  e = fma(-b, q, a);  q = fma(e, r, q)
// wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act when an operand or the
// quotient is near the ends of the exponent range (denominator denormal or above 2^126,
// |a| below 2^-103, |a/b| above 2^96 or denormal) or is 0 / inf / NaN.  u_x and u_y divide by the
// same density, so the first line is done once and the second per numerator: the very same
// instructions, hence the same bits, 28 instead of 48 issue slots per cell.  Outside the plain
// range: a density outside [2^-60, 2^60] sends the cell to the IEEE divides; a numerator of 0 gives 0
// either way; a non-zero numerator below 2^-103 gives a quotient below 2^-43 that may differ in
// its last bits -- before the collision such a u is absorbed like the tiny quotients above
// (|u/c^2| < 2^-25), after it it only enters the diagnostic sum of |u| at the 1e-13 level; a
// quotient above 2^96 trips the |u|^2 guard.
__device__ __forceinline__ bool density_in_plain_range(float rho) {
  return (__float_as_uint(rho) - 0x21800000u) < (0x5D800000u - 0x21800000u);  // 2^-60 <= rho < 2^60
}
__device__ __forceinline__ float refined_rcp(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e = __fmaf_rn(-b, r0, 1.f);
  return __fmaf_rn(e, r0, r0);
}
__device__ __forceinline__ float div_with_rcp(float a, float b, float r) {
  float q = a * r;
  float e = __fmaf_rn(-b, q, a);
  q = __fmaf_rn(e, r, q);
  e = __fmaf_rn(-b, q, a);
  return __fmaf_rn(e, r, q);
}
// density and velocity with the bits of moments_exact, the two divides sharing one refined reciprocal
__device__ __forceinline__ void moments_shared(const float (&f)[kQ], float& rho, float& ux, float& uy) {
  float d = f[0];
#pragma unroll
  for (int k = 1; k < kQ; k++) d += f[k];
  rho = d;
  const float X = f[1] + f[5] + f[8] - (f[3] + f[6] + f[7]);
  const float Y = f[2] + f[5] + f[6] - (f[4] + f[7] + f[8]);
  if (density_in_plain_range(d)) {
    const float r = refined_rcp(d);
    ux = div_with_rcp(X, d, r);
    uy = div_with_rcp(Y, d, r);
  } else {
    ux = X / d;
    uy = Y / d;
  }
}

// SPEED: where |u| for the av_velocity sum comes from
//   0  the relaxed populations, IEEE sqrt -- the reference's own arithmetic (SerialCode/d2q9-bgk.c:426-450):
//      per-cell |u| bit-identical to the reference's
//   1  the pre-collision moments (BGK conserves density and momentum, so they equal the relaxed cell's up to
//      rounding, ~1e-7 relative), native v_sqrt_f32 (1 ulp): saves the second moment pass, ~50 VALU
//      instructions per cell.  The lattice is untouched by this choice; av_vels moves by < 1e-6 relative,
//      far inside the summation-order noise it already carries (the reference adds sequentially in fp32).
#ifndef LBM_STREAM_SPEED
#define LBM_STREAM_SPEED 1
#endif
template <int SPEED = 0>
__device__ __forceinline__ void collide_exact(const float (&t)[kQ], float omega, float (&r)[kQ],
                                              float& speed, bool want_speed) {
  float rho, ux, uy;
  moments_shared(t, rho, ux, uy);
  // |u|^2 below 5e28 bounds every dividend of the fast constant divides (squares of ux, uy,
  // ux+-uy are at most 2|u|^2 < 1e29); NaN compares false and takes the IEEE path
  const float u_sq = ux * ux + uy * uy;
  const bool ok = u_sq < 5.0e28f;
  if (ok) collide_exact_body<true>(t, omega, rho, ux, uy, r);
  else    collide_exact_body<false>(t, omega, rho, ux, uy, r);
  speed = 0.f;
  if (want_speed) {
    if constexpr (SPEED == 0) {
      // av_velocity() looks at the relaxed populations (SerialCode/d2q9-bgk.c:169, 426-450)
      float rho2, ux2, uy2;
      moments_shared(r, rho2, ux2, uy2);
      speed = sqrtf((ux2 * ux2) + (uy2 * uy2));  // IEEE: __fsqrt_rn is the native approximation
    } else {
      speed = __builtin_amdgcn_sqrtf(u_sq);
    }
  }
}
template <>
__device__ __forceinline__ void collide<true>(const float (&t)[kQ], float omega, float (&r)[kQ],
                                              float& speed, bool want_speed) {
  collide_exact<0>(t, omega, r, speed, want_speed);
}

// FAST: one reciprocal, multiplies by 3, 4.5, 1.5 and explicit FMAs.  BGK conserves density and
// momentum, so |u| of the relaxed cell is taken from the pre-collision moments.
template <>
__device__ __forceinline__ void collide<false>(const float (&t)[kQ], float omega, float (&r)[kQ],
                                               float& speed, bool /*want_speed*/) {
  const float rho = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])) + t[8];
  const float inv = 1.f / rho;
  const float ux = ((t[1] + t[5] + t[8]) - (t[3] + t[6] + t[7])) * inv;
  const float uy = ((t[2] + t[5] + t[6]) - (t[4] + t[7] + t[8])) * inv;
  const float u_sq = __fmaf_rn(ux, ux, uy * uy);
  const float base = __fmaf_rn(-1.5f, u_sq, 1.f);
  const float w0r = omega * kW0 * rho, w1r = omega * kW1 * rho, w2r = omega * kW2 * rho;
  const float keep = 1.f - omega;
  auto relax = [&](float tk, float wr, float u) {
    const float poly = __fmaf_rn(u, __fmaf_rn(4.5f, u, 3.f), base);
    return __fmaf_rn(wr, poly, keep * tk);
  };
  r[0] = __fmaf_rn(w0r, base, keep * t[0]);
  r[1] = relax(t[1], w1r, ux);
  r[2] = relax(t[2], w1r, uy);
  r[3] = relax(t[3], w1r, -ux);
  r[4] = relax(t[4], w1r, -uy);
  r[5] = relax(t[5], w2r, ux + uy);
  r[6] = relax(t[6], w2r, uy - ux);
  r[7] = relax(t[7], w2r, -ux - uy);
  r[8] = relax(t[8], w2r, ux - uy);
  speed = __builtin_amdgcn_sqrtf(u_sq);  // v_sqrt_f32, 1 ulp
}

// accelerate_flow() on one cell (SerialCode/d2q9-bgk.c:229-242)
__device__ __forceinline__ void accelerate(float (&f)[kQ], float a1, float a2) {
  if ((f[3] - a1) > 0.f && (f[6] - a2) > 0.f && (f[7] - a2) > 0.f) {
    f[1] += a1;  f[5] += a2;  f[8] += a2;
    f[3] -= a1;  f[6] -= a2;  f[7] -= a2;
  }
}

// rebound(): mirrored copy, speed 0 kept (SerialCode/d2q9-bgk.c:291-298)
__device__ __forceinline__ void bounce(const float (&t)[kQ], float (&r)[kQ]) {
  r[0] = t[0];
  r[1] = t[3];  r[2] = t[4];  r[3] = t[1];  r[4] = t[2];
  r[5] = t[7];  r[6] = t[8];  r[7] = t[5];  r[8] = t[6];
}

// ---------------------------------------------------------------------------------------------
// workgroup reduction: wave64 shuffles, then one LDS hop across the 4 waves
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int BLOCK = kBlock>
__device__ __forceinline__ float block_sum(float v) {
  __shared__ float wave_part[BLOCK / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_part[wave] = v;
  __syncthreads();
  float total = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) total += wave_part[w];
  }
  return total;  // valid in thread 0
}

// ---------------------------------------------------------------------------------------------
// fused step, 4 cells per lane (nx % 4 == 0, pitch % 4 == 0)
//
// MATH : 0 exact (reference arithmetic), 1 fast (reciprocal + FMA), 2 copy-through (tuning only)
// NEIGH: how the +-1 column neighbours of the six x-moving populations are obtained
//        0 = one strided dword load per plane (simple, but each touches as many cache lines as
//            the 16-byte load beside it),
//        1 = from the adjacent lane's aligned vector by a wave64 cross-lane move
//            (__shfl_up/down -> ds_bpermute_b32, no LDS storage); only the first / last lane of a
//            wave and the row ends load a dword,
//        2 = the same with DPP wave_shr:1 / wave_shl:1 (one VALU move, no LDS crossbar).
// NTS  : nontemporal stores.
// ---------------------------------------------------------------------------------------------
template <int NEIGH>
__device__ __forceinline__ float lane_from_west(float v) {  // lane i <- lane i-1
  if constexpr (NEIGH == 2)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
  else
    return __shfl_up(v, 1, 64);
}
template <int NEIGH>
__device__ __forceinline__ float lane_from_east(float v) {  // lane i <- lane i+1
  if constexpr (NEIGH == 2)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
  else
    return __shfl_down(v, 1, 64);
}

template <bool NTS>
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  if constexpr (NTS) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
}

template <int MATH, int NEIGH, bool NTS, int BLOCK = kBlock, bool SYNC = false>
__global__ __launch_bounds__(BLOCK) void step_vec4(const StepArgs a) {
  const int quads_x = a.nx >> 2;
  const long n_quads = (long)quads_x * a.n_rows;
  // alternate the sweep direction from step to step: the rows written last by the previous step
  // (still in L2 / Infinity Cache) are then the first ones read
  const long tile = a.reverse ? (long)(gridDim.x - 1 - blockIdx.x) : (long)blockIdx.x;
  const long q = tile * BLOCK + threadIdx.x;
  const bool active = q < n_quads;
  const long ps = a.plane_stride;
  float my_sum = 0.f;
  float r[4][kQ];
  int row = 0, x0 = 0;

  if (active) {
    const int rsel = (int)(q / quads_x);
    x0 = (int)(q - (long)rsel * quads_x) << 2;
    row = a.row_first + rsel * a.row_stride;

    // neighbour columns with periodic wrap (SerialCode/d2q9-bgk.c:258,260)
    const int xw = (x0 == 0) ? a.nx - 1 : x0 - 1;
    const int xe = (x0 + 4 == a.nx) ? 0 : x0 + 4;

    const float* c_row = a.src + (long)row * a.row_pitch;  // plane 0, this row
    // row below (speeds 2,5,6 arrive from it) and above (4,7,8), :257,259.  In a multi-slab run the
    // slab is surrounded by halo rows (-1 and `rows`) that hold the neighbours' boundary rows.
    const int rs = (row == 0 && a.wrap) ? a.rows - 1 : row - 1;
    const int rn = (row == a.rows - 1 && a.wrap) ? 0 : row + 1;
    const float* sb = a.src + (long)rs * a.row_pitch;
    const float* nb = a.src + (long)rn * a.row_pitch;
    const float *s2 = sb + 2 * ps, *s5 = sb + 5 * ps, *s6 = sb + 6 * ps;
    const float *n4 = nb + 4 * ps, *n7 = nb + 7 * ps, *n8 = nb + 8 * ps;

    // 9 aligned 16-byte loads
    const float4 v0 = *reinterpret_cast<const float4*>(c_row + x0);
    const float4 v1 = *reinterpret_cast<const float4*>(c_row + 1 * ps + x0);
    const float4 v3 = *reinterpret_cast<const float4*>(c_row + 3 * ps + x0);
    const float4 v2 = *reinterpret_cast<const float4*>(s2 + x0);
    const float4 v5 = *reinterpret_cast<const float4*>(s5 + x0);
    const float4 v6 = *reinterpret_cast<const float4*>(s6 + x0);
    const float4 v4 = *reinterpret_cast<const float4*>(n4 + x0);
    const float4 v7 = *reinterpret_cast<const float4*>(n7 + x0);
    const float4 v8 = *reinterpret_cast<const float4*>(n8 + x0);
    // the cell just west of the quad (for speeds 1,5,8) and just east of it (3,6,7)
    float e1, e3, e5, e6, e7, e8;
    if constexpr (NEIGH == 0) {
      e1 = c_row[1 * ps + xw];  // speed 1 travels east: comes from the west cell
      e3 = c_row[3 * ps + xe];
      e5 = s5[xw];
      e6 = s6[xe];
      e7 = n7[xe];
      e8 = n8[xw];
    } else {
      const int lane = threadIdx.x & 63;
      e1 = lane_from_west<NEIGH>(v1.w);
      e5 = lane_from_west<NEIGH>(v5.w);
      e8 = lane_from_west<NEIGH>(v8.w);
      e3 = lane_from_east<NEIGH>(v3.x);
      e6 = lane_from_east<NEIGH>(v6.x);
      e7 = lane_from_east<NEIGH>(v7.x);
      // the wave's first / last lane and the row ends have no such lane: one dword each
      if (lane == 0 || x0 == 0) {
        e1 = c_row[1 * ps + xw];  e5 = s5[xw];  e8 = n8[xw];
      }
      if (lane == 63 || x0 + 4 == a.nx || q + 1 == n_quads) {
        e3 = c_row[3 * ps + xe];  e6 = s6[xe];  e7 = n7[xe];
      }
    }
    const uchar4 m = *reinterpret_cast<const uchar4*>(a.mask + (long)row * a.pitch + x0);

    // streamed populations of the 4 cells: t[j][k]
    float t[4][kQ] = {
        {v0.x, e1,   v2.x, v3.y, v4.x, e5,   v6.y, v7.y, e8},
        {v0.y, v1.x, v2.y, v3.z, v4.y, v5.x, v6.z, v7.z, v8.x},
        {v0.z, v1.y, v2.z, v3.w, v4.z, v5.y, v6.w, v7.w, v8.y},
        {v0.w, v1.z, v2.w, e3,   v4.w, v5.z, e6,   e7,   v8.z}};
    const unsigned char blocked[4] = {m.x, m.y, m.z, m.w};
    const bool lid = (row == a.accel_row);

#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (blocked[j]) {
        bounce(t[j], r[j]);
      } else {
        float speed = 0.f;
        if constexpr (MATH == 0) collide<true>(t[j], a.omega, r[j], speed);
        else if constexpr (MATH == 1) collide<false>(t[j], a.omega, r[j], speed);
        else {
#pragma unroll
          for (int k = 0; k < kQ; k++) r[j][k] = t[j][k];
        }
        my_sum += speed;
        if (lid) accelerate(r[j], a.a1, a.a2);
      }
    }
  }

  // optionally let the whole workgroup reach its stores together (one contiguous burst per plane)
  if constexpr (SYNC) __syncthreads();

  if (active) {
    float* d_row = a.dst + (long)row * a.row_pitch + x0;
#pragma unroll
    for (int k = 0; k < kQ; k++) store4<NTS>(d_row + k * ps, r[0][k], r[1][k], r[2][k], r[3][k]);

  }

  const float total = block_sum<BLOCK>(my_sum);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// TWO timesteps per pass over memory (temporal blocking); periodic in x, in y either periodic (single
// slab) or fed by the two halo rows stored below and above a slab (wrap = 0).
//
// The one-step kernel above sits at the float4-copy ceiling of the chip (72 B per update cannot
// move faster).  The only way past that roofline is to move fewer bytes: this kernel advances the
// lattice by two steps while reading it once and writing it once (36 B per update).
//
// One WAVE (64 lanes, no LDS, no barrier) owns a vertical strip of 62 quads (248 cells; lanes 0
// and 63 are halo quads that only feed their neighbours) and sweeps a band of rows upwards.
// Per row iteration it
//   1. pulls row r of the source lattice and relaxes it            -> step t   results N (registers)
//   2. relaxes row r-1 a second time from a sliding window of step-t results kept in registers:
//        speeds 2,5,6 of row r-2, speeds 0,1,3 of row r-1, speeds 4,7,8 of row r (= N);
//      the +-1 column neighbours come from the adjacent lanes (DPP wave_shr:1 / wave_shl:1)
//                                                                   -> step t+1 results, stored
//   3. rotates the window.
// Every step-t value is consumed exactly once, so the window is 36 floats per lane and nothing
// is staged in LDS.  Redundant work: 2 of 64 lanes and 2 warm-up rows per band.
// Per-cell arithmetic is the same code as the one-step kernel: results stay bit-identical.
//
// accelerate_flow: step t+1's acceleration is applied to the step-t results of the lid row
// (always); step t+2's to the stored results when accel_after != 0 (not on the last step of a run).
// ---------------------------------------------------------------------------------------------
struct Step2Args {
  const float* src;
  float* dst;
  const unsigned char* mask;
  long plane_stride;
  long row_pitch;
  int pitch;
  int nx;
  int rows;        // rows owned by the slab
  int wrap;        // 1: rows wrap periodically (single slab); 0: two halo rows surround the slab
  int band_rows;   // output rows per wave (band height)
  int row_first;   // band b of this launch starts at row_first + b*band_pitch ...
  int band_pitch;  // (= band_rows for a contiguous region)
  int row_end;     // ... and ends before row_end
  int n_strips;    // waves across x
  int accel_row;
  int accel_after;
  float omega, a1, a2;
  float* partials1;  // per wave: sum |u| after step t   (own cells only)
  float* partials2;  // per wave: sum |u| after step t+1
};

constexpr int kStripQuads = 62;  // output lanes per wave (lanes 1..62; each owns C cells)

template <int MATH, int SPEED = 0>
__device__ __forceinline__ void relax_cell(const float (&t)[kQ], bool blocked, bool lid, float omega, float a1,
                                           float a2, float (&r)[kQ], float& speed, bool want_speed = true) {
  speed = 0.f;
  if (blocked) {
    bounce(t, r);
  } else {
    if constexpr (MATH == 0) collide_exact<SPEED>(t, omega, r, speed, want_speed);
    else collide<false>(t, omega, r, speed, want_speed);
    if (lid) accelerate(r, a1, a2);
  }
}

// the 9 aligned vectors and the mask bytes one lane pulls for one row; C cells per lane
template <int C>
struct RowPull {
  typedef float vec __attribute__((ext_vector_type(C)));
  vec v[kQ];
  unsigned m;  // C mask bytes
};

template <int C>
__device__ __forceinline__ RowPull<C> pull_row(const Step2Args& a, int r, int x0) {
  typedef typename RowPull<C>::vec vec;
  const long ps = a.plane_stride;
  const int rs = (r == 0 && a.wrap) ? a.rows - 1 : r - 1;
  const int rn = (r == a.rows - 1 && a.wrap) ? 0 : r + 1;
  const float* c_row = a.src + (long)r * a.row_pitch + x0;
  const float* sb = a.src + (long)rs * a.row_pitch + x0;
  const float* nb = a.src + (long)rn * a.row_pitch + x0;
  RowPull<C> p;
  p.v[0] = *reinterpret_cast<const vec*>(c_row);
  p.v[1] = *reinterpret_cast<const vec*>(c_row + 1 * ps);
  p.v[3] = *reinterpret_cast<const vec*>(c_row + 3 * ps);
  p.v[2] = *reinterpret_cast<const vec*>(sb + 2 * ps);
  p.v[5] = *reinterpret_cast<const vec*>(sb + 5 * ps);
  p.v[6] = *reinterpret_cast<const vec*>(sb + 6 * ps);
  p.v[4] = *reinterpret_cast<const vec*>(nb + 4 * ps);
  p.v[7] = *reinterpret_cast<const vec*>(nb + 7 * ps);
  p.v[8] = *reinterpret_cast<const vec*>(nb + 8 * ps);
  const unsigned char* mp = a.mask + (long)r * a.pitch + x0;
  if constexpr (C == 4) p.m = *reinterpret_cast<const unsigned*>(mp);
  else if constexpr (C == 2) p.m = *reinterpret_cast<const unsigned short*>(mp);
  else p.m = *mp;
  return p;
}

// periodic single slab: fold the row index; multi-slab: rows -2..rows+1 exist as halo rows
__device__ __forceinline__ int wrap_row(int r, int rows, int wrap) {
  if (wrap) {
    if (r < 0) r += rows;
    if (r >= rows) r -= rows;
  }
  return r;
}

// C = cells per lane (4: 16-byte accesses, 138 VGPRs, 3 waves/SIMD; 2: 8-byte accesses, about half
// the registers, twice the waves -- better for slabs too small to fill the chip with 4-cell lanes)
template <int MATH, bool NTS, int C>
__global__ __launch_bounds__(64) void step2_stream(const Step2Args a) {
  typedef typename RowPull<C>::vec vec;
  const int lane = threadIdx.x;
  const int strip = blockIdx.x % a.n_strips;
  const int units_x = a.nx / C;
  const int y0 = a.row_first + (int)(blockIdx.x / a.n_strips) * a.band_pitch;
  const int band_n = min(a.band_rows, a.row_end - y0);

  // this lane's group of C cells (may lie outside the grid: halo lanes and the tail of the last
  // strip wrap around)
  const int ux_raw = strip * kStripQuads + lane - 1;
  int ux = ux_raw % units_x;
  if (ux < 0) ux += units_x;
  const int x0 = ux * C;
  const bool out_lane = (lane >= 1) && (lane <= kStripQuads) && (ux_raw < units_x);
  const long ps = a.plane_stride;

  // sliding window of step-t results (registers)
  float w256[3][C];  // speeds 2,5,6 of row r-2
  float w013[3][C];  // speeds 0,1,3 of row r-1
  float n256[3][C];  // speeds 2,5,6 of row r-1 (become w256 after the rotation)
  unsigned m_prev = 0;
  float sum1 = 0.f, sum2 = 0.f;

  for (int i = 0; i < band_n + 2; i++) {
    // ---- step t on row r ------------------------------------------------------------------
    const int r = wrap_row(y0 - 1 + i, a.rows, a.wrap);
    const RowPull<C> p = pull_row<C>(a, r, x0);
    // +-1 column neighbours of the pulled vectors come from the adjacent lanes.  Lane 0 has no
    // west lane and lane 63 no east lane: their outermost cells get zeros and produce garbage,
    // which nothing consumes (those lanes are halo groups; only their inner edge feeds a neighbour).
    const float e1 = lane_from_west<2>(p.v[1][C - 1]), e5 = lane_from_west<2>(p.v[5][C - 1]),
                e8 = lane_from_west<2>(p.v[8][C - 1]);
    const float e3 = lane_from_east<2>(p.v[3][0]), e6 = lane_from_east<2>(p.v[6][0]),
                e7 = lane_from_east<2>(p.v[7][0]);

    float t[C][kQ];
#pragma unroll
    for (int j = 0; j < C; j++) {
      const int jw = (j == 0) ? 0 : j - 1, je = (j == C - 1) ? C - 1 : j + 1;
      t[j][0] = p.v[0][j];
      t[j][1] = (j == 0) ? e1 : p.v[1][jw];
      t[j][2] = p.v[2][j];
      t[j][3] = (j == C - 1) ? e3 : p.v[3][je];
      t[j][4] = p.v[4][j];
      t[j][5] = (j == 0) ? e5 : p.v[5][jw];
      t[j][6] = (j == C - 1) ? e6 : p.v[6][je];
      t[j][7] = (j == C - 1) ? e7 : p.v[7][je];
      t[j][8] = (j == 0) ? e8 : p.v[8][jw];
    }
    const bool lid = (r == a.accel_row);
    const bool own_row = (i >= 1) && (i <= band_n);  // rows y0 .. y0+band_n-1 belong to this band
    float N[C][kQ];
#pragma unroll
    for (int j = 0; j < C; j++) {
      float speed;
      relax_cell<MATH, LBM_STREAM_SPEED>(t[j], ((p.m >> (8 * j)) & 0xffu) != 0, lid, a.omega, a.a1, a.a2, N[j], speed, own_row);
      if (own_row && out_lane) sum1 += speed;
    }

    // ---- step t+1 on row r-1 (needs step-t rows r-2, r-1, r) -------------------------------
    if (i >= 2) {
      const int ro = wrap_row(r - 1, a.rows, a.wrap);
      // neighbours in x from the adjacent lanes: west cell = lane-1's last cell, east = lane+1's first
      const float w1 = lane_from_west<2>(w013[1][C - 1]);
      const float w5 = lane_from_west<2>(w256[1][C - 1]);
      const float w8 = lane_from_west<2>(N[C - 1][8]);
      const float x3 = lane_from_east<2>(w013[2][0]);
      const float x6 = lane_from_east<2>(w256[2][0]);
      const float x7 = lane_from_east<2>(N[0][7]);
      float u[C][kQ];
#pragma unroll
      for (int j = 0; j < C; j++) {
        const int jw = (j == 0) ? 0 : j - 1, je = (j == C - 1) ? C - 1 : j + 1;
        u[j][0] = w013[0][j];
        u[j][1] = (j == 0) ? w1 : w013[1][jw];
        u[j][2] = w256[0][j];
        u[j][3] = (j == C - 1) ? x3 : w013[2][je];
        u[j][4] = N[j][4];
        u[j][5] = (j == 0) ? w5 : w256[1][jw];
        u[j][6] = (j == C - 1) ? x6 : w256[2][je];
        u[j][7] = (j == C - 1) ? x7 : N[je][7];
        u[j][8] = (j == 0) ? w8 : N[jw][8];
      }
      const bool lid2 = (ro == a.accel_row) && a.accel_after;
      float R[C][kQ];
#pragma unroll
      for (int j = 0; j < C; j++) {
        float speed;
        relax_cell<MATH, LBM_STREAM_SPEED>(u[j], ((m_prev >> (8 * j)) & 0xffu) != 0, lid2, a.omega, a.a1, a.a2, R[j], speed);
        if (out_lane) sum2 += speed;
      }
      if (out_lane) {
        float* d_row = a.dst + (long)ro * a.row_pitch + x0;
#pragma unroll
        for (int k = 0; k < kQ; k++) {
          vec o;
#pragma unroll
          for (int j = 0; j < C; j++) o[j] = R[j][k];
          if constexpr (NTS) __builtin_nontemporal_store(o, reinterpret_cast<vec*>(d_row + k * ps));
          else *reinterpret_cast<vec*>(d_row + k * ps) = o;
        }
      }
    }

    // ---- rotate the window ------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < C; j++) {
      w256[0][j] = n256[0][j];  w256[1][j] = n256[1][j];  w256[2][j] = n256[2][j];
      n256[0][j] = N[j][2];     n256[1][j] = N[j][5];     n256[2][j] = N[j][6];
      w013[0][j] = N[j][0];     w013[1][j] = N[j][1];     w013[2][j] = N[j][3];
    }
    m_prev = p.m;
  }

  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  if (lane == 0) {
    a.partials1[blockIdx.x] = sum1;
    a.partials2[blockIdx.x] = sum2;
  }
}

// ---------------------------------------------------------------------------------------------
// K timesteps per pass over memory (K = 2, 3 or 4): the sliding register window of step2_stream generalised
// to a chain of K-1 windows.  The two-step kernel at 8192^2 is bound by DRAM traffic (round-2 PMC: 5.4-5.8 TB/s
// at the memory controllers whatever the band height and however cheap the arithmetic), so the only way
// on is fewer bytes per update: K = 3 reads and writes the lattice once per THREE updates (24 B per update).
//
// Iteration i of a wave pulls row r = y0 - (K-1) + i and relaxes it (stage 1); stage s+1 then relaxes row
// r - s from window s (speeds 2,5,6 of row r-s-1, speeds 0,1,3 of row r-s) and the stage-s results of row
// r-s+1 just produced (speeds 4,7,8), +-1 columns by DPP from the adjacent lanes; the last stage's row
// r - (K-1) is stored.  Every intermediate value is consumed exactly once: 36 floats per window and lane.
// Redundant work: 2 of 64 lanes (with 4 cells per lane one halo lane per side covers up to four steps: after
// stage s the s cells of a halo lane nearest the strip's edge are garbage, and stage s+1 of the first owned
// cell needs the halo lane's LAST cell of stage s) and 2(K-1) warm-up rows per band in stage 1, 2(K-2) in
// stage 2, ...  Per-cell arithmetic is relax_cell, as in every other kernel: the lattice stays bit-identical.
//
// PREFETCH: the next row's nine vectors are requested before the current row is relaxed (36 more VGPRs; with
// two resident waves per SIMD the loads of one wave would otherwise only overlap the other wave's arithmetic
// when the two happen to be out of phase).
// XCD: workgroups are handed to the 8 XCDs round-robin; with chunk > 0, consecutive workgroups OF ONE XCD take
// horizontally adjacent strips of one band (a chunk), so the 128-byte lines that straddle two strips
// (a strip is 62 x 16 B = 992 B wide) are fetched from the fabric once per chunk instead of once per strip.
// ---------------------------------------------------------------------------------------------
struct StepKArgs {
  const float* src;
  float* dst;
  const unsigned char* mask;
  long plane_stride;
  long row_pitch;
  int pitch;
  int nx;
  int rows;        // rows owned by the slab
  int wrap;        // 1: rows wrap periodically (single slab); 0: K halo rows surround the slab
  int band_rows;   // output rows per wave (band height)
  int row_first;   // band b of this launch starts at row_first + b*band_pitch ...
  int band_pitch;  // (= band_rows for a contiguous region)
  int row_end;     // ... and ends before row_end
  int n_strips;    // waves across x
  int n_bands;     // bands of this launch
  int chunk;       // strips per XCD chunk (0: plain order, strip fastest)
  int halo_lanes;  // lanes at each end of a wave that only feed their neighbours: ceil(K / cells per lane) of the context
  int accel_row;
  int accel_row2;  // a second periodic image of the lid row among the halo rows, or kNoRow
  int accel_after;
  float omega, a1, a2;
  float* partials;      // partials[s * slot_stride + wave] = sum |u| after step t+s (own cells only)
  long slot_stride;
};

template <int C>
struct Window {
  float w256[3][C];  // speeds 2,5,6 of the row two below the newest
  float w013[3][C];  // speeds 0,1,3 of the row below the newest
  float n256[3][C];  // speeds 2,5,6 of the row below the newest (become w256 after the rotation)
  unsigned m;        // mask bytes of the row below the newest
};

template <int MATH, bool NTS, int C, int K, bool PREFETCH>
__global__ __launch_bounds__(64, (K >= 3 || PREFETCH || MATH == 0) ? 2 : 3) void stepk_stream(const StepKArgs a) {
  static_assert(K >= 2 && K <= 4 && (C == 4 || K == 2), "one halo lane per side covers K <= C steps");
  typedef typename RowPull<C>::vec vec;
  const int lane = threadIdx.x;
  int strip, band;
  if (a.chunk > 0) {
    // workgroup w runs on XCD w % 8 as that XCD's (w / 8)-th workgroup; chunk g = (strips [h*chunk, (h+1)*chunk) of band b)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunks_per_band = (a.n_strips + a.chunk - 1) / a.chunk;
    const int g = (j / a.chunk) * 8 + xcd;
    band = g / chunks_per_band;
    strip = (g - band * chunks_per_band) * a.chunk + j % a.chunk;
    if (band >= a.n_bands || strip >= a.n_strips) return;
  } else {
    strip = blockIdx.x % a.n_strips;
    band = blockIdx.x / a.n_strips;
  }
  const int wave_id = band * a.n_strips + strip;
  const int units_x = a.nx / C;
  const int y0 = a.row_first + band * a.band_pitch;
  const int band_n = min(a.band_rows, a.row_end - y0);

  const int H = a.halo_lanes;  // H * C >= K cells of halo per side
  const int ux_raw = strip * (64 - 2 * H) + lane - H;
  int ux = ux_raw % units_x;
  if (ux < 0) ux += units_x;
  const int x0 = ux * C;
  const bool out_lane = (lane >= H) && (lane < 64 - H) && (ux_raw < units_x);
  const long ps = a.plane_stride;

  Step2Args pa;  // pull_row's view of the arguments
  pa.src = a.src;  pa.mask = a.mask;  pa.plane_stride = a.plane_stride;  pa.row_pitch = a.row_pitch;
  pa.pitch = a.pitch;  pa.rows = a.rows;  pa.wrap = a.wrap;

  Window<C> win[K - 1];
#pragma unroll
  for (int s = 0; s < K - 1; s++) {
    win[s].m = 0;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int j = 0; j < C; j++) win[s].w256[q][j] = win[s].w013[q][j] = win[s].n256[q][j] = 0.f;
  }
  float sum[K];
#pragma unroll
  for (int s = 0; s < K; s++) sum[s] = 0.f;

  const int n_iter = band_n + 2 * (K - 1);
  RowPull<C> nextp;
  if constexpr (PREFETCH) nextp = pull_row<C>(pa, wrap_row(y0 - (K - 1), a.rows, a.wrap), x0);

  for (int i = 0; i < n_iter; i++) {
    // ---- stage 1: step t on row r, pulled from memory -------------------------------------------
    const int r = wrap_row(y0 - (K - 1) + i, a.rows, a.wrap);
    RowPull<C> p;
    if constexpr (PREFETCH) {
      p = nextp;
      if (i + 1 < n_iter) nextp = pull_row<C>(pa, wrap_row(y0 - (K - 1) + i + 1, a.rows, a.wrap), x0);
    } else {
      p = pull_row<C>(pa, r, x0);
    }
    float cur[C][kQ];
    {
      const float e1 = lane_from_west<2>(p.v[1][C - 1]), e5 = lane_from_west<2>(p.v[5][C - 1]),
                  e8 = lane_from_west<2>(p.v[8][C - 1]);
      const float e3 = lane_from_east<2>(p.v[3][0]), e6 = lane_from_east<2>(p.v[6][0]),
                  e7 = lane_from_east<2>(p.v[7][0]);
      const bool lid = (r == a.accel_row) || (r == a.accel_row2);
      const bool own_row = (i >= K - 1) && (i < band_n + K - 1);
#pragma unroll
      for (int j = 0; j < C; j++) {
        const int jw = (j == 0) ? 0 : j - 1, je = (j == C - 1) ? C - 1 : j + 1;
        const float t[kQ] = {p.v[0][j],
                             (j == 0) ? e1 : p.v[1][jw],
                             p.v[2][j],
                             (j == C - 1) ? e3 : p.v[3][je],
                             p.v[4][j],
                             (j == 0) ? e5 : p.v[5][jw],
                             (j == C - 1) ? e6 : p.v[6][je],
                             (j == C - 1) ? e7 : p.v[7][je],
                             (j == 0) ? e8 : p.v[8][jw]};
        float speed;
        relax_cell<MATH, LBM_STREAM_SPEED>(t, ((p.m >> (8 * j)) & 0xffu) != 0, lid, a.omega, a.a1, a.a2, cur[j], speed,
                                           own_row);
        if (own_row && out_lane) sum[0] += speed;
      }
    }
    unsigned m_cur = p.m;  // mask bytes of the row `cur` belongs to

    // ---- stages 2..K: step t+s on row r-s from window s and the stage-s results of row r-s+1 ----
#pragma unroll
    for (int s = 1; s < K; s++) {
      Window<C>& w = win[s - 1];
      float nxt[C][kQ];
      const bool active = (i >= 2 * s);
      if (active) {
        const int ro = wrap_row(y0 - (K - 1) + i - s, a.rows, a.wrap);
        const float w1 = lane_from_west<2>(w.w013[1][C - 1]);
        const float w5 = lane_from_west<2>(w.w256[1][C - 1]);
        const float w8 = lane_from_west<2>(cur[C - 1][8]);
        const float x3 = lane_from_east<2>(w.w013[2][0]);
        const float x6 = lane_from_east<2>(w.w256[2][0]);
        const float x7 = lane_from_east<2>(cur[0][7]);
        const bool last = (s == K - 1);
        const bool lid = ((ro == a.accel_row) || (ro == a.accel_row2)) && (!last || a.accel_after);
        // rows of this band: ro in [y0, y0 + band_n)  <=>  i - s - (K-1) in [0, band_n)
        const bool own_row = (i - s >= K - 1) && (i - s < band_n + K - 1);
#pragma unroll
        for (int j = 0; j < C; j++) {
          const int jw = (j == 0) ? 0 : j - 1, je = (j == C - 1) ? C - 1 : j + 1;
          const float u[kQ] = {w.w013[0][j],
                               (j == 0) ? w1 : w.w013[1][jw],
                               w.w256[0][j],
                               (j == C - 1) ? x3 : w.w013[2][je],
                               cur[j][4],
                               (j == 0) ? w5 : w.w256[1][jw],
                               (j == C - 1) ? x6 : w.w256[2][je],
                               (j == C - 1) ? x7 : cur[je][7],
                               (j == 0) ? w8 : cur[jw][8]};
          float speed;
          relax_cell<MATH, LBM_STREAM_SPEED>(u, ((w.m >> (8 * j)) & 0xffu) != 0, lid, a.omega, a.a1, a.a2, nxt[j], speed,
                                             own_row);
          if (own_row && out_lane) sum[s] += speed;
        }
      }
      // rotate window s with the stage-s row just consumed
      const unsigned m_below = w.m;
#pragma unroll
      for (int j = 0; j < C; j++) {
        w.w256[0][j] = w.n256[0][j];  w.w256[1][j] = w.n256[1][j];  w.w256[2][j] = w.n256[2][j];
        w.n256[0][j] = cur[j][2];     w.n256[1][j] = cur[j][5];     w.n256[2][j] = cur[j][6];
        w.w013[0][j] = cur[j][0];     w.w013[1][j] = cur[j][1];     w.w013[2][j] = cur[j][3];
      }
      w.m = m_cur;
      if (!active) break;
      m_cur = m_below;
#pragma unroll
      for (int j = 0; j < C; j++)
#pragma unroll
        for (int k = 0; k < kQ; k++) cur[j][k] = nxt[j][k];
      if (s == K - 1 && out_lane) {
        const int ro = wrap_row(y0 - (K - 1) + i - s, a.rows, a.wrap);
        float* d_row = a.dst + (long)ro * a.row_pitch + x0;
#pragma unroll
        for (int k = 0; k < kQ; k++) {
          vec o;
#pragma unroll
          for (int j = 0; j < C; j++) o[j] = cur[j][k];
          if constexpr (NTS) __builtin_nontemporal_store(o, reinterpret_cast<vec*>(d_row + k * ps));
          else *reinterpret_cast<vec*>(d_row + k * ps) = o;
        }
      }
    }
  }

#pragma unroll
  for (int s = 0; s < K; s++) {
    const float tot = wave_sum(sum[s]);
    if (lane == 0) a.partials[(long)s * a.slot_stride + wave_id] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// Packed arithmetic: two cells per instruction.
//
// With three or four timesteps per pass the stream kernel is bound by VALU issue again (a plain wave64 VALU
// instruction holds a CDNA4 SIMD for 4 cycles whatever it computes).  gfx950 has packed fp32 VALU operations
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): one instruction, two independent IEEE fp32 results.  A lane's
// four cells are therefore kept as two PAIRS (cells 0,1 and 2,3: the two halves of the 16-byte vectors the lane
// loads and stores) and the whole collision is written on pairs.  Every packed operation below is the same
// IEEE operation, in the same order, as in collide_exact / collide_exact_body / moments_shared -- results are
// bit-identical; only the instruction count halves.  What cannot be packed stays per cell: v_rcp_f32, the range
// guards, v_sqrt_f32.  Blocked cells and the (rare, wave-uniform) lid row are patched afterwards; a lane whose
// guards fail (density outside [2^-60, 2^60) or |u|^2 >= 5e28: never in a physical run) recomputes its four
// cells with the scalar code, so the exactness argument of div_const_fast holds unchanged.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 div_const_fast2(f2 x, float C, float R) {
  const f2 q = x * splat2(R);
  const f2 r = fma2(splat2(-C), q, x);
  return fma2(r, splat2(R), q);
}
__device__ __forceinline__ f2 div_with_rcp2(f2 a, f2 b, f2 r) {
  f2 q = a * r;
  f2 e = fma2(-b, q, a);
  q = fma2(e, r, q);
  e = fma2(-b, q, a);
  return fma2(e, r, q);
}

// One PAIR of cells of one row, no exceptions: stream inputs f[k] -> relaxed r[k] as if both cells were fluid
// cells of an ordinary row; u_sq = their |u|^2 (pre-collision moments); returns false when a guard of the fast
// constant divides / the shared reciprocal fails for either cell.
__device__ __forceinline__ bool relax_pair_core(const f2 (&f)[kQ], float omega, f2 (&r)[kQ], f2& u_sq) {
  constexpr float kInvCsq = 1.0f / kCsq, kInvTwoCsq = 1.0f / kTwoCsq, kInvTwoCsqSq = 1.0f / kTwoCsqSq;
  f2 d = f[0];
#pragma unroll
  for (int k = 1; k < kQ; k++) d = d + f[k];
  const f2 X = ((f[1] + f[5]) + f[8]) - ((f[3] + f[6]) + f[7]);
  const f2 Y = ((f[2] + f[5]) + f[6]) - ((f[4] + f[7]) + f[8]);
  bool ok = density_in_plain_range(d.x) && density_in_plain_range(d.y);
  const f2 r0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 e0 = fma2(-d, r0, splat2(1.f));
  const f2 rr = fma2(e0, r0, r0);
  const f2 ux = div_with_rcp2(X, d, rr);
  const f2 uy = div_with_rcp2(Y, d, rr);
  const f2 uxx = ux * ux, uyy = uy * uy;
  u_sq = uxx + uyy;
  ok = ok && (u_sq.x < 5.0e28f) && (u_sq.y < 5.0e28f);
  const f2 usq_term = div_const_fast2(u_sq, kTwoCsq, kInvTwoCsq);
  const f2 one = splat2(1.f), om = splat2(omega);
  {
    const f2 w0r = splat2(kW0) * d;
    r[0] = f[0] + om * (w0r * (one - usq_term) - f[0]);
  }
  {
    const f2 w1r = splat2(kW1) * d;
    const f2 x1 = div_const_fast2(ux, kCsq, kInvCsq), x2 = div_const_fast2(uxx, kTwoCsqSq, kInvTwoCsqSq);
    r[1] = f[1] + om * (w1r * (((one + x1) + x2) - usq_term) - f[1]);
    r[3] = f[3] + om * (w1r * (((one - x1) + x2) - usq_term) - f[3]);
    const f2 y1 = div_const_fast2(uy, kCsq, kInvCsq), y2 = div_const_fast2(uyy, kTwoCsqSq, kInvTwoCsqSq);
    r[2] = f[2] + om * (w1r * (((one + y1) + y2) - usq_term) - f[2]);
    r[4] = f[4] + om * (w1r * (((one - y1) + y2) - usq_term) - f[4]);
  }
  {
    const f2 w2r = splat2(kW2) * d;
    const f2 us = ux + uy, ud = uy - ux;  // -ux + uy
    const f2 s1 = div_const_fast2(us, kCsq, kInvCsq), s2 = div_const_fast2(us * us, kTwoCsqSq, kInvTwoCsqSq);
    r[5] = f[5] + om * (w2r * (((one + s1) + s2) - usq_term) - f[5]);
    r[7] = f[7] + om * (w2r * (((one - s1) + s2) - usq_term) - f[7]);
    const f2 d1 = div_const_fast2(ud, kCsq, kInvCsq), d2 = div_const_fast2(ud * ud, kTwoCsqSq, kInvTwoCsqSq);
    r[6] = f[6] + om * (w2r * (((one + d1) + d2) - usq_term) - f[6]);
    r[8] = f[8] + om * (w2r * (((one - d1) + d2) - usq_term) - f[8]);
  }
  return ok;
}

// the exceptions of a pair.  Blocked cells bounce back (rebound(): the mirrored copy of the streamed populations,
// speed 0 kept, SerialCode/d2q9-bgk.c:291-298): a select per population half under ONE wave-level branch -- on the
// reference's geometry (walls every few hundred columns) a third of all pair relaxations meet a blocked cell somewhere
// in the wave, and the per-cell scalar path they used to take cost 13.5 % of the 8192^2 step (0.2826 vs 0.2443 ms
// without obstacles).  The lid row is accelerated by selects too (accelerate_select: the resident kernel's band
// that holds the lid row sets the pace of all the others, and the scalar path cost it 2.4 x the collision time);
// only a failed guard sends a cell through the scalar code (IEEE divides): rare, per cell.  sp0 / sp1: the cells' |u|
// (0 for blocked cells).
__device__ __forceinline__ void bounce_select(const f2 (&f)[kQ], bool b0, bool b1, f2 (&r)[kQ], float& sp0, float& sp1) {
  constexpr int opp[kQ] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
#pragma unroll
  for (int k = 0; k < kQ; k++) {
    r[k].x = b0 ? f[opp[k]].x : r[k].x;
    r[k].y = b1 ? f[opp[k]].y : r[k].y;
  }
  sp0 = b0 ? 0.f : sp0;
  sp1 = b1 ? 0.f : sp1;
}
// accelerate_flow on the relaxed cells of a pair (SerialCode/d2q9-bgk.c:196-213), per cell where c0 / c1 says so: the
// same three tests and six sums as accelerate(), as selects
__device__ __forceinline__ void accelerate_select(f2 (&r)[kQ], bool c0, bool c1, float a1, float a2) {
  const bool g0 = c0 && (r[3].x - a1) > 0.f && (r[6].x - a2) > 0.f && (r[7].x - a2) > 0.f;
  const bool g1 = c1 && (r[3].y - a1) > 0.f && (r[6].y - a2) > 0.f && (r[7].y - a2) > 0.f;
  constexpr int plus[3] = {1, 5, 8}, minus[3] = {3, 6, 7};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float a = (i == 0) ? a1 : a2;
    const f2 up = f2{r[plus[i]].x + a, r[plus[i]].y + a}, dn = f2{r[minus[i]].x - a, r[minus[i]].y - a};
    r[plus[i]] = f2{g0 ? up.x : r[plus[i]].x, g1 ? up.y : r[plus[i]].y};
    r[minus[i]] = f2{g0 ? dn.x : r[minus[i]].x, g1 ? dn.y : r[minus[i]].y};
  }
}
// the cells of a pair whose guard failed: the scalar code (IEEE divides), per cell -- rare
__device__ __forceinline__ void relax_pair_guard_path(const f2 (&f)[kQ], bool b0, bool b1, float omega, f2 (&r)[kQ],
                                                      bool want_speed, float& sp0, float& sp1) {
#pragma unroll
  for (int c = 0; c < 2; c++) {
    if (!(c == 0 ? b0 : b1)) {
      float ts[kQ], rs[kQ], speed;
#pragma unroll
      for (int k = 0; k < kQ; k++) ts[k] = f[k][c];
      relax_cell<0, 1>(ts, false, false, omega, 0.f, 0.f, rs, speed, want_speed);
#pragma unroll
      for (int k = 0; k < kQ; k++) r[k][c] = rs[k];
      if (c == 0) sp0 = speed; else sp1 = speed;
    }
  }
}
__device__ __forceinline__ void relax_pair_fixup(const f2 (&f)[kQ], bool ok, unsigned blocked, bool lid, float omega,
                                                 float a1, float a2, f2 (&r)[kQ], bool want_speed, float& sp0, float& sp1) {
  if (!ok || lid || blocked != 0) {  // ONE branch on the way of a wave without exceptions
    const bool b0 = (blocked & 0xffu) != 0, b1 = (blocked & 0xff00u) != 0;
    if (!ok) relax_pair_guard_path(f, b0, b1, omega, r, want_speed, sp0, sp1);
    if (lid) accelerate_select(r, !b0, !b1, a1, a2);
    if (blocked != 0) bounce_select(f, b0, b1, r, sp0, sp1);
  }
}

// One PAIR of cells of one row: stream inputs f[k] -> relaxed r[k]; returns the sum of |u| over its fluid cells
// (SPEED = 1 form: pre-collision moments, native sqrt) when want_speed.
// blocked: the pair's two mask bytes (bits 0-7, 8-15); lid: this row receives the next step's acceleration.
__device__ __forceinline__ float relax_pair_exact(const f2 (&f)[kQ], unsigned blocked, bool lid, float omega,
                                                  float a1, float a2, f2 (&r)[kQ], bool want_speed) {
  f2 u_sq;
  const bool ok = relax_pair_core(f, omega, r, u_sq);
  float sp0 = 0.f, sp1 = 0.f;
  if (want_speed) {
    sp0 = __builtin_amdgcn_sqrtf(u_sq.x);
    sp1 = __builtin_amdgcn_sqrtf(u_sq.y);
  }
  relax_pair_fixup(f, ok, blocked, lid, omega, a1, a2, r, want_speed, sp0, sp1);
  return sp0 + sp1;
}

// Both pairs of a lane in one basic block: two independent dependency chains the scheduler can interleave (with two
// waves per SIMD there is little else to fill the issue slots of a dependent chain); exceptions after both.
__device__ __forceinline__ float relax_quad_exact(const f2 (&f0)[kQ], const f2 (&f1)[kQ], unsigned blocked, bool lid,
                                                  float omega, float a1, float a2, f2 (&r0)[kQ], f2 (&r1)[kQ],
                                                  bool want_speed) {
  f2 usq0, usq1;
  const bool ok0 = relax_pair_core(f0, omega, r0, usq0);
  const bool ok1 = relax_pair_core(f1, omega, r1, usq1);
  float sp[4] = {0.f, 0.f, 0.f, 0.f};
  if (want_speed) {
    sp[0] = __builtin_amdgcn_sqrtf(usq0.x);  sp[1] = __builtin_amdgcn_sqrtf(usq0.y);
    sp[2] = __builtin_amdgcn_sqrtf(usq1.x);  sp[3] = __builtin_amdgcn_sqrtf(usq1.y);
  }
  relax_pair_fixup(f0, ok0, blocked & 0xffffu, lid, omega, a1, a2, r0, want_speed, sp[0], sp[1]);
  relax_pair_fixup(f1, ok1, blocked >> 16, lid, omega, a1, a2, r1, want_speed, sp[2], sp[3]);
  return (sp[0] + sp[1]) + (sp[2] + sp[3]);
}

// pair P of a plane (NP pairs per lane) shifted by one cell: the value each cell receives from its west (east)
// neighbour; W = the west lane's last cell, E = the east lane's first cell
template <int NP, int P>
__device__ __forceinline__ f2 pair_from_west(const f2 (&a)[NP], float W) {
  if constexpr (P == 0) return f2{W, a[0].x};
  else return f2{a[P - 1].y, a[P].x};
}
template <int NP, int P>
__device__ __forceinline__ f2 pair_from_east(const f2 (&a)[NP], float E) {
  if constexpr (P == NP - 1) return f2{a[P].y, E};
  else return f2{a[P].y, a[P + 1].x};
}

template <int NP>
struct WindowPk {
  f2 w256[3][NP];  // speeds 2,5,6 of the row two below the newest, as pairs
  f2 w013[3][NP];  // speeds 0,1,3 of the row below the newest
  f2 n256[3][NP];  // speeds 2,5,6 of the row below the newest
  unsigned m;
};

// stepk_stream with the collision on pairs (exact arithmetic, 4 cells per lane); same launch geometry and arguments.
// LW of the K-1 sliding windows (the last ones) live in LDS instead of registers: a window is 36 floats per lane,
// K = 4 needs three of them, and with all three in registers (252 VGPRs) nothing is left to prefetch the next row
// into -- at two waves per SIMD the kernel then waits out every row's load latency (8192^2: 0.284 ms per step,
// neither VALU- nor DRAM-bound).  An LDS window is nine 16-byte quads per lane (lane-linear: conflict-free
// ds_read_b128 / ds_write_b128), private to the lane that wrote it, so no barrier is involved; speeds 2,5,6 sit in
// a two-deep ring indexed by the row parity, speeds 0,1,3 in a single slot.  9 KB per window and wave.
// NP = pairs per lane: 2 (four cells, 16-byte accesses, K <= 4) or 1 (two cells, 8-byte accesses, K <= 2: a halo lane's
// two cells cover two steps) -- the form for grids too small to fill the chip with four-cell lanes.
template <bool NTS, int K, bool PREFETCH, int LW, bool QUAD = false, int NP = 2>
__global__ __launch_bounds__(64, NP == 1 ? (K > 2 ? 3 : (PREFETCH ? 2 : 4)) : 2) void stepk_pk(const StepKArgs a) {
  static_assert(K >= 2 && K <= 4 && LW >= 0 && LW <= K - 1 && (NP == 2 || !QUAD), "halo_lanes * C >= K is the host's job");
  constexpr int C = 2 * NP;
  typedef typename RowPull<C>::vec vec;
  __shared__ vec lds_win[LW > 0 ? LW : 1][9][64];
  const int lane = threadIdx.x;
  int strip, band;
  if (a.chunk > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunks_per_band = (a.n_strips + a.chunk - 1) / a.chunk;
    const int g = (j / a.chunk) * 8 + xcd;
    band = g / chunks_per_band;
    strip = (g - band * chunks_per_band) * a.chunk + j % a.chunk;
    if (band >= a.n_bands || strip >= a.n_strips) return;
  } else {
    strip = blockIdx.x % a.n_strips;
    band = blockIdx.x / a.n_strips;
  }
  const int wave_id = band * a.n_strips + strip;
  const int units_x = a.nx / C;
  const int y0 = a.row_first + band * a.band_pitch;
  const int band_n = min(a.band_rows, a.row_end - y0);

  const int H = a.halo_lanes;  // H * C >= K cells of halo per side
  const int ux_raw = strip * (64 - 2 * H) + lane - H;
  int ux = ux_raw % units_x;
  if (ux < 0) ux += units_x;
  const int x0 = ux * C;
  const bool out_lane = (lane >= H) && (lane < 64 - H) && (ux_raw < units_x);
  const long ps = a.plane_stride;

  Step2Args pa;
  pa.src = a.src;  pa.mask = a.mask;  pa.plane_stride = a.plane_stride;  pa.row_pitch = a.row_pitch;
  pa.pitch = a.pitch;  pa.rows = a.rows;  pa.wrap = a.wrap;

  WindowPk<NP> win[K - 1];  // the first K-1-LW are used (registers); the others only for their mask bytes
#pragma unroll
  for (int s = 0; s < K - 1; s++) {
    win[s].m = 0;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int p = 0; p < NP; p++) win[s].w256[q][p] = win[s].w013[q][p] = win[s].n256[q][p] = splat2(0.f);
  }
  if constexpr (LW > 0) {
#pragma unroll
    for (int l = 0; l < LW; l++)
#pragma unroll
      for (int q = 0; q < 9; q++) {
        vec z;
#pragma unroll
        for (int j = 0; j < C; j++) z[j] = 0.f;
        lds_win[l][q][lane] = z;
      }
  }
  float sum[K];
#pragma unroll
  for (int s = 0; s < K; s++) sum[s] = 0.f;

  // the NP pairs of one plane of a loaded / LDS vector, and back
  auto pairs_of = [](const vec& v, f2 (&o)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; p++) o[p] = f2{v[2 * p], v[2 * p + 1]};
  };
  auto vec_of = [](const f2 (&c)[NP][kQ], int k) {
    vec o;
#pragma unroll
    for (int p = 0; p < NP; p++) { o[2 * p] = c[p][k].x;  o[2 * p + 1] = c[p][k].y; }
    return o;
  };
  // relax the NP pairs of one row: inputs by plane (already shifted), pair-major outputs
  auto relax_row = [&](const f2 (&in)[kQ][NP], unsigned m, bool lid, f2 (&out)[NP][kQ], bool own_row) {
    f2 t0[kQ];
#pragma unroll
    for (int k = 0; k < kQ; k++) t0[k] = in[k][0];
    if constexpr (NP == 1) {
      return relax_pair_exact(t0, m & 0xffffu, lid, a.omega, a.a1, a.a2, out[0], own_row);
    } else {
      f2 t1[kQ];
#pragma unroll
      for (int k = 0; k < kQ; k++) t1[k] = in[k][NP - 1];
      if constexpr (QUAD) {
        return relax_quad_exact(t0, t1, m, lid, a.omega, a.a1, a.a2, out[0], out[NP - 1], own_row);
      } else {
        const float s0 = relax_pair_exact(t0, m & 0xffffu, lid, a.omega, a.a1, a.a2, out[0], own_row);
        return s0 + relax_pair_exact(t1, m >> 16, lid, a.omega, a.a1, a.a2, out[NP - 1], own_row);
      }
    }
  };
  // the nine planes a cell pulls from: planes 0,2,4 as they are, 1,5,8 from the west, 3,6,7 from the east
  auto shifted = [](const f2 (&pl)[kQ][NP], const float (&W)[kQ], const float (&E)[kQ], f2 (&in)[kQ][NP]) {
#pragma unroll
    for (int k = 0; k < kQ; k++) {
      const bool west = (k == 1 || k == 5 || k == 8), east = (k == 3 || k == 6 || k == 7);
      if (west) {
        in[k][0] = pair_from_west<NP, 0>(pl[k], W[k]);
        if constexpr (NP == 2) in[k][NP - 1] = pair_from_west<NP, NP - 1>(pl[k], W[k]);
      } else if (east) {
        in[k][0] = pair_from_east<NP, 0>(pl[k], E[k]);
        if constexpr (NP == 2) in[k][NP - 1] = pair_from_east<NP, NP - 1>(pl[k], E[k]);
      } else {
#pragma unroll
        for (int p = 0; p < NP; p++) in[k][p] = pl[k][p];
      }
    }
  };

  const int n_iter = band_n + 2 * (K - 1);
  // Row bookkeeping kept incrementally (all wave-uniform, i.e. scalar registers): the row stage 1 pulls advances by
  // one per iteration, and stage s+1 works on the row stage 1 had s iterations ago -- so "is this the lid row" and
  // "does this row belong to the band" are shift registers (bit s = stage s+1), and the row the last stage stores is
  // the oldest of a K-deep history, instead of a wrap, two compares and a window test per stage, pair and iteration.
  int r_next = wrap_row(y0 - (K - 1), a.rows, a.wrap);  // the row iteration i pulls
  int r_hist[K];                                        // r_hist[s] = the row stage s+1 works on
#pragma unroll
  for (int s = 0; s < K; s++) r_hist[s] = 0;
  unsigned lid_hist = 0, own_hist = 0;
  RowPull<C> nextp;
  if constexpr (PREFETCH) nextp = pull_row<C>(pa, r_next, x0);

  // One row iteration.  WARM: the first 2 (K-1) iterations of a band, while the later stages have nothing to work on
  // yet; the iterations after them run a copy of the body without those tests (every stage active: no merges of
  // "relaxed" and "skipped" register sets, whose copies were 5 % of the steady state's vector instructions).
  auto row_iteration = [&](auto warm_c, const int i) {
    constexpr bool WARM = decltype(warm_c)::value;
    // ---- stage 1: step t on row r, pulled from memory -------------------------------------------
    const int r = r_next;
    r_next = r + 1;
    if (a.wrap && r_next >= a.rows) r_next -= a.rows;
#pragma unroll
    for (int s = K - 1; s > 0; s--) r_hist[s] = r_hist[s - 1];
    r_hist[0] = r;
    lid_hist = (lid_hist << 1) | ((r == a.accel_row || r == a.accel_row2) ? 1u : 0u);
    own_hist = (own_hist << 1) | ((i >= K - 1 && i < band_n + K - 1) ? 1u : 0u);
    RowPull<C> p;
    if constexpr (PREFETCH) {
      p = nextp;
      if (i + 1 < n_iter) nextp = pull_row<C>(pa, r_next, x0);
    } else {
      p = pull_row<C>(pa, r, x0);
    }
    f2 cur[NP][kQ];
    {
      float W[kQ] = {}, E[kQ] = {};
      W[1] = lane_from_west<2>(p.v[1][C - 1]);  W[5] = lane_from_west<2>(p.v[5][C - 1]);  W[8] = lane_from_west<2>(p.v[8][C - 1]);
      E[3] = lane_from_east<2>(p.v[3][0]);      E[6] = lane_from_east<2>(p.v[6][0]);      E[7] = lane_from_east<2>(p.v[7][0]);
      f2 pl[kQ][NP], in[kQ][NP];
#pragma unroll
      for (int k = 0; k < kQ; k++) pairs_of(p.v[k], pl[k]);
      shifted(pl, W, E, in);
      const bool lid = (lid_hist & 1u) != 0;
      const bool own_row = (own_hist & 1u) != 0;
      const float sp = relax_row(in, p.m, lid, cur, own_row);
      if (own_row && out_lane) sum[0] += sp;
    }
    unsigned m_cur = p.m;

    // ---- stages 2..K: step t+s on row r-s from window s and the stage-s results of row r-s+1 ------
    bool alive = true;
    auto stage = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr bool in_lds = (s - 1) >= (K - 1 - LW);
      constexpr int li = in_lds ? (s - 1) - (K - 1 - LW) : 0;
      if (!alive) return;
      WindowPk<NP>& w = win[s - 1];
      const int ring = (i & 1) * 3;
      f2 nxt[NP][kQ];
      const bool active = WARM ? (i >= 2 * s) : true;
      if (active) {
        // planes of the window rows: 2,5,6 of row ro-1, 0,1,3 of row ro, and 4,7,8 of row ro+1 (= cur)
        f2 pl[kQ][NP];
        constexpr int q256[3] = {2, 5, 6}, q013[3] = {0, 1, 3};
        if constexpr (in_lds) {
#pragma unroll
          for (int q = 0; q < 3; q++) {
            pairs_of(lds_win[li][ring + q][lane], pl[q256[q]]);
            pairs_of(lds_win[li][6 + q][lane], pl[q013[q]]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 3; q++)
#pragma unroll
            for (int p2 = 0; p2 < NP; p2++) { pl[q256[q]][p2] = w.w256[q][p2];  pl[q013[q]][p2] = w.w013[q][p2]; }
        }
#pragma unroll
        for (int p2 = 0; p2 < NP; p2++) { pl[4][p2] = cur[p2][4];  pl[7][p2] = cur[p2][7];  pl[8][p2] = cur[p2][8]; }
        float W[kQ] = {}, E[kQ] = {};
        W[1] = lane_from_west<2>(pl[1][NP - 1].y);  W[5] = lane_from_west<2>(pl[5][NP - 1].y);  W[8] = lane_from_west<2>(pl[8][NP - 1].y);
        E[3] = lane_from_east<2>(pl[3][0].x);       E[6] = lane_from_east<2>(pl[6][0].x);       E[7] = lane_from_east<2>(pl[7][0].x);
        f2 in[kQ][NP];
        shifted(pl, W, E, in);
        const bool last = (s == K - 1);
        const bool lid = ((lid_hist >> s) & 1u) != 0 && (!last || a.accel_after);
        const bool own_row = ((own_hist >> s) & 1u) != 0;
        const float sp = relax_row(in, w.m, lid, nxt, own_row);
        if (own_row && out_lane) sum[s] += sp;
      }
      // rotate window s with the stage-s row just consumed
      const unsigned m_below = w.m;
      if constexpr (in_lds) {
        lds_win[li][ring + 0][lane] = vec_of(cur, 2);
        lds_win[li][ring + 1][lane] = vec_of(cur, 5);
        lds_win[li][ring + 2][lane] = vec_of(cur, 6);
        lds_win[li][6][lane] = vec_of(cur, 0);
        lds_win[li][7][lane] = vec_of(cur, 1);
        lds_win[li][8][lane] = vec_of(cur, 3);
      } else {
#pragma unroll
        for (int p2 = 0; p2 < NP; p2++) {
          w.w256[0][p2] = w.n256[0][p2];  w.w256[1][p2] = w.n256[1][p2];  w.w256[2][p2] = w.n256[2][p2];
          w.n256[0][p2] = cur[p2][2];     w.n256[1][p2] = cur[p2][5];     w.n256[2][p2] = cur[p2][6];
          w.w013[0][p2] = cur[p2][0];     w.w013[1][p2] = cur[p2][1];     w.w013[2][p2] = cur[p2][3];
        }
      }
      w.m = m_cur;
      if (!active) { alive = false; return; }
      m_cur = m_below;
#pragma unroll
      for (int p2 = 0; p2 < NP; p2++)
#pragma unroll
        for (int k = 0; k < kQ; k++) cur[p2][k] = nxt[p2][k];
      if (s == K - 1 && out_lane) {
        const int ro = r_hist[K - 1];
        float* d_row = a.dst + (long)ro * a.row_pitch + x0;
#pragma unroll
        for (int k = 0; k < kQ; k++) {
          const vec o = vec_of(cur, k);
          if constexpr (NTS) __builtin_nontemporal_store(o, reinterpret_cast<vec*>(d_row + k * ps));
          else *reinterpret_cast<vec*>(d_row + k * ps) = o;
        }
      }
    };
    stage(std::integral_constant<int, 1>{});
    if constexpr (K > 2) stage(std::integral_constant<int, 2>{});
    if constexpr (K > 3) stage(std::integral_constant<int, 3>{});
  };
  const int n_warm = min(n_iter, 2 * (K - 1));
  int i = 0;
  for (; i < n_warm; i++) row_iteration(std::true_type{}, i);
  for (; i < n_iter; i++) row_iteration(std::false_type{}, i);

#pragma unroll
  for (int s = 0; s < K; s++) {
    const float tot = wave_sum(sum[s]);
    if (lane == 0) a.partials[(long)s * a.slot_stride + wave_id] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// Up to KMAX timesteps per launch from an LDS tile (temporal blocking for small, cache-resident grids,
// single periodic slab).
//
// A grid of a few hundred thousand cells is not bound by bandwidth or arithmetic but by the latency of
// one kernel (~3 us): every launch pays it, so the only lever is more timesteps per launch.  A workgroup
// stages its TW x TH tile plus a halo of KMAX cells on every side -- all 9 populations and the obstacle
// flags -- in LDS, and relaxes it n <= KMAX times in place: step j covers the cells at distance >= j from
// the staged border (their 9 sources are cells of step j-1 inside the region), each thread pulls its
// cells' populations from LDS into registers, the workgroup synchronises, the results go back to LDS.
// After n steps the tile's own cells are exact and are stored; the halo cells were relaxed redundantly,
// exactly as the neighbouring workgroups relax them (same per-cell code as every other kernel, so the
// lattice stays bit-identical).  Periodic wrap in x and y is resolved when the tile is staged.
// Sum of |u| per step: over the tile's own cells, one partial per workgroup and step.
// ---------------------------------------------------------------------------------------------
struct TileArgs {
  const float* src;
  float* dst;
  const unsigned char* mask;
  long plane_stride;
  long row_pitch;
  int pitch;
  int nx, ny;
  int tiles_x;      // workgroups across x
  int n_steps;      // timesteps this launch advances (1..KMAX)
  int accel_row;    // global row that receives accelerate_flow (ny - 2)
  int accel_after;  // also apply the acceleration of the step after this launch's last one
  float omega, a1, a2;
  float* partials;  // partials[j * slot_stride + workgroup] = sum |u| of the tile's own cells after step j
  long slot_stride;
};

template <int MATH, int TW, int TH, int KMAX, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS) void step_tile(const TileArgs a) {
  constexpr int EW = TW + 2 * KMAX, EH = TH + 2 * KMAX, EC = EW * EH;
  constexpr int EWP = EW + 1;                                           // row padding against bank conflicts
  constexpr int MAXC = ((EW - 2) * (EH - 2) + THREADS - 1) / THREADS;   // cells per thread in step 1
  __shared__ float f[kQ][EH][EWP];
  __shared__ unsigned char blocked[EH][EW];
  __shared__ int global_row[EH];

  const int tid = threadIdx.x;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int gx0 = bx * TW - KMAX, gy0 = by * TH - KMAX;
  const long ps = a.plane_stride;

  // stage the tile and its halo (periodic images where the halo leaves the grid)
  for (int idx = tid; idx < EC; idx += THREADS) {
    const int y = idx / EW, x = idx - y * EW;
    int gy = (gy0 + y) % a.ny;  if (gy < 0) gy += a.ny;
    int gx = (gx0 + x) % a.nx;  if (gx < 0) gx += a.nx;
    const float* cell = a.src + (long)gy * a.row_pitch + gx;
#pragma unroll
    for (int k = 0; k < kQ; k++) f[k][y][x] = cell[k * ps];
    blocked[y][x] = a.mask[(long)gy * a.pitch + gx];
    if (x == 0) global_row[y] = gy;
  }
  __syncthreads();

  for (int j = 1; j <= a.n_steps; j++) {
    const int w = EW - 2 * j, n = w * (EH - 2 * j);
    const bool accel = (j < a.n_steps) || a.accel_after;
    float r[MAXC][kQ];
    float my_sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      const int idx = tid + c * THREADS;
      if (idx < n) {
        const int ry = idx / w;
        const int y = j + ry, x = j + (idx - ry * w);
        // pull (SerialCode/d2q9-bgk.c:263-271): speed k arrives from the cell it left one step ago
        const float t[kQ] = {f[0][y][x],         f[1][y][x - 1],     f[2][y - 1][x],
                             f[3][y][x + 1],     f[4][y + 1][x],     f[5][y - 1][x - 1],
                             f[6][y - 1][x + 1], f[7][y + 1][x + 1], f[8][y + 1][x - 1]};
        const int oy = y - KMAX, ox = x - KMAX;  // position inside the tile's own cells
        const bool own = (oy >= 0) && (oy < TH) && (ox >= 0) && (ox < TW) && (by * TH + oy < a.ny) &&
                         (bx * TW + ox < a.nx);
        const bool lid = accel && (global_row[y] == a.accel_row);
        float speed;
        relax_cell<MATH>(t, blocked[y][x] != 0, lid, a.omega, a.a1, a.a2, r[c], speed, own);
        if (own) my_sum += speed;
      }
    }
    __syncthreads();  // every source of this step has been read
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      const int idx = tid + c * THREADS;
      if (idx < n) {
        const int ry = idx / w;
        const int y = j + ry, x = j + (idx - ry * w);
#pragma unroll
        for (int k = 0; k < kQ; k++) f[k][y][x] = r[c][k];
      }
    }
    const float total = block_sum<THREADS>(my_sum);  // synchronises the workgroup
    if (tid == 0) a.partials[(long)(j - 1) * a.slot_stride + blockIdx.x] = total;
    __syncthreads();
  }

  // the tile's own cells
  for (int idx = tid; idx < TW * TH; idx += THREADS) {
    const int oy = idx / TW, ox = idx - oy * TW;
    const int gy = by * TH + oy, gx = bx * TW + ox;
    if (gy < a.ny && gx < a.nx) {
      float* cell = a.dst + (long)gy * a.row_pitch + gx;
#pragma unroll
      for (int k = 0; k < kQ; k++) cell[k * ps] = f[k][oy + KMAX][ox + KMAX];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Resident lattice: ONE launch advances the whole (single, periodic) slab by many timesteps with the lattice kept
// in registers -- the form for cache-resident grids (the reference's own data sets, 128^2 ... 1024^2 x 20 000 - 80 000
// steps, dataSet/input_*.params; loop SerialCode/d2q9-bgk.c:166-170), where a launch per pass costs a kernel
// boundary (~1.5-3 us) plus a reload of the lattice through L2 however little it computes.
//
// Geometry: workgroup b owns the band of rows [4b, 4b+4), the full width; thread x owns column x of the band: four
// cells, kept as two PAIRS for the packed arithmetic -- the interior pair (rows 1, 2) and the edge pair (rows 0, 3).
// 36 VGPRs of state per lane; a 1024 x 1024 lattice is 256 workgroups of 1024 threads, one per CU.  (ROWS = 2: bands
// of two rows, one pair per lane, where the chip has CUs to spare.)
//   streaming in y inside the band : register renaming (free)
//   streaming in x                 : DPP wave_shr / wave_shl; lane 0 / 63 of a wave take the neighbouring wave's edge
//                                    lane from LDS (one s_barrier per timestep, two LDS slots by parity)
//   rows 0 and 3 also need the adjacent row of the neighbouring BAND: the 3 populations that cross the seam travel
//   through L2 as one 16-byte granule per cell and direction, {v, v, v, tag} -- ONE sc1 (write-through, L1-bypassing)
//   dwordx4 store, naturally aligned, so it lies inside one 64-byte memory request on the way out and on the way in;
//   the tag is the global timestep index + 1, i.e. the data is its own flag: no fence, no separate flag, no
//   device-wide barrier.  (8-byte {value, tag} granules, the form MI355X_MICROARCH.md prices, cost three narrow sc1
//   stores per cell and direction: 12.6 MB per step at 1024^2, store-bound at 5.9 us per step.)
//   Two slots by step parity: a slot is overwritten two steps later, by which time the neighbour has provably read
//   it (it cannot have published the step in between otherwise).  A lane loads the granule of its own column; the
//   diagonal populations shift by DPP, the wave's first / last lane taking the column beyond from a second load.
// Per timestep: publish the edge rows (computed last in the previous step) -> LDS edge lanes -> barrier -> relax the
// interior pair (hides the hop) -> poll the six halo granules -> relax the edge pair.  No redundant work at all.
// Every spin is bounded (wall clock): on expiry the workgroup raises *status and leaves; every other workgroup
// sees the status in its own spin and leaves too.  An error, never a hang.  The grid must be co-resident (host:
// one workgroup per CU at most as many as the device has CUs).
// Arithmetic: relax_pair_core + the per-cell exception path, i.e. the lattice stays bit-identical to the oracle;
// |u| from the pre-collision moments as in the stream kernels.
// ---------------------------------------------------------------------------------------------
struct ResidentArgs {
  const float* src;           // lattice to start from (row 0 of the slab)
  float* dst;                 // lattice to leave the result in (may equal src)
  const unsigned char* mask;
  long plane_stride;
  long row_pitch;
  int pitch;
  int nx, ny;
  int n_steps;                // timesteps this launch advances
  int accel_row;              // global row of accelerate_flow (ny - 2)
  int accel_last;             // also apply the acceleration of the step after this launch's last one
  float omega, a1, a2;
  uint4* gran;                // seam granules {v, v, v, tag}: [2 directions][bands][2 slots][nx]
  unsigned gran_bytes;        // size of the granule buffer
  unsigned epoch0;            // tag of this launch's step s = epoch0 + s + 1 (global step index + 1: never repeats)
  float* partials;            // partials[s * bands + b] = sum |u| over band b after step s
  int* status;                // 0, or kResidentTimeout once any workgroup gave up waiting
  long long timeout_ticks;    // bound of one halo wait in wall_clock64() ticks (100 MHz)
  int xcd_affinity;           // 1: seams inside one XCD use L2-resident stores (see resident_band); 0: sc1 everywhere
  int absent_band;            // tests: this band's workgroup returns at once, as if it had never been scheduled (-1: none)
  int group;                  // bands per workgroup (blockDim.x = group * nx)
  int one_xcd;                // 1: the launch has 8 workgroups per working one and only those dealt to the first XCD work
#ifdef LBM_RESIDENT_PROFILE
  long long* prof;            // tools/resident_profile.sh: [band][8] shader-clock sums of the phases of a step
#endif
};
#ifdef LBM_RESIDENT_PROFILE
__device__ __forceinline__ long long prof_clock() {
  long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define RESIDENT_PROF(i) do { const long long now_ = prof_clock(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define RESIDENT_PROF(i) do { } while (0)
#endif
constexpr int kResidentTimeout = 1;

// per-cell lid flags (bit 0: cell .x, bit 1: cell .y): the pair's two cells lie in different rows here
__device__ __forceinline__ void relax_pair_fixup_rows(const f2 (&f)[kQ], bool ok, unsigned blocked, unsigned lid, float omega,
                                                      float a1, float a2, f2 (&r)[kQ], float& sp0, float& sp1) {
  if (!ok || lid != 0 || blocked != 0) {
    const bool b0 = (blocked & 0xffu) != 0, b1 = (blocked & 0xff00u) != 0;
    if (!ok) relax_pair_guard_path(f, b0, b1, omega, r, true, sp0, sp1);
    if (lid != 0) accelerate_select(r, (lid & 1u) != 0 && !b0, (lid & 2u) != 0 && !b1, a1, a2);
    if (blocked != 0) bounce_select(f, b0, b1, r, sp0, sp1);
  }
}
__device__ __forceinline__ float relax_pair_rows(const f2 (&f)[kQ], unsigned blocked, unsigned lid, float omega, float a1,
                                                 float a2, f2 (&r)[kQ]) {
  f2 u_sq;
  const bool ok = relax_pair_core(f, omega, r, u_sq);
  float sp0 = __builtin_amdgcn_sqrtf(u_sq.x), sp1 = __builtin_amdgcn_sqrtf(u_sq.y);
  relax_pair_fixup_rows(f, ok, blocked, lid, omega, a1, a2, r, sp0, sp1);
  return sp0 + sp1;
}

// sum over the first 16 lanes (one DPP row), valid in every lane of that row
__device__ __forceinline__ float row16_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// both pairs of a lane in one basic block (two independent dependency chains), the exceptions after both
__device__ __forceinline__ float relax_two_pairs_rows(const f2 (&fa)[kQ], const f2 (&fb)[kQ], unsigned blocked_a, unsigned blocked_b,
                                                      unsigned lid_a, unsigned lid_b, float omega, float a1, float a2,
                                                      f2 (&ra)[kQ], f2 (&rb)[kQ]) {
  f2 usq_a, usq_b;
  const bool ok_a = relax_pair_core(fa, omega, ra, usq_a);
  const bool ok_b = relax_pair_core(fb, omega, rb, usq_b);
  float sp[4] = {__builtin_amdgcn_sqrtf(usq_a.x), __builtin_amdgcn_sqrtf(usq_a.y), __builtin_amdgcn_sqrtf(usq_b.x), __builtin_amdgcn_sqrtf(usq_b.y)};
  relax_pair_fixup_rows(fa, ok_a, blocked_a, lid_a, omega, a1, a2, ra, sp[0], sp[1]);
  relax_pair_fixup_rows(fb, ok_b, blocked_b, lid_b, omega, a1, a2, rb, sp[2], sp[3]);
  return (sp[0] + sp[1]) + (sp[2] + sp[3]);
}

// lane i <- lane i-1 (i+1); the wave's first (last) lane, which has no such lane, keeps `edge`
__device__ __forceinline__ float shift_from_west(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float shift_from_east(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// sum over the wave by DPP (no LDS); valid in lane 63
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
  return v;
}

// 16-byte seam granule: three populations and the tag in ONE naturally aligned dwordx4 access with sc1 (agent scope:
// the store writes through the XCD's L2, the load bypasses L1 and refetches), i.e. the raw-buffer forms with aux = sc1
typedef int granule_vec __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t granule_rsrc(uint4* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void granule_store(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float a, float b, float c, unsigned tag) {
  const granule_vec v = {(int)__float_as_uint(a), (int)__float_as_uint(b), (int)__float_as_uint(c), (int)tag};
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, 16);  // aux 16 = sc1
}
// the same store without sc1: the line stays (dirty) in this XCD's L2, where a reader on the SAME XCD finds it with
// its sc1 (L1-bypassing) load at L2 latency instead of a round trip over the fabric
__device__ __forceinline__ void granule_store_local(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float a, float b, float c, unsigned tag) {
  const granule_vec v = {(int)__float_as_uint(a), (int)__float_as_uint(b), (int)__float_as_uint(c), (int)tag};
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, 0);
}
__device__ __forceinline__ granule_vec granule_load(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);
}

// JOINT: wait for the halo first and relax both pairs as ONE block of independent work -- for narrow grids, whose few
// waves sit alone on their SIMDs: there the step is the dependent-instruction latency of the two collisions one after
// the other, and two independent chains interleave (issue-bound instead of latency-bound).  Wide grids (four waves
// per SIMD) keep the interior pair in front of the halo wait: their SIMDs are busy anyway and the interior pair
// hides the hop.
// ROWS: rows per band, 4 (above) or 2 -- one pair per lane, both rows seam rows: twice the workgroups and half the
// dependent work per lane and step, for grids small enough that the chip has CUs to spare (ny / 2 <= CUs): there the
// step is one wave's chain "hop + collisions", and the shorter chain wins although nothing hides the hop any more.
template <int MAXT, bool JOINT = false, int ROWS = 4>
__global__ __launch_bounds__(MAXT) void resident_band(const ResidentArgs a) {
  static_assert(ROWS == 4 || ROWS == 2, "bands of four or two rows");
  constexpr int NE = (ROWS == 4) ? 10 : 4;  // wave-edge values per side
  // a workgroup holds a.group bands side by side (1: the usual case; more where a band has fewer than four waves and
  // the whole grid fits one XCD with one wave per SIMD, see a.one_xcd): `wave`, `n_waves` count inside the band,
  // `wv` inside the workgroup
  const int grp = (int)threadIdx.x / a.nx;
  const int x = (int)threadIdx.x - grp * a.nx, lane = x & 63, wave = x >> 6, n_waves = a.nx >> 6, wv = (int)threadIdx.x >> 6, wv0 = wv - wave;
  const int n_wgs = a.one_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x, wg = a.one_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (a.one_xcd && (blockIdx.x & 7) != 0) return;
  const int bands = n_wgs * a.group;
  // Workgroups are dealt to the 8 XCDs round-robin (observed, not promised): consecutive bands are given to
  // workgroups 8 apart, so that most seams join two bands on ONE XCD.  Speed only -- which seams really do is
  // established below from the hardware's own XCC id, and the protocol is correct for any placement.
  const int b = a.group * ((!a.one_xcd && a.xcd_affinity && (n_wgs & 7) == 0) ? (wg & 7) * (n_wgs >> 3) + (wg >> 3) : wg) + grp;
  const long ps = a.plane_stride;
  // wave-edge values: [parity][wave][side: 0 = lane 0's west-moving, 1 = lane 63's east-moving][NE used of 12]
  __shared__ __attribute__((aligned(16))) float edge[2][MAXT / 64][2][12];
  __shared__ float wave_part[2][MAXT / 64];
  if (b == a.absent_band) return;

  // ---- the band's rows.  ROWS = 4: interior pair ri = rows (1, 2), edge pair re = rows (0, 3); ROWS = 2: re = rows (0, 1)
  constexpr int TOP = ROWS - 1;  // the band's last row: the .y half of the edge pair
  f2 ri[kQ], re[kQ];
  {
    const float* base = a.src + (long)(ROWS * b) * a.row_pitch + x;
#pragma unroll
    for (int k = 0; k < kQ; k++) {
      re[k] = f2{base[k * ps], base[TOP * a.row_pitch + k * ps]};
      if constexpr (ROWS == 4) ri[k] = f2{base[1 * a.row_pitch + k * ps], base[2 * a.row_pitch + k * ps]};
      else ri[k] = splat2(0.f);
    }
  }
  const unsigned char* mp = a.mask + (long)(ROWS * b) * a.pitch + x;
  const unsigned blocked_e = (unsigned)mp[0] | ((unsigned)mp[TOP * a.pitch] << 8);
  unsigned blocked_i = 0;
  if constexpr (ROWS == 4) blocked_i = (unsigned)mp[a.pitch] | ((unsigned)mp[2 * a.pitch] << 8);
  // accelerate_flow's row, if this band holds it: bit per cell of the pair it falls in
  const int lid_local = a.accel_row - ROWS * b;
  const unsigned lid_i = (ROWS == 4) ? ((lid_local == 1) ? 1u : (lid_local == 2 ? 2u : 0u)) : 0u;
  const unsigned lid_e = (lid_local == 0) ? 1u : (lid_local == TOP ? 2u : 0u);

  // seam granules: `up` carries a band's top-row populations 2,5,6 northwards, `down` its row-0 populations 4,7,8;
  // byte offsets into the one buffer (32-bit: it is at most 16 MiB)
  const __amdgpu_buffer_rsrc_t grsrc = granule_rsrc(a.gran, a.gran_bytes);
  const unsigned band_bytes = 2u * (unsigned)a.nx * 16u, slot_bytes = (unsigned)a.nx * 16u;
  const unsigned down_base = (unsigned)bands * band_bytes;
  const int bs = (b == 0) ? bands - 1 : b - 1, bn = (b == bands - 1) ? 0 : b + 1;
  const int xw = (x == 0) ? a.nx - 1 : x - 1, xe = (x == a.nx - 1) ? 0 : x + 1;
  const unsigned my_up = (unsigned)b * band_bytes + (unsigned)x * 16u;
  const unsigned my_down = down_base + (unsigned)b * band_bytes + (unsigned)x * 16u;
  const unsigned from_south = (unsigned)bs * band_bytes, from_north = down_base + (unsigned)bn * band_bytes;
  // which XCD do my neighbours run on?  Every band announces its XCC id in a granule of its own (sc1, tag = this
  // launch's first epoch); a seam whose two bands share an XCD keeps its granules in that XCD's L2 (plain stores),
  // any other seam writes them through (sc1).  The reader's loads are sc1 either way.
  bool north_local = false, south_local = false;
  if (a.xcd_affinity) {
    const unsigned hello_base = 2u * down_base;
    const unsigned my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;  // HW_REG_XCC_ID[3:0]
    const unsigned hello_tag = a.epoch0 + 1u;
    if (x == 0) granule_store(grsrc, hello_base + (unsigned)b * 16u, __uint_as_float(my_xcc), 0.f, 0.f, hello_tag);
    long long t_start = 0;
    for (unsigned spins = 0;; spins++) {
      asm volatile("" ::: "memory");
      const granule_vec hs = granule_load(grsrc, hello_base + (unsigned)bs * 16u);
      const granule_vec hn = granule_load(grsrc, hello_base + (unsigned)bn * 16u);
      if ((unsigned)hs.w == hello_tag && (unsigned)hn.w == hello_tag) {
        south_local = ((unsigned)hs.x == my_xcc);
        north_local = ((unsigned)hn.x == my_xcc);
        break;
      }
      if ((spins & 63u) == 63u) {
        const long long now = wall_clock64();
        if (t_start == 0) t_start = now;
        const int st = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st != 0 || now - t_start > a.timeout_ticks) {
          if (st == 0 && lane == 0) __hip_atomic_store(a.status, kResidentTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  const int west_wave = wv0 + ((wave == 0) ? n_waves - 1 : wave - 1), east_wave = wv0 + ((wave == n_waves - 1) ? 0 : wave + 1);

  // publish the edge rows of a state: step s of this launch reads what was published with its tag into its slot
  auto publish = [&](int s) {
    const unsigned tag = a.epoch0 + (unsigned)s + 1u;
    const unsigned off = (unsigned)(s & 1) * slot_bytes;
    if (north_local) granule_store_local(grsrc, my_up + off, re[2].y, re[5].y, re[6].y, tag);
    else granule_store(grsrc, my_up + off, re[2].y, re[5].y, re[6].y, tag);
    if (south_local) granule_store_local(grsrc, my_down + off, re[4].x, re[7].x, re[8].x, tag);
    else granule_store(grsrc, my_down + off, re[4].x, re[7].x, re[8].x, tag);
  };
  if (a.n_steps > 0) publish(0);

  bool alive = true;
#ifdef LBM_RESIDENT_PROFILE
  long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = prof_clock();
#endif
  for (int s = 0; s < a.n_steps && alive; s++) {
    const unsigned tag = a.epoch0 + (unsigned)s + 1u;
    const int slot = s & 1;
    // (the edge rows of the current state were published as soon as they existed: before the loop / at the end of
    // the previous iteration)
    // ---- wave-edge lanes through LDS ------------------------------------------------------------------------
    // ROWS = 4: east-moving (from lane 63) 1 of rows 0..3, 5 of rows 0..2, 8 of rows 1..3; west-moving (from lane 0) 3, 6, 7
    // ROWS = 2: east-moving 1 of rows 0, 1, 5 of row 0, 8 of row 1; west-moving 3 of rows 0, 1, 6 of row 0, 7 of row 1
    if (lane == 63) {
      float* e = edge[slot][wv][1];
      if constexpr (ROWS == 4) {
        e[0] = re[1].x; e[1] = ri[1].x; e[2] = ri[1].y; e[3] = re[1].y;
        e[4] = re[5].x; e[5] = ri[5].x; e[6] = ri[5].y; e[7] = ri[8].x;
        e[8] = ri[8].y; e[9] = re[8].y;
      } else {
        e[0] = re[1].x; e[1] = re[1].y; e[2] = re[5].x; e[3] = re[8].y;
      }
    }
    if (lane == 0) {
      float* e = edge[slot][wv][0];
      if constexpr (ROWS == 4) {
        e[0] = re[3].x; e[1] = ri[3].x; e[2] = ri[3].y; e[3] = re[3].y;
        e[4] = re[6].x; e[5] = ri[6].x; e[6] = ri[6].y; e[7] = ri[7].x;
        e[8] = ri[7].y; e[9] = re[7].y;
      } else {
        e[0] = re[3].x; e[1] = re[3].y; e[2] = re[6].x; e[3] = re[7].y;
      }
    }
    RESIDENT_PROF(0);  // wave-edge values into LDS
    __syncthreads();
    RESIDENT_PROF(1);  // barrier
    if (s > 0 && wave == 0) {
      // the per-wave sums of the previous step, written before this barrier: one partial per band and step
      const float v = row16_sum_dpp((lane < n_waves) ? wave_part[slot ^ 1][wv0 + lane] : 0.f);
      if (lane == 0) a.partials[(long)(s - 1) * bands + b] = v;
    }
    float W[NE], E[NE];
    {
      const float* w = edge[slot][west_wave][1];
      const float* e = edge[slot][east_wave][0];
#pragma unroll
      for (int j = 0; j < NE; j++) { W[j] = w[j]; E[j] = e[j]; }
    }
    // ---- the halo granules are asked for NOW, before anything else is computed: when the neighbours are not late
    // (the usual case: their edge rows were published about when ours were) the answer is there by the time the
    // interior pair is done, and the round trip of the load is hidden behind it
    const unsigned gs = from_south + (unsigned)slot * slot_bytes, gn = from_north + (unsigned)slot * slot_bytes;
    // the wave's first lane also needs the column west of it, its last lane the column east of it (every lane issues
    // the second load, the inner lanes for their own column again: four loads in flight, one wait, no divergence)
    const bool first = (lane == 0), last = (lane == 63);
    const unsigned x_side = (unsigned)(first ? xw : (last ? xe : x)) * 16u;
    granule_vec cs = granule_load(grsrc, gs + (unsigned)x * 16u);
    granule_vec cn = granule_load(grsrc, gn + (unsigned)x * 16u);
    granule_vec ss = granule_load(grsrc, gs + x_side);
    granule_vec sn = granule_load(grsrc, gn + x_side);

    const bool accel = (s + 1 < a.n_steps) || a.accel_last;
    // shifted populations (the value each cell receives from its west / east neighbour) and the streamed inputs of
    // the pair(s), as far as they come from inside the band
    f2 ti[kQ], te[kQ];
    if constexpr (ROWS == 4) {
      const float s1_0 = shift_from_west(re[1].x, W[0]), s1_1 = shift_from_west(ri[1].x, W[1]);
      const float s1_2 = shift_from_west(ri[1].y, W[2]), s1_3 = shift_from_west(re[1].y, W[3]);
      const float s5_0 = shift_from_west(re[5].x, W[4]), s5_1 = shift_from_west(ri[5].x, W[5]), s5_2 = shift_from_west(ri[5].y, W[6]);
      const float s8_1 = shift_from_west(ri[8].x, W[7]), s8_2 = shift_from_west(ri[8].y, W[8]), s8_3 = shift_from_west(re[8].y, W[9]);
      const float s3_0 = shift_from_east(re[3].x, E[0]), s3_1 = shift_from_east(ri[3].x, E[1]);
      const float s3_2 = shift_from_east(ri[3].y, E[2]), s3_3 = shift_from_east(re[3].y, E[3]);
      const float s6_0 = shift_from_east(re[6].x, E[4]), s6_1 = shift_from_east(ri[6].x, E[5]), s6_2 = shift_from_east(ri[6].y, E[6]);
      const float s7_1 = shift_from_east(ri[7].x, E[7]), s7_2 = shift_from_east(ri[7].y, E[8]), s7_3 = shift_from_east(re[7].y, E[9]);
      // interior pair: rows 1 and 2 pull from rows 0..3 of the band only
      ti[0] = ri[0];               ti[1] = f2{s1_1, s1_2};      ti[2] = f2{re[2].x, ri[2].x};
      ti[3] = f2{s3_1, s3_2};      ti[4] = f2{ri[4].y, re[4].y}; ti[5] = f2{s5_0, s5_1};
      ti[6] = f2{s6_0, s6_1};      ti[7] = f2{s7_2, s7_3};      ti[8] = f2{s8_2, s8_3};
      // edge pair: rows 0 and 3; the halves that come from the neighbouring bands are filled in below
      te[0] = re[0];               te[1] = f2{s1_0, s1_3};      te[2] = f2{0.f, ri[2].y};
      te[3] = f2{s3_0, s3_3};      te[4] = f2{ri[4].x, 0.f};     te[5] = f2{0.f, s5_2};
      te[6] = f2{0.f, s6_2};       te[7] = f2{s7_1, 0.f};       te[8] = f2{s8_1, 0.f};
    } else {
      const float s1_0 = shift_from_west(re[1].x, W[0]), s1_1 = shift_from_west(re[1].y, W[1]);
      const float s5_0 = shift_from_west(re[5].x, W[2]), s8_1 = shift_from_west(re[8].y, W[3]);
      const float s3_0 = shift_from_east(re[3].x, E[0]), s3_1 = shift_from_east(re[3].y, E[1]);
      const float s6_0 = shift_from_east(re[6].x, E[2]), s7_1 = shift_from_east(re[7].y, E[3]);
      te[0] = re[0];               te[1] = f2{s1_0, s1_1};      te[2] = f2{0.f, re[2].x};
      te[3] = f2{s3_0, s3_1};      te[4] = f2{re[4].y, 0.f};     te[5] = f2{0.f, s5_0};
      te[6] = f2{0.f, s6_0};       te[7] = f2{s7_1, 0.f};       te[8] = f2{s8_1, 0.f};
#pragma unroll
      for (int k = 0; k < kQ; k++) ti[k] = splat2(0.f);
    }
    f2 ni[kQ];
    float sum = 0.f;
    if constexpr (ROWS == 4 && !JOINT) sum = relax_pair_rows(ti, blocked_i, accel ? lid_i : 0u, a.omega, a.a1, a.a2, ni);

    RESIDENT_PROF(2);  // partial of the previous step, LDS edges read, shifts, halo loads issued, interior pair (ROWS 4, !JOINT)
    // ---- edge pair: rows 0 and TOP also pull from the neighbouring bands --------------------------------------
    {
      long long t_start = 0;
      for (unsigned spins = 0;; spins++) {
        const bool ok = ((unsigned)cs.w == tag) & ((unsigned)cn.w == tag) & ((unsigned)ss.w == tag) & ((unsigned)sn.w == tag);
#ifdef LBM_RESIDENT_PROFILE
        if (spins == 0) RESIDENT_PROF(3);  // first answer of the halo loads
        else prof_acc[7] += 1;
#endif
        if (__all(ok)) break;
        // not there yet: every so often look at the clock and at what the other workgroups say (wave-uniform)
        if ((spins & 63u) == 63u) {
          const long long now = wall_clock64();
          if (t_start == 0) t_start = now;
          const int st = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (st != 0 || now - t_start > a.timeout_ticks) {
            if (st == 0 && lane == 0) __hip_atomic_store(a.status, kResidentTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
            break;
          }
        }
        // how long to stay off the issue ports before asking again: short where the wave sits alone on its SIMD
        // (latency is everything), long where three other waves want the slots (a poll is ~25 instructions)
        // (a compile-time choice: measured over 0 ... 16 at every size the value hardly matters, and as a run-time switch
        // it cost a dozen scalar instructions and four branches per look)
        __builtin_amdgcn_s_sleep(MAXT > 512 ? 4 : 1);
        asm volatile("" ::: "memory");  // the loads must be issued again in every spin
        cs = granule_load(grsrc, gs + (unsigned)x * 16u);
        cn = granule_load(grsrc, gn + (unsigned)x * 16u);
        ss = granule_load(grsrc, gs + x_side);
        sn = granule_load(grsrc, gn + x_side);
      }
    }
    RESIDENT_PROF(4);  // further polls
    // south: {2, 5, 6} of its top row; north: {4, 7, 8} of its row 0; 5 and 8 come from the west column, 6 and 7 from the east
    const float side_s = __uint_as_float((unsigned)(first ? ss.y : ss.z)), side_n = __uint_as_float((unsigned)(first ? sn.z : sn.y));
    te[2].x = __uint_as_float((unsigned)cs.x);
    te[4].y = __uint_as_float((unsigned)cn.x);
    te[5].x = shift_from_west(__uint_as_float((unsigned)cs.y), side_s);
    te[6].x = shift_from_east(__uint_as_float((unsigned)cs.z), side_s);
    te[7].y = shift_from_east(__uint_as_float((unsigned)cn.y), side_n);
    te[8].y = shift_from_west(__uint_as_float((unsigned)cn.z), side_n);
    f2 ne[kQ];
    if constexpr (ROWS == 4 && JOINT) sum = relax_two_pairs_rows(ti, te, blocked_i, blocked_e, accel ? lid_i : 0u, accel ? lid_e : 0u, a.omega, a.a1, a.a2, ni, ne);
    else sum += relax_pair_rows(te, blocked_e, accel ? lid_e : 0u, a.omega, a.a1, a.a2, ne);
#pragma unroll
    for (int k = 0; k < kQ; k++) {
      re[k] = ne[k];
      if constexpr (ROWS == 4) ri[k] = ni[k];
    }
    RESIDENT_PROF(5);  // collision(s)
    // the neighbours wait for exactly these rows: out they go, before anything else
    if (s + 1 < a.n_steps && alive) publish(s + 1);
    // blocked cells report 0; sum over the wave, one partial per wave into LDS (summed after the next barrier)
    const float tot = wave_sum_dpp(sum);
    if (lane == 63) wave_part[slot][wv] = tot;
    RESIDENT_PROF(6);  // publish, wave sum
    // (a wave that gave up leaves the loop alone: the hardware barrier counts only waves that have not ended, and
    // the others find *status set in their next spin)
  }

#ifdef LBM_RESIDENT_PROFILE
  if (x == 0)
    for (int i = 0; i < 8; i++) a.prof[b * 8 + i] = prof_acc[i];
#endif
  __syncthreads();
  if (a.n_steps > 0 && wave == 0) {
    const float v = row16_sum_dpp((lane < n_waves) ? wave_part[(a.n_steps - 1) & 1][wv0 + lane] : 0.f);
    if (lane == 0) a.partials[(long)(a.n_steps - 1) * bands + b] = v;
  }
  float* out = a.dst + (long)(ROWS * b) * a.row_pitch + x;
#pragma unroll
  for (int k = 0; k < kQ; k++) {
    out[k * ps] = re[k].x;
    out[TOP * a.row_pitch + k * ps] = re[k].y;
    if constexpr (ROWS == 4) {
      out[1 * a.row_pitch + k * ps] = ri[k].x;
      out[2 * a.row_pitch + k * ps] = ri[k].y;
    }
  }
}

// resident kernel: sum the per-band partials of each step in a fixed order (double) -> tot_u[step_base + s]
__global__ __launch_bounds__(64) void reduce_band_partials(const float* partials, int bands, double* tot_u, int step_base) {
  const float* p = partials + (long)blockIdx.x * bands;
  double acc = 0.0;
  for (int i = threadIdx.x; i < bands; i += 64) acc += (double)p[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) tot_u[step_base + blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------
// fused step, 1 cell per lane: any nx (fallback for widths that are not a multiple of 4)
// ---------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ __launch_bounds__(kBlock) void step_scalar(const StepArgs a) {
  const long q = (long)blockIdx.x * kBlock + threadIdx.x;
  const long n_cells = (long)a.nx * a.n_rows;
  float my_sum = 0.f;
  if (q < n_cells) {
    const int rsel = (int)(q / a.nx);
    const int x = (int)(q - (long)rsel * a.nx);
    const int row = a.row_first + rsel * a.row_stride;
    const int xw = (x == 0) ? a.nx - 1 : x - 1;
    const int xe = (x + 1 == a.nx) ? 0 : x + 1;
    const long ps = a.plane_stride;
    const float* c_row = a.src + (long)row * a.row_pitch;
    const int rs = (row == 0 && a.wrap) ? a.rows - 1 : row - 1;
    const int rn = (row == a.rows - 1 && a.wrap) ? 0 : row + 1;
    const float* sb = a.src + (long)rs * a.row_pitch;
    const float* nb = a.src + (long)rn * a.row_pitch;
    const float *s2 = sb + 2 * ps, *s5 = sb + 5 * ps, *s6 = sb + 6 * ps;
    const float *n4 = nb + 4 * ps, *n7 = nb + 7 * ps, *n8 = nb + 8 * ps;
    float t[kQ] = {c_row[x], c_row[1 * ps + xw], s2[x], c_row[3 * ps + xe], n4[x],
                   s5[xw],   s6[xe],             n7[xe], n8[xw]};
    float r[kQ];
    if (a.mask[(long)row * a.pitch + x]) {
      bounce(t, r);
    } else {
      float speed;
      collide<EXACT>(t, a.omega, r, speed);
      my_sum = speed;
      if (row == a.accel_row) accelerate(r, a.a1, a.a2);
    }
    float* d = a.dst + (long)row * a.row_pitch + x;
#pragma unroll
    for (int k = 0; k < kQ; k++) d[k * ps] = r[k];
  }
  const float total = block_sum(my_sum);
  if (threadIdx.x == 0) a.partials[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// small kernels around the step
// ---------------------------------------------------------------------------------------------

// accelerate_flow() as its own pass (SerialCode/d2q9-bgk.c:216-246): used once before the first
// step of a run; later steps get it from the epilogue of the step kernel.
__global__ void accelerate_row(float* lat, const unsigned char* mask, long ps, long row_pitch,
                               int pitch, int nx, int row, float a1, float a2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  if (mask[(long)row * pitch + x]) return;
  const long c = (long)row * row_pitch + x;
  float f[kQ];
#pragma unroll
  for (int k = 0; k < kQ; k++) f[k] = lat[k * ps + c];
  accelerate(f, a1, a2);
  lat[1 * ps + c] = f[1];  lat[3 * ps + c] = f[3];  lat[5 * ps + c] = f[5];
  lat[6 * ps + c] = f[6];  lat[7 * ps + c] = f[7];  lat[8 * ps + c] = f[8];
}

// ---------------------------------------------------------------------------------------------
// "freshest available" halo mode (LBM_HALO_FRESHEST): the reference's MPI_Testall idea, MPI_Testall_OptimizedVersion/
// d2q9-bgk.c:279-290 -- look once whether this step's halo rows have arrived, never wait for them.  The rows travel
// into a staging row per side, followed in stream order by the id of the step they belong to (fresh_mark / a 4-byte
// copy); after the interior rows, fresh_decide looks at the ids ONCE, notes which sides made it (the engine's log of
// decisions) and fresh_adopt moves those staging rows into the halo rows, whole rows only.  The sides that did not make
// it keep the rows of the step before, which the one-pass-late exchange of the stale mode has put there.
// ---------------------------------------------------------------------------------------------
__global__ void fresh_mark(unsigned* a, unsigned* b, unsigned id) {
  if (a) __hip_atomic_store(a, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (b) __hip_atomic_store(b, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// tests (LBM_FRESH_TEST_DELAY_US): holds a comm stream back so that looks find nothing; bounded by the clock
__global__ void fresh_test_delay(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// arrived[0] / arrived[1]: id of the step whose south / north halo row the staging holds; mode 0: look (the product),
// 1: this pass's halo rows are fresh anyway (first pass of a call): note 3, adopt nothing
__global__ void fresh_decide(const unsigned* arrived, unsigned id, int mode, int* decision, unsigned char* log_entry) {
  unsigned d = 0;
  if (mode == 0) {
    d = (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == id ? 1u : 0u) |
        (__hip_atomic_load(arrived + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == id ? 2u : 0u);
    *decision = (int)d;
  } else {
    *decision = 0;
    d = 3;
  }
  *log_entry = (unsigned char)d;
}
// staging rows (written by a peer: read past the caches) -> halo rows, per side as decided
__global__ void fresh_adopt(const int* decision, const unsigned* stage_s, const unsigned* stage_n, unsigned* halo_s, unsigned* halo_n, long n) {
  const int d = *decision;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (d & 1) halo_s[i] = __hip_atomic_load(stage_s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (d & 2) halo_n[i] = __hip_atomic_load(stage_n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum the per-workgroup partials of up to gridDim.x steps: block s adds partials[s][0..n[s])
// in a fixed order (double accumulation) -> tot_u[step_base + s].  Deterministic, no atomics.
constexpr int kPartSlotsMax = 64;
struct SlotCounts {
  int n[kPartSlotsMax];  // valid partials in each slot (launch geometries differ between kernels)
};
// base_dev (optional): the first step's index is read from device memory, so that a captured hipGraph can be
// replayed for later steps (advance_counter moves it on)
__global__ __launch_bounds__(kBlock) void reduce_partials(const float* partials, SlotCounts counts,
                                                          long slot_stride, double* tot_u,
                                                          int step_base, const int* base_dev) {
  __shared__ double sh[kBlock];
  if (base_dev) step_base += *base_dev;
  const float* p = partials + (long)blockIdx.x * slot_stride;
  const int n_part = counts.n[blockIdx.x];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_part; i += kBlock) acc += (double)p[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) tot_u[step_base + blockIdx.x] = sh[0];
}

__global__ void set_counter(int* p, int value) { *p = value; }
__global__ void advance_counter(int* p, int by) { *p += by; }

// obstacle flags: the reference's int map (1 = blocked, SerialCode/d2q9-bgk.c:541, 570-601) -> uint8 rows of `pitch`
__global__ void mask_from_int(const int* src, unsigned char* dst, int nx, int pitch, int nrows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nx * nrows) return;
  const int r = (int)(i / nx), x = (int)(i - (long)r * nx);
  dst[(long)r * pitch + x] = src[i] ? 1 : 0;
}

// ... or a small tile repeated over the grid: cell (x, y) is blocked iff the tile's cell (x mod tnx, y mod tny) is
// (the synthetic 8192^2 / 16384^2 grids tile the reference's 1024^2 map).  Mask row r is
// global row row0 + r, folded periodically into [0, ny).
__global__ void mask_from_tile(const unsigned char* tile, int tnx, int tny, unsigned char* dst, int nx, int pitch,
                               int row0, int nrows, int ny) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nx * nrows) return;
  const int r = (int)(i / nx), x = (int)(i - (long)r * nx);
  int g = (row0 + r) % ny;
  if (g < 0) g += ny;
  dst[(long)r * pitch + x] = tile[(long)(g % tny) * tnx + (x % tnx)];
}

// number of blocked cells among n mask bytes (pitch padding is zero)
__global__ __launch_bounds__(256) void count_blocked(const unsigned char* mask, long n, unsigned long long* out) {
  __shared__ unsigned int sh[256];
  unsigned int acc = 0;
  const long base = ((long)blockIdx.x * 256 + threadIdx.x) * 16;
  for (int k = 0; k < 16; k++)
    if (base + k < n) acc += mask[base + k] ? 1u : 0u;
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicAdd(out, (unsigned long long)sh[0]);
}

// uniform equilibrium start (SerialCode/d2q9-bgk.c:546-567)
__global__ void init_equilibrium(float* lat, long ps, long row_pitch, int nx, int rows, float r0,
                                 float r1, float r2) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (long)nx * rows) return;
  const long row = j / nx;
  const long i = row * row_pitch + (j - row * nx);
  lat[i] = r0;
  lat[1 * ps + i] = r1;  lat[2 * ps + i] = r1;  lat[3 * ps + i] = r1;  lat[4 * ps + i] = r1;
  lat[5 * ps + i] = r2;  lat[6 * ps + i] = r2;  lat[7 * ps + i] = r2;  lat[8 * ps + i] = r2;
}

// AoS (reference host layout, 9 floats per cell) <-> SoA planes, rows [row0, row0+nrows)
__global__ void aos_to_soa(const float* aos, float* lat, long ps, long row_pitch, int nx, int row0,
                           int nrows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)nx * nrows * kQ;
  if (i >= n) return;
  const long cell = i / kQ;
  const int k = (int)(i - cell * kQ);
  const int r = (int)(cell / nx), x = (int)(cell - (long)r * nx);
  lat[k * ps + (long)(row0 + r) * row_pitch + x] = aos[i];
}

__global__ void soa_to_aos(const float* lat, float* aos, long ps, long row_pitch, int nx, int row0,
                           int nrows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)nx * nrows * kQ;
  if (i >= n) return;
  const long cell = i / kQ;
  const int k = (int)(i - cell * kQ);
  const int r = (int)(cell / nx), x = (int)(cell - (long)r * nx);
  aos[i] = lat[k * ps + (long)(row0 + r) * row_pitch + x];
}

// write_values() quantities (SerialCode/d2q9-bgk.c:684-719), always in the exact arithmetic
__global__ void final_state(const float* lat, const unsigned char* mask, long ps, long row_pitch,
                            int pitch, int nx, int row0, int nrows, float density, float* ux_o,
                            float* uy_o, float* um_o, float* pr_o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)nx * nrows;
  if (i >= n) return;
  const int r = (int)(i / nx), x = (int)(i - (long)r * nx);
  const long c = (long)(row0 + r) * row_pitch + x;
  if (mask[(long)(row0 + r) * pitch + x]) {
    ux_o[i] = 0.f;  uy_o[i] = 0.f;  um_o[i] = 0.f;
    pr_o[i] = density * kCsq;
  } else {
    float f[kQ];
#pragma unroll
    for (int k = 0; k < kQ; k++) f[k] = lat[k * ps + c];
    float rho, ux, uy;
    moments_exact(f, rho, ux, uy);
    ux_o[i] = ux;  uy_o[i] = uy;
    um_o[i] = sqrtf((ux * ux) + (uy * uy));
    pr_o[i] = rho * kCsq;
  }
}

// av_velocity() of a stored lattice (SerialCode/d2q9-bgk.c:409-458): per-workgroup partials of
// sum |u| in double; reduced by reduce_doubles.  Also serves total_density() (:644-660).
__global__ __launch_bounds__(kBlock) void lattice_sums(const float* lat, const unsigned char* mask,
                                                       long ps, long row_pitch, int pitch, int nx,
                                                       int rows, double* speed_part,
                                                       double* mass_part) {
  __shared__ double sh_s[kBlock], sh_m[kBlock];
  double s = 0.0, m = 0.0;
  const long n = (long)nx * rows;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
    const int r = (int)(i / nx), x = (int)(i - (long)r * nx);
    const long c = (long)r * row_pitch + x;
    float f[kQ];
#pragma unroll
    for (int k = 0; k < kQ; k++) f[k] = lat[k * ps + c];
    float rho, ux, uy;
    moments_exact(f, rho, ux, uy);
    m += (double)rho;
    if (!mask[(long)r * pitch + x]) s += (double)sqrtf((ux * ux) + (uy * uy));
  }
  sh_s[threadIdx.x] = s;  sh_m[threadIdx.x] = m;
  __syncthreads();
  for (int k = kBlock / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + k];
      sh_m[threadIdx.x] += sh_m[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    speed_part[blockIdx.x] = sh_s[0];
    mass_part[blockIdx.x] = sh_m[0];
  }
}

}  // namespace lbm
