/*
 * lbm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, see lbm_oracle.h).
 *
 * Restates the D2Q9-BGK timestep of /root/reference/SerialCode/d2q9-bgk.c in this repo's own
 * code.  fp32 throughout, the reference's operation order, no FMA contraction.
 *
 * Speed numbering (SerialCode/d2q9-bgk.c:9-15):   6 2 5
 *                                                  3 0 1
 *                                                  7 4 8
 */
#include "lbm_oracle.h"

#include <math.h>
#include <stddef.h>

enum { Q = LBM_ORACLE_Q };

/* lattice velocity of speed k, and the speed that points the opposite way */
static const int CX[Q]  = { 0, 1, 0, -1, 0, 1, -1, -1, 1 };
static const int CY[Q]  = { 0, 0, 1, 0, -1, 1, 1, -1, -1 };
static const int OPP[Q] = { 0, 3, 4, 1, 2, 7, 8, 5, 6 };

/* ------------------------------------------------------------------------------------------
 * per-cell arithmetic
 * ---------------------------------------------------------------------------------------- */

/* density and velocity of one cell; the expression trees of SerialCode/d2q9-bgk.c:325-347
 * (identical in av_velocity :426-448 and write_values :692-714) */
static inline void cell_moments(const float f[Q], float* rho, float* ux, float* uy)
{
  float d = 0.f;
  for (int k = 0; k < Q; k++) d += f[k];
  *rho = d;
  *ux = (f[1] + f[5] + f[8] - (f[3] + f[6] + f[7])) / d;
  *uy = (f[2] + f[5] + f[6] - (f[4] + f[7] + f[8])) / d;
}

/* one equilibrium population: w*rho * (1 + u/c^2 + u^2/(2c^4) - |u|^2/(2c^2)),
 * evaluated left to right exactly as SerialCode/d2q9-bgk.c:369-393 */
static inline float equilibrium(float w_rho, float u, float usq_over_2csq)
{
  const float c_sq        = 1.f / 3.f;
  const float two_c_sq_sq = 2.f * c_sq * c_sq;
  return w_rho * (1.f + u / c_sq + (u * u) / two_c_sq_sq - usq_over_2csq);
}

/* BGK relaxation of one cell: t = streamed populations, out = relaxed ones.
 * SerialCode/d2q9-bgk.c:306-407 (collision body). */
static inline void bgk_collide(const float t[Q], float omega, float out[Q])
{
  const float c_sq = 1.f / 3.f;
  const float w0 = 4.f / 9.f, w1 = 1.f / 9.f, w2 = 1.f / 36.f;
  float rho, ux, uy;
  cell_moments(t, &rho, &ux, &uy);

  const float u_sq = ux * ux + uy * uy;
  const float usq_term = u_sq / (2.f * c_sq);

  float u[Q];
  u[1] = ux;        u[2] = uy;
  u[3] = -ux;       u[4] = -uy;
  u[5] = ux + uy;   u[6] = -ux + uy;
  u[7] = -ux - uy;  u[8] = ux - uy;

  float eq[Q];
  eq[0] = w0 * rho * (1.f - usq_term);
  for (int k = 1; k <= 4; k++) eq[k] = equilibrium(w1 * rho, u[k], usq_term);
  for (int k = 5; k <= 8; k++) eq[k] = equilibrium(w2 * rho, u[k], usq_term);

  for (int k = 0; k < Q; k++) out[k] = t[k] + omega * (eq[k] - t[k]);
}

/* |u| of one cell, SerialCode/d2q9-bgk.c:426-450 */
static inline float cell_speed(const float f[Q])
{
  float rho, ux, uy;
  cell_moments(f, &rho, &ux, &uy);
  return sqrtf((ux * ux) + (uy * uy));
}

/* the accelerate_flow() update of one cell, SerialCode/d2q9-bgk.c:229-242.
 * f1,f3,f5,f6,f7,f8 point at the six affected populations. */
static inline void accelerate_cell(float a1, float a2, float* f1, float* f3, float* f5,
                                   float* f6, float* f7, float* f8)
{
  if ((*f3 - a1) > 0.f && (*f6 - a2) > 0.f && (*f7 - a2) > 0.f) {
    *f1 += a1;  *f5 += a2;  *f8 += a2;
    *f3 -= a1;  *f6 -= a2;  *f7 -= a2;
  }
}

/* ------------------------------------------------------------------------------------------
 * AoS four-sweep form (the serial reference's structure)
 * ---------------------------------------------------------------------------------------- */

void lbm_oracle_init_cells(const lbm_oracle_params* p, float* cells)
{
  /* SerialCode/d2q9-bgk.c:546-548 */
  const float r0 = p->density * 4.f / 9.f;
  const float r1 = p->density / 9.f;
  const float r2 = p->density / 36.f;
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  for (size_t c = 0; c < n; c++) {
    float* f = cells + Q * c;
    f[0] = r0;
    f[1] = f[2] = f[3] = f[4] = r1;
    f[5] = f[6] = f[7] = f[8] = r2;
  }
}

void lbm_oracle_accelerate_flow(const lbm_oracle_params* p, float* cells, const int* obstacles)
{
  /* SerialCode/d2q9-bgk.c:219-223 */
  const float a1 = p->density * p->accel / 9.f;
  const float a2 = p->density * p->accel / 36.f;
  const size_t row = (size_t)(p->ny - 2) * (size_t)p->nx;
  for (int x = 0; x < p->nx; x++) {
    if (obstacles[row + x]) continue;
    float* f = cells + Q * (row + x);
    accelerate_cell(a1, a2, &f[1], &f[3], &f[5], &f[6], &f[7], &f[8]);
  }
}

void lbm_oracle_propagate(const lbm_oracle_params* p, const float* cells, float* tmp_cells)
{
  /* periodic pull, SerialCode/d2q9-bgk.c:251-273: speed k arrives from (x - cx_k, y - cy_k).
   * Row pointers for the three source rows, column offsets for the three source columns. */
  const int nx = p->nx, ny = p->ny;
  for (int y = 0; y < ny; y++) {
    const float* row_n = cells + (size_t)Q * nx * ((y + 1) % ny);          /* cy = -1 pulls from the north */
    const float* row_c = cells + (size_t)Q * nx * y;
    const float* row_s = cells + (size_t)Q * nx * ((y == 0) ? ny - 1 : y - 1); /* cy = +1 from the south */
    float* out = tmp_cells + (size_t)Q * nx * y;
    for (int x = 0; x < nx; x++) {
      const size_t e = (size_t)Q * ((x + 1) % nx);            /* cx = -1 pulls from the east cell */
      const size_t c = (size_t)Q * x;
      const size_t w = (size_t)Q * ((x == 0) ? nx - 1 : x - 1); /* cx = +1 from the west cell */
      float* o = out + c;
      o[0] = row_c[c + 0];
      o[1] = row_c[w + 1];
      o[2] = row_s[c + 2];
      o[3] = row_c[e + 3];
      o[4] = row_n[c + 4];
      o[5] = row_s[w + 5];
      o[6] = row_s[e + 6];
      o[7] = row_n[e + 7];
      o[8] = row_n[w + 8];
    }
  }
}

void lbm_oracle_rebound(const lbm_oracle_params* p, float* cells, const float* tmp_cells,
                        const int* obstacles)
{
  /* SerialCode/d2q9-bgk.c:282-301: mirrored copy on blocked cells, speed 0 untouched */
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  for (size_t c = 0; c < n; c++) {
    if (!obstacles[c]) continue;
    for (int k = 1; k < Q; k++) cells[Q * c + k] = tmp_cells[Q * c + OPP[k]];
  }
}

void lbm_oracle_collision(const lbm_oracle_params* p, float* cells, const float* tmp_cells,
                          const int* obstacles)
{
  /* SerialCode/d2q9-bgk.c:317-404 */
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  for (size_t c = 0; c < n; c++) {
    if (obstacles[c]) continue;
    bgk_collide(tmp_cells + Q * c, p->omega, cells + Q * c);
  }
}

void lbm_oracle_timestep(const lbm_oracle_params* p, float* cells, float* tmp_cells,
                         const int* obstacles)
{
  /* SerialCode/d2q9-bgk.c:207-214 */
  lbm_oracle_accelerate_flow(p, cells, obstacles);
  lbm_oracle_propagate(p, cells, tmp_cells);
  lbm_oracle_rebound(p, cells, tmp_cells, obstacles);
  lbm_oracle_collision(p, cells, tmp_cells, obstacles);
}

float lbm_oracle_sum_velocity(const lbm_oracle_params* p, const float* cells, const int* obstacles,
                              int* fluid_cells)
{
  /* SerialCode/d2q9-bgk.c:411-455 */
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  float tot_u = 0.f;
  int count = 0;
  for (size_t c = 0; c < n; c++) {
    if (obstacles[c]) continue;
    tot_u += cell_speed(cells + Q * c);
    ++count;
  }
  if (fluid_cells) *fluid_cells = count;
  return tot_u;
}

float lbm_oracle_av_velocity(const lbm_oracle_params* p, const float* cells, const int* obstacles)
{
  int count = 0;
  const float tot_u = lbm_oracle_sum_velocity(p, cells, obstacles, &count);
  return tot_u / (float)count; /* SerialCode/d2q9-bgk.c:457 */
}

void lbm_oracle_run(const lbm_oracle_params* p, float* cells, float* tmp_cells,
                    const int* obstacles, float* av_vels, int n_steps)
{
  /* SerialCode/d2q9-bgk.c:166-170 */
  for (int tt = 0; tt < n_steps; tt++) {
    lbm_oracle_timestep(p, cells, tmp_cells, obstacles);
    av_vels[tt] = lbm_oracle_av_velocity(p, cells, obstacles);
  }
}

float lbm_oracle_calc_reynolds(const lbm_oracle_params* p, const float* cells, const int* obstacles)
{
  /* SerialCode/d2q9-bgk.c:639-641 */
  const float viscosity = 1.f / 6.f * (2.f / p->omega - 1.f);
  return lbm_oracle_av_velocity(p, cells, obstacles) * p->reynolds_dim / viscosity;
}

float lbm_oracle_total_density(const lbm_oracle_params* p, const float* cells)
{
  /* SerialCode/d2q9-bgk.c:646-659 */
  const size_t n = (size_t)p->nx * (size_t)p->ny * Q;
  float total = 0.f;
  for (size_t i = 0; i < n; i++) total += cells[i];
  return total;
}

void lbm_oracle_final_state(const lbm_oracle_params* p, const float* cells, const int* obstacles,
                            float* u_x, float* u_y, float* u_mag, float* pressure)
{
  /* SerialCode/d2q9-bgk.c:679-719 */
  const float c_sq = 1.f / 3.f;
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  for (size_t c = 0; c < n; c++) {
    if (obstacles[c]) {
      u_x[c] = u_y[c] = u_mag[c] = 0.f;
      pressure[c] = p->density * c_sq;
    } else {
      float rho, ux, uy;
      cell_moments(cells + Q * c, &rho, &ux, &uy);
      u_x[c] = ux;
      u_y[c] = uy;
      u_mag[c] = sqrtf((ux * ux) + (uy * uy));
      pressure[c] = rho * c_sq;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * SoA two-lattice fused pull form (OpenMP/d2q9-bgk.c:260-498, MPI_Waitall/d2q9-bgk.c:352-555)
 * ---------------------------------------------------------------------------------------- */

/* Advance one row.  south/centre/north point at the first cell of plane 0 of the three source
 * rows; plane k is +k*ps floats further.  Returns the row's fp32 sum of |u| over fluid cells. */
static float fused_row(int nx, float omega, const float* south, const float* centre,
                       const float* north, float* out, long ps, const int* obstacle_row)
{
  const float* rowsrc[3] = { north, centre, south }; /* index cy+1: cy=-1 pulls from north */
  float row_sum = 0.f;
  for (int x = 0; x < nx; x++) {
    int xsrc[3];
    xsrc[0] = (x + 1 == nx) ? 0 : x + 1;
    xsrc[1] = x;
    xsrc[2] = (x == 0) ? nx - 1 : x - 1;
    float t[Q];
    for (int k = 0; k < Q; k++) t[k] = rowsrc[CY[k] + 1][(long)k * ps + xsrc[CX[k] + 1]];
    if (obstacle_row[x]) {
      out[x] = t[0];
      for (int k = 1; k < Q; k++) out[(long)k * ps + x] = t[OPP[k]];
    } else {
      float r[Q];
      bgk_collide(t, omega, r);
      for (int k = 0; k < Q; k++) out[(long)k * ps + x] = r[k];
      row_sum += cell_speed(r);
    }
  }
  return row_sum;
}

void lbm_oracle_accelerate_row_soa(int nx, float density, float accel, float* planes,
                                   long ps, const int* obstacle_row, int slab_row)
{
  const float a1 = density * accel / 9.f;
  const float a2 = density * accel / 36.f;
  float* r = planes + (long)slab_row * nx;
  for (int x = 0; x < nx; x++) {
    if (obstacle_row[x]) continue;
    accelerate_cell(a1, a2, &r[1 * ps + x], &r[3 * ps + x], &r[5 * ps + x], &r[6 * ps + x],
                    &r[7 * ps + x], &r[8 * ps + x]);
  }
}

float lbm_oracle_fused_rows(int nx, int rows, float density, float accel, float omega,
                            float* src, float* dst, long ps,
                            const int* obstacles, int accel_row, int row_first, int row_last)
{
  (void)rows;
  if (accel_row > 0)
    lbm_oracle_accelerate_row_soa(nx, density, accel, src, ps,
                                  obstacles + (long)(accel_row - 1) * nx, accel_row);
  float tot = 0.f;
  for (int r = row_first; r <= row_last; r++) {
    const float* c = src + (long)r * nx;
    tot += fused_row(nx, omega, c - nx, c, c + nx, dst + (long)r * nx, ps,
                     obstacles + (long)(r - 1) * nx);
  }
  return tot;
}

float lbm_oracle_fused_step_periodic(const lbm_oracle_params* p, float* src, float* dst,
                                     const int* obstacles)
{
  const int nx = p->nx, ny = p->ny;
  const long ps = (long)nx * ny;
  lbm_oracle_accelerate_row_soa(nx, p->density, p->accel, src, ps,
                                obstacles + (long)(ny - 2) * nx, ny - 2);
  float tot = 0.f;
#pragma omp parallel for reduction(+ : tot) schedule(static)
  for (int y = 0; y < ny; y++) {
    const int ys = (y == 0) ? ny - 1 : y - 1;
    const int yn = (y + 1 == ny) ? 0 : y + 1;
    tot += fused_row(nx, p->omega, src + (long)ys * nx, src + (long)y * nx, src + (long)yn * nx,
                     dst + (long)y * nx, ps, obstacles + (long)y * nx);
  }
  return tot;
}

void lbm_oracle_aos_to_soa(int n_cells, const float* aos, float* soa, long ps)
{
  for (long c = 0; c < n_cells; c++)
    for (int k = 0; k < Q; k++) soa[(long)k * ps + c] = aos[Q * c + k];
}

void lbm_oracle_soa_to_aos(int n_cells, const float* soa, long ps, float* aos)
{
  for (long c = 0; c < n_cells; c++)
    for (int k = 0; k < Q; k++) aos[Q * c + k] = soa[(long)k * ps + c];
}
