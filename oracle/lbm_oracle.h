/*
 * lbm_oracle.h -- CPU ORACLE for the D2Q9-BGK timestep hot path.
 *
 * *** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load, link or run
 * anything under oracle/.  The product path (include/lbm_hip.h -> liblbm_hip.so) never does and
 * has no CPU fallback.
 *
 * This is a fresh restatement (not a copy) of the algorithm of the reference program
 * /root/reference/SerialCode/d2q9-bgk.c.  Every function names the reference lines it follows.
 * Arithmetic is fp32 with the reference's operation order and NO fused multiply-add
 * (the reference builds with -std=c99, under which GCC never contracts a*b+c), so that the
 * final lattice is bit-identical to the reference binary's; oracle/Makefile builds that binary
 * into oracle/_ref/ and tests/test_oracle_vs_ref.py pins the equality.
 *
 * PARITY PINNED BY: (1) oracle/_ref/d2q9-bgk-serial (the reference itself, compiled here)
 * producing byte-identical final_state.dat on the reference's four data sets; (2) the
 * reference's double-precision goldens check/ *.dat through the check.py rule (<= 1 %);
 * (3) the known-answer table of SURVEY.md section 8c.  See tests/golden/README.md.
 */
#ifndef LBM_ORACLE_H
#define LBM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ORACLE_Q 9

/* Run constants; field-for-field the reference's t_param (SerialCode/d2q9-bgk.c:66-75). */
typedef struct {
  int   nx;
  int   ny;
  int   max_iters;
  int   reynolds_dim;
  float density;
  float accel;
  float omega;
} lbm_oracle_params;

/* ---- Array-of-structures lattice: cell c = ii + jj*nx holds 9 floats at cells[9*c + k] ----
 * (same memory image as the reference's t_speed array, SerialCode/d2q9-bgk.c:78-81). */

/* uniform equilibrium start, SerialCode/d2q9-bgk.c:546-567 */
void  lbm_oracle_init_cells(const lbm_oracle_params* p, float* cells);

/* the four sweeps of timestep(), SerialCode/d2q9-bgk.c:207-407 */
void  lbm_oracle_accelerate_flow(const lbm_oracle_params* p, float* cells, const int* obstacles);
void  lbm_oracle_propagate(const lbm_oracle_params* p, const float* cells, float* tmp_cells);
void  lbm_oracle_rebound(const lbm_oracle_params* p, float* cells, const float* tmp_cells,
                         const int* obstacles);
void  lbm_oracle_collision(const lbm_oracle_params* p, float* cells, const float* tmp_cells,
                           const int* obstacles);
void  lbm_oracle_timestep(const lbm_oracle_params* p, float* cells, float* tmp_cells,
                          const int* obstacles);

/* av_velocity(), SerialCode/d2q9-bgk.c:409-458: sequential fp32 sum / (float)count */
float lbm_oracle_av_velocity(const lbm_oracle_params* p, const float* cells, const int* obstacles);
/* same sweep, but returns the raw sum and the fluid-cell count (MPI variants keep the sum,
 * MPI_Waitall/d2q9-bgk.c:256, 321-327) */
float lbm_oracle_sum_velocity(const lbm_oracle_params* p, const float* cells, const int* obstacles,
                              int* fluid_cells);

/* the driver loop, SerialCode/d2q9-bgk.c:166-170: n_steps x (timestep, av_velocity) */
void  lbm_oracle_run(const lbm_oracle_params* p, float* cells, float* tmp_cells,
                     const int* obstacles, float* av_vels, int n_steps);

/* calc_reynolds(), SerialCode/d2q9-bgk.c:637-642; total_density(), :644-660 */
float lbm_oracle_calc_reynolds(const lbm_oracle_params* p, const float* cells, const int* obstacles);
float lbm_oracle_total_density(const lbm_oracle_params* p, const float* cells);

/* per-cell output quantities of write_values(), SerialCode/d2q9-bgk.c:679-719
 * (u_x, u_y, |u|, pressure); each output is nx*ny floats, row-major */
void  lbm_oracle_final_state(const lbm_oracle_params* p, const float* cells, const int* obstacles,
                             float* u_x, float* u_y, float* u_mag, float* pressure);

/* ---- Structure-of-arrays, two-lattice, fused pull form ----
 * The reference's own fast formulation: fusion_more(), OpenMP/d2q9-bgk.c:260-498, and its
 * row-range / halo-padded slab form, MPI_Waitall/d2q9-bgk.c:352-555.
 *
 * A slab is (rows + 2) x nx: row 0 and row rows+1 are halo rows, rows 1..rows are owned.
 * Plane k of the slab is src[k*plane_stride ...].  The obstacle mask has NO halo rows
 * (owned row r -> mask row r-1), as MPI_Waitall/d2q9-bgk.c:371,424.
 * x wraps periodically; y does NOT wrap inside a slab (the halo rows carry the neighbours).
 *
 * accel_row: owned-row index (1-based slab row) to accelerate before streaming, or 0 for none
 *            (the global row ny-2, SerialCode/d2q9-bgk.c:223, lives in exactly one slab).
 *            Acceleration is applied in place to src, as the reference does.
 * Rows row_first..row_last (inclusive, 1-based slab rows) are advanced from src into dst.
 * Returns the fp32 sum of |u| over the fluid cells of those rows (row by row, left to right).
 */
float lbm_oracle_fused_rows(int nx, int rows, float density, float accel, float omega,
                            float* src, float* dst, long plane_stride,
                            const int* obstacles, int accel_row, int row_first, int row_last);

/* in-place acceleration of one slab row of an SoA slab (the accelerate part of fusion_more,
 * OpenMP/d2q9-bgk.c:293-321) */
void  lbm_oracle_accelerate_row_soa(int nx, float density, float accel, float* planes,
                                    long plane_stride, const int* obstacle_row, int slab_row);

/* multi-threaded whole-grid fused step on a periodic SoA lattice (no halo rows): the
 * CPU multi-core baseline, same formulation as OpenMP/d2q9-bgk.c:334.  Returns sum |u|. */
float lbm_oracle_fused_step_periodic(const lbm_oracle_params* p, float* src, float* dst,
                                     const int* obstacles);

/* AoS <-> SoA transposes (no arithmetic) */
void  lbm_oracle_aos_to_soa(int n_cells, const float* aos, float* soa, long plane_stride);
void  lbm_oracle_soa_to_aos(int n_cells, const float* soa, long plane_stride, float* aos);

#ifdef __cplusplus
}
#endif
#endif /* LBM_ORACLE_H */
