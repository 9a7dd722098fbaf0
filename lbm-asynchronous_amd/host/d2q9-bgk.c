/*
 * d2q9-bgk.c -- the reference's command line on the MI355X engine (plain C99 host program).
 *
 *   ./d2q9-bgk <paramfile> <obstaclefile>
 *
 * Same arguments, same final_state.dat / av_vels.dat, same stdout block as
 * /root/reference/SerialCode/d2q9-bgk.c (main() :132-205).  The timestep loop (:166-170) is one
 * call into the C-ABI engine (include/lbm_hip.h); everything numerical happens on the GPU.
 *
 * Optional environment (the two-argument form stays valid):
 *   LBM_GPUS=<n>          row slabs / GPUs in this process (default 1)
 *   LBM_MATH=exact|fast   collision arithmetic (default exact: bit-identical to SerialCode)
 *   LBM_TILE=<tx>x<ty>    the obstacle file describes a tx x ty tile that is repeated over the
 *                         nx x ny grid of the parameter file (synthetic 8192^2 / 16384^2 grids)
 *   LBM_OUTPUT=text|none  write final_state.dat / av_vels.dat (default) or skip final_state.dat
 *   LBM_PRESSURE_BIN=<f>  additionally dump the fp32 pressure field (ny*nx floats) to <f>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "lbm_io.h"

static double wall_seconds(void)
{
  struct timeval t;
  gettimeofday(&t, NULL);
  return t.tv_sec + (t.tv_usec / 1000000.0);
}

int main(int argc, char* argv[])
{
  if (argc != 3) lbm_usage(argv[0]);
  const char* paramfile = argv[1];
  const char* obstaclefile = argv[2];

  const char* env;
  int n_gpus = 1;
  if ((env = getenv("LBM_GPUS")) && *env) n_gpus = atoi(env);
  int math_mode = LBM_MATH_EXACT;
  if ((env = getenv("LBM_MATH")) && !strcmp(env, "fast")) math_mode = LBM_MATH_FAST;
  int write_text = 1;
  if ((env = getenv("LBM_OUTPUT")) && !strcmp(env, "none")) write_text = 0;

  /* Total/init time starts here (SerialCode/d2q9-bgk.c:156-159) */
  const double tot_tic = wall_seconds();
  lbm_params params;
  lbm_read_params(paramfile, &params);

  /* uniform equilibrium start is generated on the device (cells_aos == NULL); with LBM_TILE the mask is
   * expanded on the device from the tile, and the global map is only built if final_state.dat is written
   * (its last column is the obstacle flag, SerialCode/d2q9-bgk.c:722) */
  int* obstacles = NULL;
  int tile_nx = 0, tile_ny = 0;
  lbm_ctx* ctx;
  if ((env = getenv("LBM_TILE")) && sscanf(env, "%dx%d", &tile_nx, &tile_ny) == 2) {
    lbm_params tile_params = params;
    tile_params.nx = tile_nx;
    tile_params.ny = tile_ny;
    int* tile = lbm_read_obstacles(obstaclefile, &tile_params);
    ctx = lbm_create_tiled(&params, tile, tile_nx, tile_ny, NULL, n_gpus, math_mode);
    if (write_text) obstacles = lbm_tile_obstacles(tile, tile_nx, tile_ny, params.nx, params.ny);
    free(tile);
  } else {
    obstacles = lbm_read_obstacles(obstaclefile, &params);
    ctx = lbm_create(&params, obstacles, NULL, n_gpus, math_mode);
  }
  if (n_gpus > 1) {
    /* the MPI programs announce their ranks ("Process %d of %d started.", MPI/d2q9-bgk.c:151): here, what the halo
       transport is -- which RCCL, and how many ranks its communicators count */
    lbm_rccl_status st;
    lbm_info info;
    if (lbm_rccl_info(ctx, &st) == LBM_SUCCESS && lbm_get_info(ctx, &info) == LBM_SUCCESS) {
      if (st.n_comms > 0)
        fprintf(stderr, "%d row slabs on %d device(s): RCCL %d.%d.%d (%s), %d communicators of %d ranks\n", info.n_slabs,
                lbm_device_count() < n_gpus ? lbm_device_count() : n_gpus, st.version / 10000, st.version / 100 % 100,
                st.version % 100, st.library, st.n_comms, st.nranks);
      else
        fprintf(stderr, "%d row slabs on %d device(s): halo rows by device copies\n", info.n_slabs,
                lbm_device_count() < n_gpus ? lbm_device_count() : n_gpus);
    }
  }
  lbm_sync(ctx);
  const double init_toc = wall_seconds();

  /* Compute time: the whole timestep loop (:166-170) */
  lbm_run(ctx, params.max_iters);
  lbm_sync(ctx);
  const double comp_toc = wall_seconds();

  /* Collate: bring the results back to the host */
  const size_t n_cells = (size_t)params.nx * (size_t)params.ny;
  float* av_vels = (float*)malloc(sizeof(float) * (size_t)(params.max_iters > 0 ? params.max_iters : 1));
  if (av_vels == NULL) lbm_die("cannot allocate memory for av_vels", __LINE__, __FILE__);
  lbm_read_av_vels(ctx, av_vels, params.max_iters);
  float reynolds = 0.f;
  lbm_calc_reynolds(ctx, &reynolds);
  float* fields = NULL;
  const int want_fields = write_text || getenv("LBM_PRESSURE_BIN");
  if (want_fields) {
    fields = (float*)malloc(sizeof(float) * 4 * n_cells);
    if (fields == NULL) lbm_die("cannot allocate memory for cells", __LINE__, __FILE__);
    lbm_read_final_state(ctx, fields, fields + n_cells, fields + 2 * n_cells, fields + 3 * n_cells);
  }
  const double col_toc = wall_seconds();

  /* the reference's report (:195-200) */
  printf("==done==\n");
  printf("Reynolds number:\t\t%.12E\n", reynolds);
  printf("Elapsed Init time:\t\t\t%.6lf (s)\n", init_toc - tot_tic);
  printf("Elapsed Compute time:\t\t\t%.6lf (s)\n", comp_toc - init_toc);
  printf("Elapsed Collate time:\t\t\t%.6lf (s)\n", col_toc - comp_toc);
  printf("Elapsed Total time:\t\t\t%.6lf (s)\n", col_toc - tot_tic);

  if (write_text) {
    FILE* fp = fopen(LBM_FINALSTATEFILE, "w");
    if (fp == NULL) lbm_die("could not open file output file", __LINE__, __FILE__);
    static char iobuf[1 << 22];
    setvbuf(fp, iobuf, _IOFBF, sizeof(iobuf));
    lbm_write_final_state_rows(fp, &params, 0, params.ny, fields, fields + n_cells, fields + 2 * n_cells,
                               fields + 3 * n_cells, obstacles);
    fclose(fp);
  }
  lbm_write_av_vels(LBM_AVVELSFILE, av_vels, params.max_iters);
  if ((env = getenv("LBM_PRESSURE_BIN")) && *env) {
    FILE* fp = fopen(env, "wb");
    if (fp == NULL) lbm_die("could not open file output file", __LINE__, __FILE__);
    fwrite(fields + 3 * n_cells, sizeof(float), n_cells, fp);
    fclose(fp);
  }

  lbm_destroy(ctx);
  free(fields);
  free(av_vels);
  free(obstacles);
  return EXIT_SUCCESS;
}
