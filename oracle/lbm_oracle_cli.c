/*
 * lbm_oracle_cli.c -- TEST INFRASTRUCTURE: the d2q9-bgk command line on the CPU oracle.
 *
 *   lbm_oracle_cli <paramfile> <obstaclefile> [steps]
 *
 * Used to (a) pin the oracle byte-for-byte against the reference binary oracle/_ref/d2q9-bgk-serial
 * and (b) as the CPU baseline ("port") beside the GPU numbers.  It reuses the product's file
 * I/O (lbm-asynchronous_amd/host/lbm_io.c) so those readers/writers are checked against the
 * reference's formats too; the product never links anything from oracle/.
 *
 * Environment: LBM_ORACLE_FORM=serial (default: four-sweep AoS, single thread, the shape of
 * SerialCode) | fused (two-lattice SoA pull, OpenMP threads, the shape of OpenMP/d2q9-bgk.c);
 * LBM_TILE=<tx>x<ty> as in the product CLI; LBM_OUTPUT=text|none; LBM_PRESSURE_BIN=<file>.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "../lbm-asynchronous_amd/host/lbm_io.h"
#include "lbm_oracle.h"

static double wall_seconds(void)
{
  struct timeval t;
  gettimeofday(&t, NULL);
  return t.tv_sec + (t.tv_usec / 1000000.0);
}

int main(int argc, char* argv[])
{
  if (argc != 3 && argc != 4) lbm_usage(argv[0]);
  const double tot_tic = wall_seconds();
  lbm_params hp;
  lbm_read_params(argv[1], &hp);
  if (argc == 4) hp.max_iters = atoi(argv[3]);
  const char* env;
  int* obstacles;
  int tx = 0, ty = 0;
  if ((env = getenv("LBM_TILE")) && sscanf(env, "%dx%d", &tx, &ty) == 2) {
    lbm_params tp = hp;
    tp.nx = tx;
    tp.ny = ty;
    int* tile = lbm_read_obstacles(argv[2], &tp);
    obstacles = lbm_tile_obstacles(tile, tx, ty, hp.nx, hp.ny);
    free(tile);
  } else {
    obstacles = lbm_read_obstacles(argv[2], &hp);
  }
  const int fused = (env = getenv("LBM_ORACLE_FORM")) && !strcmp(env, "fused");
  const int write_text = !((env = getenv("LBM_OUTPUT")) && !strcmp(env, "none"));

  lbm_oracle_params p = { hp.nx, hp.ny, hp.max_iters, hp.reynolds_dim, hp.density, hp.accel, hp.omega };
  const size_t n = (size_t)p.nx * p.ny;
  float* cells = (float*)malloc(sizeof(float) * 9 * n);
  float* tmp = (float*)malloc(sizeof(float) * 9 * n);
  float* av_vels = (float*)malloc(sizeof(float) * (size_t)(p.max_iters > 0 ? p.max_iters : 1));
  if (!cells || !tmp || !av_vels) lbm_die("cannot allocate memory for cells", __LINE__, __FILE__);
  lbm_oracle_init_cells(&p, cells);
  const double init_toc = wall_seconds();

  if (!fused) {
    lbm_oracle_run(&p, cells, tmp, obstacles, av_vels, p.max_iters);
  } else {
    int fluid = 0;
    for (size_t c = 0; c < n; c++) fluid += !obstacles[c];
    float* a = (float*)malloc(sizeof(float) * 9 * n);
    if (!a) lbm_die("cannot allocate memory for cells", __LINE__, __FILE__);
    lbm_oracle_aos_to_soa((int)n, cells, a, (long)n);
    float* src = a;
    float* dst = tmp;
    for (int tt = 0; tt < p.max_iters; tt++) {
      av_vels[tt] = lbm_oracle_fused_step_periodic(&p, src, dst, obstacles) / (float)fluid;
      float* s = src; src = dst; dst = s;
    }
    lbm_oracle_soa_to_aos((int)n, src, (long)n, cells);
    free(a);
  }
  const double comp_toc = wall_seconds();

  printf("==done==\n");
  printf("Reynolds number:\t\t%.12E\n", lbm_oracle_calc_reynolds(&p, cells, obstacles));
  printf("Elapsed Init time:\t\t\t%.6lf (s)\n", init_toc - tot_tic);
  printf("Elapsed Compute time:\t\t\t%.6lf (s)\n", comp_toc - init_toc);
  printf("Elapsed Collate time:\t\t\t%.6lf (s)\n", 0.0);
  printf("Elapsed Total time:\t\t\t%.6lf (s)\n", comp_toc - tot_tic);

  float* f = (float*)malloc(sizeof(float) * 4 * n);
  if (!f) lbm_die("cannot allocate memory for cells", __LINE__, __FILE__);
  lbm_oracle_final_state(&p, cells, obstacles, f, f + n, f + 2 * n, f + 3 * n);
  if (write_text) {
    FILE* fp = fopen(LBM_FINALSTATEFILE, "w");
    if (fp == NULL) lbm_die("could not open file output file", __LINE__, __FILE__);
    lbm_write_final_state_rows(fp, &hp, 0, hp.ny, f, f + n, f + 2 * n, f + 3 * n, obstacles);
    fclose(fp);
  }
  lbm_write_av_vels(LBM_AVVELSFILE, av_vels, p.max_iters);
  if ((env = getenv("LBM_PRESSURE_BIN")) && *env) {
    FILE* fp = fopen(env, "wb");
    if (fp == NULL) lbm_die("could not open file output file", __LINE__, __FILE__);
    fwrite(f + 3 * n, sizeof(float), n, fp);
    fclose(fp);
  }
  free(f); free(cells); free(tmp); free(av_vels); free(obstacles);
  return EXIT_SUCCESS;
}
