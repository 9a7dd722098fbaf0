/*
 * lbm_io.c -- parameter / obstacle readers and result writers of the d2q9-bgk command line.
 * Formats and messages follow /root/reference/SerialCode/d2q9-bgk.c (lines cited in lbm_io.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "lbm_io.h"

void lbm_die(const char* message, const int line, const char* file)
{
  fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  fprintf(stderr, "%s\n", message);
  fflush(stderr);
  exit(EXIT_FAILURE);
}

void lbm_usage(const char* exe)
{
  fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", exe);
  exit(EXIT_FAILURE);
}

void lbm_read_params(const char* paramfile, lbm_params* params)
{
  char message[1024];
  FILE* fp = fopen(paramfile, "r");
  if (fp == NULL) {
    snprintf(message, sizeof(message), "could not open input parameter file: %s", paramfile);
    lbm_die(message, __LINE__, __FILE__);
  }

  /* seven values, one per line, in this order (SerialCode/d2q9-bgk.c:480-506) */
  struct { const char* fmt; void* dst; const char* what; } field[7] = {
    { "%d\n", &params->nx,           "could not read param file: nx" },
    { "%d\n", &params->ny,           "could not read param file: ny" },
    { "%d\n", &params->max_iters,    "could not read param file: maxIters" },
    { "%d\n", &params->reynolds_dim, "could not read param file: reynolds_dim" },
    { "%f\n", &params->density,      "could not read param file: density" },
    { "%f\n", &params->accel,        "could not read param file: accel" },
    { "%f\n", &params->omega,        "could not read param file: omega" },
  };
  for (int i = 0; i < 7; i++) {
    if (fscanf(fp, field[i].fmt, field[i].dst) != 1) lbm_die(field[i].what, __LINE__, __FILE__);
  }
  fclose(fp);
}

int* lbm_read_obstacles(const char* obstaclefile, const lbm_params* params)
{
  char message[1024];
  const size_t n = (size_t)params->nx * (size_t)params->ny;
  int* map = (int*)calloc(n, sizeof(int));
  if (map == NULL) lbm_die("cannot allocate column memory for obstacles", __LINE__, __FILE__);

  FILE* fp = fopen(obstaclefile, "r");
  if (fp == NULL) {
    snprintf(message, sizeof(message), "could not open input obstacles file: %s", obstaclefile);
    lbm_die(message, __LINE__, __FILE__);
  }

  int xx, yy, blocked, got;
  while ((got = fscanf(fp, "%d %d %d\n", &xx, &yy, &blocked)) != EOF) {
    /* the reference's checks and messages, SerialCode/d2q9-bgk.c:591-597 */
    if (got != 3) lbm_die("expected 3 values per line in obstacle file", __LINE__, __FILE__);
    if (xx < 0 || xx > params->nx - 1) lbm_die("obstacle x-coord out of range", __LINE__, __FILE__);
    if (yy < 0 || yy > params->ny - 1) lbm_die("obstacle y-coord out of range", __LINE__, __FILE__);
    if (blocked != 1) lbm_die("obstacle blocked value should be 1", __LINE__, __FILE__);
    map[(size_t)xx + (size_t)yy * params->nx] = blocked;
  }
  fclose(fp);
  return map;
}

int* lbm_tile_obstacles(const int* tile, int tile_nx, int tile_ny, int nx, int ny)
{
  int* map = (int*)malloc((size_t)nx * (size_t)ny * sizeof(int));
  if (map == NULL) lbm_die("cannot allocate column memory for obstacles", __LINE__, __FILE__);
  for (int y = 0; y < ny; y++) {
    const int* trow = tile + (size_t)(y % tile_ny) * tile_nx;
    int* row = map + (size_t)y * nx;
    for (int x = 0; x < nx; x++) row[x] = trow[x % tile_nx];
  }
  return map;
}

/* one line per cell is ~85 bytes of printf work; 8192^2 is 5.7 GB of text.  Rows are formatted by
 * all cores into per-thread buffers (snprintf, so the bytes are exactly what the reference's
 * fprintf produces) and written out in order. */
#define LBM_LINE_MAX 128 /* "%d %d" + 4 x "%.12E" (<= 20 chars each) + " %d\n" */

void lbm_write_final_state_rows(FILE* fp, const lbm_params* params, int row_first, int row_count,
                                const float* u_x, const float* u_y, const float* u_mag,
                                const float* pressure, const int* obstacles)
{
  const int nx = params->nx;
  int n_threads = 1;
#ifdef _OPENMP
  n_threads = omp_get_max_threads();
#endif
  /* rows per thread and round: keep each buffer around 4 MiB */
  int rows_per_task = (int)((4L << 20) / ((long)nx * LBM_LINE_MAX));
  if (rows_per_task < 1) rows_per_task = 1;
  const size_t buf_bytes = (size_t)rows_per_task * nx * LBM_LINE_MAX;
  char** buf = (char**)malloc(sizeof(char*) * (size_t)n_threads);
  size_t* used = (size_t*)malloc(sizeof(size_t) * (size_t)n_threads);
  if (buf == NULL || used == NULL) lbm_die("cannot allocate memory for output buffers", __LINE__, __FILE__);
  for (int t = 0; t < n_threads; t++) {
    buf[t] = (char*)malloc(buf_bytes);
    if (buf[t] == NULL) lbm_die("cannot allocate memory for output buffers", __LINE__, __FILE__);
  }

  for (int r0 = 0; r0 < row_count; r0 += rows_per_task * n_threads) {
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < n_threads; t++) {
      const int begin = r0 + t * rows_per_task;
      int end = begin + rows_per_task;
      if (end > row_count) end = row_count;
      char* out = buf[t];
      for (int r = begin; r < end; r++) {
        const int jj = row_first + r;
        for (int ii = 0; ii < nx; ii++) {
          const size_t c = (size_t)r * nx + ii;
          /* the reference's format, SerialCode/d2q9-bgk.c:722 */
          out += snprintf(out, LBM_LINE_MAX, "%d %d %.12E %.12E %.12E %.12E %d\n", ii, jj, u_x[c], u_y[c],
                          u_mag[c], pressure[c], obstacles[(size_t)jj * nx + ii]);
        }
      }
      used[t] = (size_t)(out - buf[t]);
    }
    for (int t = 0; t < n_threads; t++)
      if (used[t] > 0 && fwrite(buf[t], 1, used[t], fp) != used[t])
        lbm_die("could not write file output file", __LINE__, __FILE__);
  }
  for (int t = 0; t < n_threads; t++) free(buf[t]);
  free(buf);
  free(used);
}

void lbm_write_av_vels(const char* path, const float* av_vels, int n)
{
  FILE* fp = fopen(path, "w");
  if (fp == NULL) lbm_die("could not open file output file", __LINE__, __FILE__);
  for (int ii = 0; ii < n; ii++) fprintf(fp, "%d:\t%.12E\n", ii, av_vels[ii]);
  fclose(fp);
}
