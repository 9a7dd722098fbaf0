/*
 * lbm_io.h -- host-side file formats of the d2q9-bgk command line (plain C99).
 *
 * Reads the 7-value parameter file and the "x y 1" obstacle file, and writes
 * final_state.dat / av_vels.dat, in the formats and with the error messages of the reference
 * program (/root/reference/SerialCode/d2q9-bgk.c: initialise() :460-613, write_values()
 * :662-743, die() :745-751, usage() :753-757).  No lattice arithmetic happens here: the
 * per-cell output quantities arrive already computed (on the device, lbm_read_final_state()).
 */
#ifndef LBM_IO_H
#define LBM_IO_H

#include <stdio.h>

#include "../../include/lbm_hip.h"

#define LBM_FINALSTATEFILE "final_state.dat" /* SerialCode/d2q9-bgk.c:62 */
#define LBM_AVVELSFILE     "av_vels.dat"     /* :63 */

/* message to stderr + exit(EXIT_FAILURE), the reference's die() (:745-751) */
void lbm_die(const char* message, const int line, const char* file);
/* "Usage: %s <paramfile> <obstaclefile>" + exit(EXIT_FAILURE) (:753-757) */
void lbm_usage(const char* exe);

/* the seven fscanf reads of initialise() (:471-509); dies with the reference's messages */
void lbm_read_params(const char* paramfile, lbm_params* params);

/* malloc + zero an int[ny*nx] map and fill it from the obstacle file (:541-604);
 * dies with the reference's messages on malformed or out-of-range lines */
int* lbm_read_obstacles(const char* obstaclefile, const lbm_params* params);

/* synthetic large grids: tile a small obstacle map periodically,
 * cell (x,y) of the big map is blocked iff (x % tile_nx, y % tile_ny) is blocked in the tile */
int* lbm_tile_obstacles(const int* tile, int tile_nx, int tile_ny, int nx, int ny);

/* final_state.dat: "%d %d %.12E %.12E %.12E %.12E %d\n" = ii jj u_x u_y u pressure obstacle,
 * jj outer / ii inner (:679-723).  Rows [row_first, row_first+row_count) are appended to fp. */
void lbm_write_final_state_rows(FILE* fp, const lbm_params* params, int row_first, int row_count,
                                const float* u_x, const float* u_y, const float* u_mag,
                                const float* pressure, const int* obstacles);

/* av_vels.dat: "%d:\t%.12E\n" (:735-738) */
void lbm_write_av_vels(const char* path, const float* av_vels, int n);

#endif /* LBM_IO_H */
