/*
 * lbm_hip.h -- C-ABI boundary of the MI355X (gfx950) D2Q9-BGK lattice-Boltzmann engine.
 *
 * This is the drop-in boundary for the reference's timestep hot path.  The reference
 * (Xinran1205/LBM-Asynchronous) has no plugin / FFI layer: its boundary is five C functions
 * called from main()'s loop (SerialCode/d2q9-bgk.c:97-101,113,166-170) over caller-owned host
 * arrays.  Each entry point below names the reference interface it replaces.
 *
 * Plain C types only; no C++ or torch types cross this boundary.  One host thread per context.
 *
 * Error behaviour follows the reference: by default an error prints
 *     "Error at line <n> of file <f>:\n<message>\n"
 * to stderr and calls exit(EXIT_FAILURE), exactly like die() (SerialCode/d2q9-bgk.c:745-751).
 * A host that must survive errors (the Python test harness) switches to return codes with
 * lbm_set_error_mode(LBM_ERRORS_RETURN); functions then return LBM_FAILURE / NULL and
 * lbm_last_error() holds the message.  There is NO CPU fallback: without a usable HIP device
 * every compute entry point fails.
 *
 * Data layout on the device: structure-of-arrays interleaved by row, fp32 --
 * value (speed k, row y, column x) at base + (y*9 + k)*pitch + x -- with four halo rows below
 * and above the rows a slab owns (a K-step pass reads K rows beyond the slab, K <= 4); uint8 obstacle mask.
 * Host-facing arrays keep the reference's layouts: cells are array-of-structures
 * (9 consecutive floats per cell, cell index ii + jj*nx, SerialCode/d2q9-bgk.c:78-81),
 * obstacles are int[ny*nx] with 1 = blocked (:541, :570-601).
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_SUCCESS 0
#define LBM_FAILURE 1

#define LBM_ERRORS_DIE    0 /* reference behaviour: message to stderr + exit(EXIT_FAILURE) */
#define LBM_ERRORS_RETURN 1 /* return LBM_FAILURE / NULL, message kept for lbm_last_error() */

/* numerics mode of the collision kernel */
#define LBM_MATH_EXACT 0 /* reference operation order, IEEE / and sqrt, no FMA contraction:
                            the lattice is bit-identical to SerialCode's */
#define LBM_MATH_FAST  1 /* reciprocal multiplies + FMA; validated through the check.py rule */

/* How a pass across several slabs / ranks treats its halo rows */
#define LBM_HALO_SYNC  0 /* halo rows of the same timestep: the MPI_Waitall pattern
                            (MPI_Waitall/d2q9-bgk.c:225-253); results equal the single-domain run */
#define LBM_HALO_STALE 1 /* EXPERIMENTAL.  Halo rows one pass old: reproducible analogue of the reference's
                            MPI_Testall "stale halo" variant (MPI_Testall_OptimizedVersion/
                            d2q9-bgk.c:256-301); no pass ever waits for an exchange of its own.
                            Parity unpinned (the reference variant is non-deterministic, so no fixture
                            can exist); measured against the synchronous run by the check.py rule it
                            misses 1 % in dense decompositions: av_vels 4.7 % (128x256 / 2 slabs, step 2),
                            4.0 % (128x128 / 8 slabs, mid-transient), 1.2 % (256x256 / 4 slabs);
                            pressure stays within 0.01 % */
#define LBM_HALO_FRESHEST 2 /* EXPERIMENTAL.  The reference's rule itself -- post the exchange, relax the interior rows,
                            look ONCE whether the halo rows have arrived, relax the boundary rows either way
                            (MPI_Testall_OptimizedVersion/d2q9-bgk.c:262-290) -- with two guarantees the reference
                            does not give: a halo row is this step's or the step before's, never older (the stale
                            mode's exchange backs it), and never torn (whole rows are adopted, by a look at an id
                            that travels behind them).  Which of the two each side got in each step is logged
                            (lbm_read_halo_log); given the log the run is reproducible on the CPU
                            (tests/slab_model.py).  Results lie between the synchronous and the stale run and
                            differ from run to run.  Parity unpinned, as for LBM_HALO_STALE.  RCCL and
                            device-copy transports only */

/* Run constants: field-for-field the reference's t_param (SerialCode/d2q9-bgk.c:66-75). */
typedef struct {
  int   nx;           /* cells in x */
  int   ny;           /* cells in y */
  int   max_iters;    /* iterations (capacity of the av_vels record) */
  int   reynolds_dim; /* dimension for the Reynolds number */
  float density;      /* density per link */
  float accel;        /* density redistribution */
  float omega;        /* relaxation parameter */
} lbm_params;

typedef struct lbm_ctx lbm_ctx; /* opaque engine handle */

/* Static facts a host may query (no device needed). */
typedef struct {
  int    n_slabs;        /* row slabs this context owns (1 per GPU in single-process mode) */
  int    row_first;      /* first global row owned by this context */
  int    row_count;      /* number of global rows owned by this context */
  int    fluid_cells;    /* GLOBAL number of non-blocked cells (av_velocity's divisor) */
  int    steps_done;     /* timesteps advanced so far */
  int    math_mode;      /* LBM_MATH_EXACT or LBM_MATH_FAST */
  int    world_rank;     /* rank of this context in a multi-process run (0 otherwise) */
  int    world_size;     /* number of processes sharing the grid (1 otherwise) */
  int    steps_per_launch; /* timesteps one launch of the main kernel advances: 2-4 for the stream kernels
                              (three from 300 Ki cells, four from 3.5 Mi cells per slab), 3-4 for the LDS-tile
                              kernel (small single slabs), else 1 */
  int    halo_mode;      /* LBM_HALO_SYNC, LBM_HALO_STALE or LBM_HALO_FRESHEST (meaningful with several slabs / ranks) */
  int    band_rows;      /* launch geometry of the multi-step stream kernel: rows one wave sweeps ... */
  int    lane_cells;     /* ... and cells per lane (4 or 2); 0 / 0 when another kernel is the main one */
  int    nontemporal;    /* 1: the step kernels store with the nontemporal hint */
  int    graph_steps;    /* timesteps one hipGraph chunk replays (0: loop issued launch by launch) */
  int    resident_steps; /* > 0: lbm_run calls of at least resident_min_steps timesteps run as launches of the resident
                            kernel (lattice in registers, up to this many timesteps per launch); single periodic slabs of
                            at most 1024 x 4*CUs cells */
  int    resident_min_steps;
  int    resident_rows;      /* resident kernel: rows per band (4 or 2) ... */
  int    resident_group;     /* ... bands per workgroup (1, 2 or 4) ... */
  int    resident_one_xcd;   /* ... and 1 where the whole grid (at most 128 waves) runs on one XCD, one wave per SIMD */
} lbm_info;

/* ---- error handling -------------------------------------------------------------------- */
void        lbm_set_error_mode(int mode);
const char* lbm_last_error(void);

/* ---- library / device probes ------------------------------------------------------------ */
const char* lbm_version(void);     /* "lbm_hip <ver> gfx950" */
int         lbm_device_count(void); /* visible HIP devices; 0 when none (never dies) */

/*
 * Row decomposition used for slabs and ranks (host arithmetic only, no device):
 * part `index` of `parts` gets rows [*first, *first + *count) of ny.  Balanced blocks,
 * the first ny % parts parts get one row more.  (The reference's rule,
 * MPI_Waitall/d2q9-bgk.c:694-704, additionally forces 3 rows onto the last rank because its
 * acceleration pass runs after the halo rows were posted; here acceleration is fused into the
 * kernel that produces the row, so no such constraint exists.)  Returns LBM_FAILURE when a
 * part would own fewer than 2 rows.
 */
int lbm_partition_rows(int ny, int parts, int index, int* first, int* count);

/*
 * The halo exchange of one pass, as the engine posts it (host arithmetic only, no device): the reference's
 * MPI_Isend x2 + MPI_Irecv x2 (MPI_Waitall/d2q9-bgk.c:225-230) with whole boundary rows, `depth` rows per side.
 * out[0..3], in posting order: send my top rows to the north neighbour, send my bottom rows to the south
 * neighbour, receive my south halo from the south neighbour, receive my north halo from the north neighbour
 * (north = (index+1) % parts, south = (index-1+parts) % parts: the periodic ring of MPI/d2q9-bgk.c:210-211; with
 * two parts both neighbours are the same peer and the first send pairs with that peer's first receive).
 * row_first counts slab-local rows: 0 is the first owned row, negative rows are the south halo, rows >= `rows` the
 * north halo.  lbm_plan_halo_depth: the depth (= timesteps per pass) the engine uses for a grid cut into `parts`
 * row slabs with halos in the given math mode (environment overrides included); its own exchange is built from lbm_halo_plan, so a host
 * that replays the protocol (tests/test_multirank_gloo.py) cannot drift from it.
 */
typedef struct {
  int is_send;   /* 1: ncclSend / MPI_Isend, 0: ncclRecv / MPI_Irecv */
  int peer;      /* neighbour's index in the ring */
  int row_first; /* first slab-local row of the message */
  int row_count; /* rows in the message (each row: 9 planes x pitch floats on the device) */
} lbm_halo_op;
int lbm_halo_plan(int rows, int parts, int index, int depth, lbm_halo_op out[4]);
int lbm_plan_halo_depth(const lbm_params* params, int parts, int math_mode);

/* ---- create / destroy --------------------------------------------------------------------
 * Replaces the buffer set-up half of initialise() (SerialCode/d2q9-bgk.c:531-567) and
 * finalise() (:615-634).
 *
 * obstacles : int[ny*nx], 1 = blocked (the array initialise() builds, :570-601).
 * cells_aos : float[ny*nx*9] initial lattice in the reference's AoS layout, or NULL to start
 *             from the uniform equilibrium of :546-567 generated on the device.
 * n_gpus    : row slabs / devices to spread the grid over in THIS process (1..8).  Slab g runs
 *             on device g % lbm_device_count(); several slabs may share one device (used to
 *             test the halo path on a 1-GPU box).
 * math_mode : LBM_MATH_EXACT or LBM_MATH_FAST.
 */
lbm_ctx* lbm_create(const lbm_params* params, const int* obstacles, const float* cells_aos,
                    int n_gpus, int math_mode);

/*
 * One-process-per-GPU form (torchrun / RANK, WORLD_SIZE): this process owns the rows
 * lbm_partition_rows(ny, world_size, rank) of the global grid on HIP device `device`;
 * halo rows travel by RCCL send/recv (the GPU analogue of MPI_Isend/Irecv + Waitall,
 * MPI_Waitall/d2q9-bgk.c:225-243).  `unique_id` is the 128-byte RCCL id obtained by rank 0
 * from lbm_rccl_unique_id() and broadcast by the host (e.g. over torch.distributed).
 * obstacles is the GLOBAL mask (every rank parses the same file), cells_aos the GLOBAL
 * initial lattice or NULL.
 */
#define LBM_RCCL_ID_BYTES 128
int      lbm_rccl_unique_id(void* id_out);

/*
 * Which RCCL serves the halo exchange, and what it says about the ring -- the question "did RCCL see N ranks, and
 * which RCCL" of a multi-GPU run, answered from the run's own record (the reference prints "Process %d of %d started",
 * MPI/d2q9-bgk.c:151, from MPI_Comm_rank / MPI_Comm_size, MPI_Waitall/d2q9-bgk.c:143-147).
 * RCCL is bound at first use, not at link time, in this order: the file named by LBM_RCCL_LIB; a librccl.so.1 the
 * process has already mapped (a host that imported PyTorch carries PyTorch's bundled RCCL next to its bundled HIP
 * runtime -- a communicator has to come from the RCCL built for the HIP runtime in the process); ROCm's own
 * /opt/rocm/lib/librccl.so.1.  A single-GPU run never loads it.
 * ctx == NULL: binds the library and reports `loaded`, `version`, `library`.  With a context: its communicators
 * (n_comms: one per slab in the one-process form, one in the one-process-per-GPU form, none for a single slab or the
 * device-copy / hosted transports) and nranks / rank as ncclCommCount / ncclCommUserRank of the first one report them.
 */
typedef struct {
  int  loaded;        /* 1: a librccl is bound to this engine */
  int  version;       /* ncclGetVersion(), e.g. 22707 = 2.27.7 */
  int  n_comms;       /* communicators the context holds */
  int  nranks;        /* ranks in the ring as RCCL counts them (ncclCommCount); 0 without a communicator */
  int  rank;          /* this context's rank in it (ncclCommUserRank) */
  char library[512];  /* file the RCCL entry points were bound from */
} lbm_rccl_status;
int      lbm_rccl_info(const lbm_ctx* ctx, lbm_rccl_status* out);
lbm_ctx* lbm_create_rank(const lbm_params* params, const int* obstacles, const float* cells_aos,
                         int rank, int world_size, const void* unique_id, int device,
                         int math_mode);

/*
 * The same, without the global map on every rank -- the reference's scatter (rank 0 parses, every rank
 * receives only its rows, MPI_Waitall/d2q9-bgk.c:794-842):
 *   obstacle_rows  : int[(row_count + 2*LBM_MASK_HALO_ROWS) * nx] -- the rows lbm_partition_rows gives this
 *                    rank, preceded and followed by LBM_MASK_HALO_ROWS periodic neighbour rows (global rows
 *                    row_first - LBM_MASK_HALO_ROWS ... row_first + row_count + LBM_MASK_HALO_ROWS - 1, folded
 *                    into [0, ny)): a multi-step pass relaxes that many rows beyond the slab redundantly;
 *   cells_rows_aos : float[row_count * nx * 9], this rank's rows only, or NULL (uniform equilibrium).
 * The mask is built on the device; the fluid-cell count (av_velocity's divisor) is a device reduction over the
 * owned rows summed over the ranks by one all-reduce.  lbm_create and lbm_create_rank count the same way.
 */
#define LBM_MASK_HALO_ROWS 3
lbm_ctx* lbm_create_rank_rows(const lbm_params* params, const int* obstacle_rows, const float* cells_rows_aos,
                              int rank, int world_size, const void* unique_id, int device,
                              int math_mode);

/*
 * Obstacles given as a small tile repeated periodically over the grid: cell (x, y) is blocked iff the tile's cell
 * (x mod tile_nx, y mod tile_ny) is (tile: int[tile_ny*tile_nx]).  This is how BASELINE.md section 4 defines the
 * synthetic 8192x8192 and 16384x16384 grids (the reference's 1024x1024 map tiled); the mask is expanded on the
 * device from the tile, so a 16384x16384 run never holds a 1 GiB int map on the host.
 */
lbm_ctx* lbm_create_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                          const float* cells_aos, int n_gpus, int math_mode);
lbm_ctx* lbm_create_rank_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                               int rank, int world_size, const void* unique_id, int device,
                               int math_mode);

/*
 * One process per GPU with the HOST's own message passing instead of RCCL -- the closest fit to the reference's MPI
 * programs, which keep MPI_Isend / MPI_Irecv / MPI_Waitall (MPI_Waitall/d2q9-bgk.c:225-243) and MPI_Reduce
 * (:321): the engine hands the boundary rows of a pass to `exchange` in pinned host buffers and takes the halo rows
 * back, and sums its per-rank totals through `allreduce_sum`.
 *   exchange(user, 4, ops, buffers, floats): the four messages of lbm_halo_plan in posting order; for ops[i].is_send
 *     send buffers[i][0 .. floats) to rank ops[i].peer, else receive that many floats from it into buffers[i].  Post all
 *     four before waiting for any (two ranks are each other's north AND south neighbour).  Return 0 on success.
 *   allreduce_sum(user, values, n): in-place sum of n doubles over all ranks; every rank receives it.  Return 0.
 * The callbacks block the host, so this transport does not hide the exchange behind the interior rows; it exists for
 * MPI-launched hosts and to run the rank decomposition with several ranks on ONE device (tests).  obstacles /
 * cells_aos: the global arrays, as lbm_create_rank (the _rows / _tiled forms below take only a rank's share).
 * Every rank must issue the same sequence of calls.
 */
typedef struct {
  int (*exchange)(void* user, int n_ops, const lbm_halo_op* ops, float* const* buffers, size_t floats_per_message);
  int (*allreduce_sum)(void* user, double* values, int n);
  void* user;
} lbm_host_comm;
lbm_ctx* lbm_create_rank_hosted(const lbm_params* params, const int* obstacles, const float* cells_aos,
                                int rank, int world_size, const lbm_host_comm* comm, int device,
                                int math_mode);
/*
 * The same without the global arrays on every rank -- what the reference's MPI programs do: rank 0 parses the
 * obstacle file and sends every rank its rows (MPI_Waitall/d2q9-bgk.c:816-842).  Arguments as lbm_create_rank_rows
 * (this rank's rows with LBM_MASK_HALO_ROWS periodic neighbour rows on each side; this rank's cells or NULL) and
 * lbm_create_rank_tiled (a small tile repeated over the grid, expanded on the device).
 */
lbm_ctx* lbm_create_rank_hosted_rows(const lbm_params* params, const int* obstacle_rows, const float* cells_rows_aos,
                                     int rank, int world_size, const lbm_host_comm* comm, int device,
                                     int math_mode);
lbm_ctx* lbm_create_rank_hosted_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                                      int rank, int world_size, const lbm_host_comm* comm, int device,
                                      int math_mode);

void     lbm_destroy(lbm_ctx* ctx);
int      lbm_get_info(const lbm_ctx* ctx, lbm_info* out);

/*
 * Halo treatment for the following lbm_run calls (default LBM_HALO_SYNC, or LBM_HALO_STALE / LBM_HALO_FRESHEST when
 * the environment holds LBM_HALO_MODE=stale / freshest).  Replaces the choice between the reference's
 * MPI_Waitall and MPI_Testall_OptimizedVersion programs (main loop :256-301 of the latter).
 * In stale mode every lbm_run call starts from freshly exchanged halos; from its second pass on, a
 * pass reads the halo rows its neighbours produced one pass earlier.  Every rank of a multi-process
 * run must select the same mode.  No effect on a single periodic slab.
 */
int      lbm_set_halo_mode(lbm_ctx* ctx, int mode);

/*
 * LBM_HALO_FRESHEST's record of what each look found: out[t * n_slabs + s], timesteps t < n_steps <= steps done,
 * slabs s of this context (n_slabs = info.n_slabs; 1 for rank contexts): bit 0 set = the south halo row of that
 * step was the neighbour's row of the same step, bit 1 = the north one; a clear bit = the row of the step before.
 * Steps run in the other modes read 3 (synchronous; first pass of every call) or are not recorded (stale).
 * The reference has no counterpart: its MPI_Testall result is discarded (d2q9-bgk.c:279-280).
 */
int      lbm_read_halo_log(lbm_ctx* ctx, unsigned char* out, int n_steps);

/* ---- the hot path ------------------------------------------------------------------------
 * lbm_run replaces n_steps trips of the driver loop (SerialCode/d2q9-bgk.c:166-170):
 *     timestep(params, cells, tmp_cells, obstacles);      // accelerate_flow, propagate,
 *                                                         // rebound, collision  (:207-407)
 *     av_vels[tt] = av_velocity(params, cells, obstacles); // (:409-458)
 * with no host round trip per step.  Per-step sums of |u| accumulate on the device.
 * lbm_sync waits for the device (call it before reading the clock, as the reference's
 * "Elapsed Compute time" brackets the loop, :162-185).
 */
int lbm_run(lbm_ctx* ctx, int n_steps);
int lbm_sync(lbm_ctx* ctx);

/*
 * Same as lbm_run, additionally timing the step kernels with HIP events recorded on the
 * stream(s) the kernels are launched on.  *kernel_ms_per_step receives the average device
 * time of one timestep (max over slabs).
 */
int lbm_run_timed(lbm_ctx* ctx, int n_steps, float* kernel_ms_per_step);

/* ---- results -----------------------------------------------------------------------------
 * lbm_read_av_vels: out[t] = tot_u[t] / (float)fluid_cells for the first n recorded steps
 *   (the values main() stores at SerialCode/d2q9-bgk.c:169; fp32 division as :457).
 *   In a multi-process context the per-rank sums are first all-reduced (the reference's
 *   MPI_Reduce, MPI/d2q9-bgk.c:298-309); every rank receives the result.
 * lbm_read_cells: the owned rows of the lattice in the reference's AoS layout, so that
 *   write_values()/calc_reynolds() logic (:637-642, :662-743) can run on it unchanged.
 * lbm_read_final_state: u_x, u_y, |u|, pressure of the owned rows computed on the device
 *   with the formulas of write_values() (:684-719); blocked cells give 0,0,0,density*c_sq.
 * lbm_av_velocity: av_velocity() of the current lattice (:409-458) -- what calc_reynolds()
 *   (:637-642) multiplies; lbm_total_density: total_density() (:644-660).
 *   Both are global (all slabs / all ranks).
 */
int   lbm_read_av_vels(lbm_ctx* ctx, float* out, int n);
int   lbm_read_cells(lbm_ctx* ctx, float* cells_aos);
int   lbm_read_final_state(lbm_ctx* ctx, float* u_x, float* u_y, float* u_mag, float* pressure);
int   lbm_av_velocity(lbm_ctx* ctx, float* out);
int   lbm_total_density(lbm_ctx* ctx, double* out);
int   lbm_calc_reynolds(lbm_ctx* ctx, float* out);

#ifdef __cplusplus
}
#endif
#endif /* LBM_HIP_H */
