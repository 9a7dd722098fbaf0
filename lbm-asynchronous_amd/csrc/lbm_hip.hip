// lbm_hip.hip -- host side of the C-ABI engine declared in include/lbm_hip.h.
//
// Owns the device lattices (row slabs, one per GPU), the per-step launch sequence, the halo
// exchange (RCCL send/recv on a side stream, overlapped with the interior rows -- the GPU
// analogue of /root/reference/MPI_Waitall/d2q9-bgk.c:225-253) and the result read-back.
// No CPU compute path exists here: without a HIP device every compute entry point fails.
#include "../../include/lbm_hip.h"
#include "lbm_kernels.hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <algorithm>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxSlabs = 8;
constexpr int kPartSlots = lbm::kPartSlotsMax;  // steps whose partial sums are buffered before one reduce launch

enum HaloMode { HALO_SELF = 0, HALO_MEMCPY = 1, HALO_RCCL = 2, HALO_HOST = 3 };

// ---- error handling (reference: die(), SerialCode/d2q9-bgk.c:745-751) -----------------------
int g_error_mode = LBM_ERRORS_DIE;
thread_local char g_last_error[1024] = "";

void raise_error(int line, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  if (g_error_mode == LBM_ERRORS_DIE) {
    fprintf(stderr, "Error at line %d of file %s:\n", line, __FILE__);
    fprintf(stderr, "%s\n", g_last_error);
    fflush(stderr);
    exit(EXIT_FAILURE);
  }
}

#define LBM_FAIL(ret, ...)              \
  do {                                  \
    raise_error(__LINE__, __VA_ARGS__); \
    return ret;                         \
  } while (0)

#define HIP_TRY(ret, expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) LBM_FAIL(ret, "HIP error: %s (%s)", hipGetErrorString(e_), #expr); \
  } while (0)

#define NCCL_TRY(ret, expr)                                                               \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess) LBM_FAIL(ret, "RCCL error: %s (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", #expr); \
  } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}


// ---- RCCL, bound at first use -----------------------------------------------------------------------------------
// The library is NOT a link-time dependency: a single-GPU run never loads it, and WHICH librccl serves a multi-GPU
// run is a decision taken here, not an accident of load order:
//   1. LBM_RCCL_LIB=<path>: that file (RTLD_LOCAL | RTLD_DEEPBIND);
//   2. a librccl.so.1 the process has already mapped -- a host that imported PyTorch first carries torch's bundled
//      RCCL together with torch's bundled HIP runtime (both resolve by soname before anything of this engine loads),
//      and a communicator must come from the RCCL built for the HIP runtime it runs on;
//   3. ROCm's own, /opt/rocm/lib/librccl.so.1 (then the bare soname): the C host program and torch-free hosts.
// lbm_rccl_info() reports which one it was, its version and what the communicator says about the ring.
struct RcclApi {
  void* handle = nullptr;
  char path[512] = "";
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
char g_rccl_error[768] = "";

void rccl_bind() {
  RcclApi& r = g_rccl;
  const char* forced = getenv("LBM_RCCL_LIB");
  if (forced && *forced) {
    r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!r.handle) { snprintf(g_rccl_error, sizeof(g_rccl_error), "LBM_RCCL_LIB=%s: %s", forced, dlerror()); return; }
  } else {
    r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // already in the process (e.g. PyTorch's)
    if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) { snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl.so.1 not found: %s", dlerror()); return; }
  }
  bool ok = true;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.handle, name);
    if (!p) { ok = false; snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl lacks %s", name); }
    return p;
  };
#define LBM_RCCL_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(sym(name))
  LBM_RCCL_SYM(GetVersion, "ncclGetVersion");        LBM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  LBM_RCCL_SYM(CommInitRank, "ncclCommInitRank");    LBM_RCCL_SYM(CommInitAll, "ncclCommInitAll");
  LBM_RCCL_SYM(CommDestroy, "ncclCommDestroy");      LBM_RCCL_SYM(CommCount, "ncclCommCount");
  LBM_RCCL_SYM(CommUserRank, "ncclCommUserRank");    LBM_RCCL_SYM(GroupStart, "ncclGroupStart");
  LBM_RCCL_SYM(GroupEnd, "ncclGroupEnd");            LBM_RCCL_SYM(Send, "ncclSend");
  LBM_RCCL_SYM(Recv, "ncclRecv");                    LBM_RCCL_SYM(AllReduce, "ncclAllReduce");
  LBM_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef LBM_RCCL_SYM
  if (!ok) { r.handle = nullptr; return; }
  Dl_info di;
  if (dladdr(reinterpret_cast<void*>(r.GetVersion), &di) && di.dli_fname) {
    char real[512];
    const char* shown = realpath(di.dli_fname, real) ? real : di.dli_fname;
    strncpy(r.path, shown, sizeof(r.path) - 1);
  }
}

// the bound RCCL, or nullptr with the reason in g_rccl_error
RcclApi* rccl() {
  std::call_once(g_rccl_once, rccl_bind);
  return g_rccl.handle ? &g_rccl : nullptr;
}

#define RCCL_OR_FAIL(ret)                                                       \
  RcclApi* rc_api_ = rccl();                                                    \
  if (!rc_api_) LBM_FAIL(ret, "RCCL is not available: %s", g_rccl_error)

struct Slab {
  int device = 0;
  int row_first = 0;  // global row of slab row 0
  int rows = 0;       // owned rows
  int accel_row = lbm::kNoRow;  // slab row (may be a halo row) holding global row ny-2
  int accel_row2 = lbm::kNoRow; // its second periodic image among the halo rows (a ring of ONE slab with 3-step passes)
  float* lat_alloc[2] = {nullptr, nullptr};  // (rows + 2*kHaloRows) x row_pitch each
  float* lat[2] = {nullptr, nullptr};        // row 0 of each lattice (= lat_alloc + kHaloRows rows)
  unsigned char* mask_alloc = nullptr;       // (rows + 2*kMaskHalo) x pitch: neighbour rows below and above
  unsigned char* mask = nullptr;             // row 0 of the mask
  float* partials = nullptr;  // kPartSlots x part_stride
  double* tot_u = nullptr;    // capacity entries: per-step sum of |u| over this slab
  double* scratch = nullptr;  // 2 x kSumBlocks doubles for lattice_sums
  double* reduce_buf = nullptr;  // ranked contexts: capacity doubles for the av_vels all-reduce
  int* flushed_dev = nullptr; // graph replay: index of the first step of the chunk being reduced
  uint4* res_gran = nullptr;               // resident kernel: seam granules {v, v, v, tag}: [2][bands][2][nx]
  float* res_part = nullptr;               // resident kernel: per-band partial sums of a launch, [kResidentChunk][bands]
  int* res_status = nullptr;               // resident kernel: 0, or the reason a workgroup gave up
  int* res_status_host = nullptr;          // pinned copy of it, refreshed behind every launch (read by lbm_sync)
  hipGraphExec_t chunk_graph[2] = {nullptr, nullptr};  // kPartSlots timesteps + their reduce, by lattice parity
  hipStream_t compute = nullptr, comm = nullptr;
  hipEvent_t ev_boundary = nullptr, ev_halo = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_interior[2] = {nullptr, nullptr};  // interior kernel of step t -> [t & 1]
  hipEvent_t ev_flush = nullptr;                   // partials reduced: their slots may be reused
  hipEvent_t ev_step = nullptr;                    // stale-halo mode: whole-slab pass finished; graph replay: join
  hipEvent_t ev_fork = nullptr;                    // graph replay: the other streams join the capture / follow the chunks
  hipEvent_t ev_x[2] = {nullptr, nullptr};         // stale-halo mode: exchange for pass m landed -> [(m + 1) & 1]
  // freshest-available mode (LBM_HALO_FRESHEST), allocated at its first use
  float* fresh_stage = nullptr;          // [parity][side: 0 south halo, 1 north halo][row_pitch]: this pass's rows, if they make it
  unsigned* fresh_arrived = nullptr;     // [parity][side]: id (global step + 1) of the step whose row the staging holds
  unsigned* fresh_id_src = nullptr;      // [parity]: the id this slab ships behind its rows (device copies)
  int* fresh_decision = nullptr;         // bit 0 / 1: south / north staging row adopted in the current pass
  unsigned char* fresh_log = nullptr;    // [capacity]: the decision of every step (3 where the halos were fresh anyway)
  hipEvent_t ev_fresh[2] = {nullptr, nullptr};  // LBM_FRESH_FORCE=wait: this pass's rows and ids are out
  ncclComm_t nccl = nullptr;
  lbm::SlotCounts slot_counts;  // partials written into each buffered slot (launch geometries differ)
  int blocks_main = 0;      // interior rows (or all rows in HALO_SELF)
  int blocks_boundary = 0;  // rows 0 and rows-1 (halo modes)
  long fluid_cells = 0;     // non-blocked cells among the owned rows
};

constexpr int kSumBlocks = 1024;
constexpr int kResidentChunk = 4096;  // most timesteps one launch of the resident kernel advances
// step_tile instantiations: own cells per workgroup (tw x th), halo depth = most timesteps per launch, threads
struct TileShape { int tw, th, kmax, threads; void (*exact)(const lbm::TileArgs); void (*fast)(const lbm::TileArgs); };
#define LBM_TILE_SHAPE(TW, TH, K, T) {TW, TH, K, T, lbm::step_tile<0, TW, TH, K, T>, lbm::step_tile<1, TW, TH, K, T>}
const TileShape kTileShapes[] = {
    LBM_TILE_SHAPE(16, 8, 4, 384),   // 0: tiny grids: one thread per staged cell (24 x 16)
    LBM_TILE_SHAPE(16, 8, 8, 768),   // 1: same, halo of 8
    LBM_TILE_SHAPE(32, 16, 2, 640),  // 2..: larger tiles, less redundant halo work
    LBM_TILE_SHAPE(32, 16, 3, 768),
    LBM_TILE_SHAPE(32, 16, 4, 896),
    LBM_TILE_SHAPE(64, 16, 2, 640),
    LBM_TILE_SHAPE(64, 8, 2, 704),
};
constexpr int kTileShapeCount = (int)(sizeof(kTileShapes) / sizeof(kTileShapes[0]));
constexpr int kHaloRows = 4;  // halo rows kept below and above every slab (a K-step pass reads K rows beyond the slab)
constexpr int kMaskHalo = LBM_MASK_HALO_ROWS;  // mask rows kept beyond the slab: a K-step pass relaxes K-1 halo rows redundantly
static_assert(kMaskHalo == kHaloRows - 1, "mask halo");

// where the obstacle flags come from (the reference: initialise() fills int[ny*nx] on rank 0 and, in the MPI variants,
// sends every rank its rows, MPI_Waitall/d2q9-bgk.c:794-842)
enum ObstacleKind { OBST_GLOBAL = 0, OBST_ROWS = 1, OBST_TILE = 2 };
struct ObstacleSource {
  int kind;
  const int* data;  // GLOBAL: int[ny*nx]; ROWS: this context's rows with kMaskHalo periodic neighbour rows each side;
                    // TILE: int[tile_ny*tile_nx], repeated periodically over the grid
  int tile_nx, tile_ny;
  bool local_cells;  // cells_aos holds only this context's rows (ROWS form)
};

// One host thread per slab for the issue loop of a one-process multi-GPU run: a pass enqueues
// ~10 runtime calls per slab, which a single thread issues at 25-30 us per slab -- more than an
// 8-GPU pass of 8192^2 takes on the devices.  The team runs the per-slab bodies of each phase
// concurrently (fork-join); phases stay ordered, so event records always precede the waits of the
// next phase.  Workers spin (yield) while a run is in flight and sleep on a condition variable
// between runs.  Single-slab contexts and the one-process-per-GPU form have no team.
struct SlabTeam {
  std::vector<std::thread> threads;
  std::function<int(int)> job;
  std::atomic<int> generation{0};
  std::atomic<int> pending{0};
  std::atomic<int> failed{0};
  std::atomic<bool> stop{false};
  std::atomic<bool> hot{false};
  std::mutex m;
  std::condition_variable cv;
  char error[kMaxSlabs][1024];

  void worker(int s) {
    int seen = 0;
    for (;;) {
      while (generation.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_acquire)) {
        if (hot.load(std::memory_order_acquire)) {
          std::this_thread::yield();
        } else {
          std::unique_lock<std::mutex> lk(m);
          cv.wait_for(lk, std::chrono::milliseconds(2));
        }
      }
      if (stop.load(std::memory_order_acquire)) return;
      seen = generation.load(std::memory_order_acquire);
      const int rc = job(s);
      if (rc != LBM_SUCCESS) {
        strncpy(error[s], g_last_error, sizeof(error[s]) - 1);
        failed.store(1, std::memory_order_release);
      }
      pending.fetch_sub(1, std::memory_order_release);
    }
  }
  void start(int n) {
    for (int s = 0; s < n; s++) {
      error[s][0] = 0;
      threads.emplace_back([this, s] { worker(s); });
    }
  }
  int run(int n, const std::function<int(int)>& f) {
    job = f;
    failed.store(0, std::memory_order_relaxed);
    pending.store(n, std::memory_order_release);
    generation.fetch_add(1, std::memory_order_release);
    if (!hot.load(std::memory_order_acquire)) cv.notify_all();
    while (pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    if (failed.load(std::memory_order_acquire)) {
      for (int s = 0; s < n; s++)
        if (error[s][0]) {
          strncpy(g_last_error, error[s], sizeof(g_last_error) - 1);
          error[s][0] = 0;
          break;
        }
      return LBM_FAILURE;
    }
    return LBM_SUCCESS;
  }
  void shutdown() {
    stop.store(true, std::memory_order_release);
    cv.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
  }
};

}  // namespace

// A hipGraph chunk is BUILT, not captured: the issue code below runs against these virtual streams and events, which
// keep exactly the bookkeeping stream capture would (a stream's pending dependencies, an event's snapshot of them) and
// turn every launch / copy into an explicit node with explicit dependencies.  Multi-stream capture cannot be used:
// hip::Stream::EndCapture() of ROCm 7.2 recurses without end once three or more side streams have waited on each
// other's events (profiles/r03_graph_capture_defect.md, tools/capture_ring_repro.hip).
struct GraphBuilder {
  hipGraph_t graph = nullptr;
  struct VStream { hipStream_t key; std::vector<hipGraphNode_t> last; };
  struct VEvent { hipEvent_t key; std::vector<hipGraphNode_t> nodes; };
  std::vector<VStream> streams;
  std::vector<VEvent> events;
  std::vector<hipGraphNode_t>& last_of(hipStream_t st) {
    for (auto& v : streams) if (v.key == st) return v.last;
    streams.push_back({st, {}});
    return streams.back().last;
  }
  std::vector<hipGraphNode_t>* snapshot_of(hipEvent_t ev, bool create) {
    for (auto& v : events) if (v.key == ev) return &v.nodes;
    if (!create) return nullptr;
    events.push_back({ev, {}});
    return &events.back().nodes;
  }
};

struct lbm_ctx {
  lbm_params p;
  int pitch = 0;
  long plane_stride = 0;  // floats between the 9 planes of one row (= pitch + optional pad)
  long row_pitch = 0;     // floats between lattice rows (= 9 * plane_stride)
  int n_slabs = 0;
  Slab slab[kMaxSlabs];
  int cur = 0;  // lattice holding the current state
  int steps_done = 0;
  int capacity = 0;  // entries in tot_u
  int fluid_cells = 0;
  int math_mode = LBM_MATH_EXACT;
  int halo = HALO_SELF;
  int halo_mode = LBM_HALO_SYNC;  // LBM_HALO_STALE: passes consume the halos of the previous pass
  int rank = 0, world = 1;  // multi-process
  bool ranked = false;      // created by lbm_create_rank* (one process per GPU: rank / world describe the ring)
  bool hosted = false;      // ... with the host's own message passing instead of RCCL (lbm_create_rank_hosted)
  lbm_host_comm host_comm = {nullptr, nullptr, nullptr};
  float* host_send[2] = {nullptr, nullptr};  // pinned staging buffers of the hosted exchange: kHaloRows rows each
  float* host_recv[2] = {nullptr, nullptr};
  int row_first = 0, row_count = 0;
  int slot_fill = 0;  // partial slots used since the last reduce
  long part_stride = 0;
  bool vec4 = false;
  int neigh = 0;  // step_vec4 NEIGH flavour (LBM_NEIGH overrides)
  int nts = 1;    // nontemporal stores (LBM_NTS overrides)
  int snake = 0;  // alternate the sweep direction every step (LBM_SNAKE overrides)
  int fuse2 = 0;  // several timesteps per pass over memory (the stream kernels) where the slabs allow it
  int pass_steps = 2;  // ... how many: 2 or 3 (LBM_PASS_STEPS)
  int prefetch = 0;    // stream kernel: request the next row before relaxing the current one (LBM_PREFETCH)
  int xcd_chunk = 0;   // stream kernel: strips per XCD chunk (LBM_XCD_CHUNK; 0 = plain workgroup order)
  int use_stepk = 0;   // two-step passes through stepk_stream<K=2> instead of step2_stream (LBM_STEPK; experiments)
  int packed = 0;      // stream kernel: collision on pairs of cells, v_pk_* instructions (LBM_PACKED)
  int halo_lanes = 1;  // stream kernel: lanes at each end of a wave that only feed their neighbours
  int lds_windows = 0; // packed stream kernel: how many of the K-1 sliding windows live in LDS (LBM_LDS_WINDOWS, 0..2)
  int band_rows = 8, n_strips = 0;  // step2_stream geometry: band height, waves across x
  int lane_cells = 4;               // cells per lane in step2_stream (4 or 2; LBM_LANE_CELLS)
  SlabTeam* team = nullptr;         // one issuing thread per slab (one-process multi-GPU), or null
  int use_graph = 0;                // replay chunks of an even number of passes + their reduce as one hipGraph each
  GraphBuilder* builder = nullptr;  // non-null while a chunk is being built: launches become graph nodes
  int resident = 0;                 // single periodic slab that fits the chip's registers: lbm_run calls of at least
  int resident_min_steps = 16;      // ... this many timesteps run as launches of the resident kernel (lbm::resident_band)
  int resident_bands = 0;           // its workgroups (bands of resident_rows rows)
  int resident_joint = 0;           // narrow grids: both pairs of a lane relaxed as one block behind the halo wait
  int resident_rows = 4;            // rows per band: 4, or 2 where the chip has CUs to spare (one pair per lane)
  int resident_group = 1;           // bands per workgroup
  int resident_one_xcd = 0;         // all workgroups on one XCD (8 x the workgroups launched, 7 of 8 leave at once)
  long long resident_timeout = 0;   // bound of one halo wait, wall-clock ticks
  bool resident_used = false;       // a launch is in flight / unchecked: lbm_sync reads its status
  int tile_steps = 0;               // > 0: single slab advanced by the LDS-tile kernel, this many steps per launch
  int tile_shape = 0;               // index into kTileShapes
};

namespace {

// run body(s) for every slab: concurrently on the slab team when there is one, else in order
int for_slabs(lbm_ctx* c, const std::function<int(int)>& body) {
  if (c->team) return c->team->run(c->n_slabs, body);
  for (int s = 0; s < c->n_slabs; s++)
    if (body(s) != LBM_SUCCESS) return LBM_FAILURE;
  return LBM_SUCCESS;
}

// ---- stream operations that become graph nodes / edges while a chunk is being built ---------------------------
int q_wait(lbm_ctx* c, hipStream_t st, hipEvent_t ev) {
  if (!c->builder) { HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(st, ev, 0)); return LBM_SUCCESS; }
  const std::vector<hipGraphNode_t>* snap = c->builder->snapshot_of(ev, false);
  if (!snap) return LBM_SUCCESS;  // never recorded inside this chunk: "ready when the chunk starts"
  std::vector<hipGraphNode_t>& last = c->builder->last_of(st);
  for (hipGraphNode_t n : *snap) {
    bool have = false;
    for (hipGraphNode_t m : last) have = have || (m == n);
    if (!have) last.push_back(n);
  }
  return LBM_SUCCESS;
}
int q_record(lbm_ctx* c, hipEvent_t ev, hipStream_t st) {
  if (!c->builder) { HIP_TRY(LBM_FAILURE, hipEventRecord(ev, st)); return LBM_SUCCESS; }
  const std::vector<hipGraphNode_t> now = c->builder->last_of(st);  // copy: snapshot_of may grow the event table
  *c->builder->snapshot_of(ev, true) = now;
  return LBM_SUCCESS;
}
// launch fn(args...) on st; `done` (optional, stream mode): event bound to the kernel's own completion signal
int q_kernel(lbm_ctx* c, hipStream_t st, const void* fn, dim3 grid, dim3 block, void** args, hipEvent_t done = nullptr) {
  if (c->builder) {
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.func = const_cast<void*>(fn);
    kp.gridDim = grid;
    kp.blockDim = block;
    kp.kernelParams = args;
    std::vector<hipGraphNode_t>& last = c->builder->last_of(st);
    hipGraphNode_t node = nullptr;
    HIP_TRY(LBM_FAILURE, hipGraphAddKernelNode(&node, c->builder->graph, last.data(), last.size(), &kp));
    last.assign(1, node);
    if (done) *c->builder->snapshot_of(done, true) = last;
    return LBM_SUCCESS;
  }
  if (done) HIP_TRY(LBM_FAILURE, hipExtLaunchKernel(fn, grid, block, args, 0, st, nullptr, done, 0));
  else HIP_TRY(LBM_FAILURE, hipLaunchKernel(fn, grid, block, args, 0, st));
  return LBM_SUCCESS;
}
int q_copy(lbm_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (!c->builder) { HIP_TRY(LBM_FAILURE, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st)); return LBM_SUCCESS; }
  std::vector<hipGraphNode_t>& last = c->builder->last_of(st);
  hipGraphNode_t node = nullptr;
  HIP_TRY(LBM_FAILURE, hipGraphAddMemcpyNode1D(&node, c->builder->graph, last.data(), last.size(), dst, src, bytes, hipMemcpyDeviceToDevice));
  last.assign(1, node);
  return LBM_SUCCESS;
}

// ---- launch helpers --------------------------------------------------------------------------
// `done` (optional): event bound to the kernel's own completion signal (hipExtLaunchKernel's stop event)
// -- what hipEventRecord right after the launch would mark, without a barrier packet of its own
int launch_step(lbm_ctx* c, int s, hipStream_t stream, int row_first, int row_stride, int n_rows,
                int part_offset, bool accel_epilogue, hipEvent_t done = nullptr) {
  Slab& sl = c->slab[s];
  if (n_rows <= 0) return LBM_SUCCESS;
  lbm::StepArgs a;
  a.src = sl.lat[c->cur];
  a.dst = sl.lat[c->cur ^ 1];
  a.mask = sl.mask;
  a.plane_stride = c->plane_stride;
  a.pitch = c->pitch;
  a.row_pitch = c->row_pitch;
  a.nx = c->p.nx;
  a.rows = sl.rows;
  a.row_first = row_first;
  a.row_stride = row_stride;
  a.n_rows = n_rows;
  a.accel_row = accel_epilogue ? sl.accel_row : lbm::kNoRow;
  a.omega = c->p.omega;
  a.a1 = c->p.density * c->p.accel / 9.f;   // SerialCode/d2q9-bgk.c:219
  a.a2 = c->p.density * c->p.accel / 36.f;  // :220
  a.partials = sl.partials + (long)c->slot_fill * c->part_stride + part_offset;
  a.reverse = (c->snake && n_rows > 2) ? (c->cur & 1) : 0;
  a.wrap = (c->halo == HALO_SELF) ? 1 : 0;

  const bool exact = (c->math_mode == LBM_MATH_EXACT);
  if (c->vec4) {
    const int blocks = ceil_div((long)(c->p.nx / 4) * n_rows, lbm::kBlock);
    // kernel flavour: [math][neighbour exchange][nontemporal stores]; tuned defaults, see DESIGN.md
    typedef void (*step_fn)(const lbm::StepArgs);
    static const step_fn table[2][3][2] = {
        {{lbm::step_vec4<0, 0, false>, lbm::step_vec4<0, 0, true>},
         {lbm::step_vec4<0, 1, false>, lbm::step_vec4<0, 1, true>},
         {lbm::step_vec4<0, 2, false>, lbm::step_vec4<0, 2, true>}},
        {{lbm::step_vec4<1, 0, false>, lbm::step_vec4<1, 0, true>},
         {lbm::step_vec4<1, 1, false>, lbm::step_vec4<1, 1, true>},
         {lbm::step_vec4<1, 2, false>, lbm::step_vec4<1, 2, true>}}};
    void* args[] = {&a};
    return q_kernel(c, stream, reinterpret_cast<const void*>(table[exact ? 0 : 1][c->neigh][c->nts]), dim3(blocks), dim3(lbm::kBlock), args, done);
  } else {
    const int blocks = ceil_div((long)c->p.nx * n_rows, lbm::kBlock);
    const auto fn = exact ? lbm::step_scalar<true> : lbm::step_scalar<false>;
    void* args[] = {&a};
    return q_kernel(c, stream, reinterpret_cast<const void*>(fn), dim3(blocks), dim3(lbm::kBlock), args, done);
  }
}

// two timesteps in one pass over the rows [row_first, row_end) of slab s, cut into band_count bands of
// band_rows rows that start band_pitch rows apart; writes the partials of steps t and t+1 into
// slots slot_fill and slot_fill+1
int launch_step2(lbm_ctx* c, int s, hipStream_t stream, int row_first, int row_end, int band_rows,
                 int band_pitch, int band_count, int part_offset, bool accel_after, hipEvent_t done = nullptr) {
  Slab& sl = c->slab[s];
  if (band_count <= 0) return LBM_SUCCESS;
  lbm::Step2Args a;
  a.src = sl.lat[c->cur];
  a.dst = sl.lat[c->cur ^ 1];
  a.mask = sl.mask;
  a.plane_stride = c->plane_stride;
  a.row_pitch = c->row_pitch;
  a.pitch = c->pitch;
  a.nx = c->p.nx;
  a.rows = sl.rows;
  a.wrap = (c->halo == HALO_SELF) ? 1 : 0;
  a.band_rows = band_rows;
  a.row_first = row_first;
  a.band_pitch = band_pitch;
  a.row_end = row_end;
  a.n_strips = c->n_strips;
  a.accel_row = sl.accel_row;
  a.accel_after = accel_after ? 1 : 0;
  a.omega = c->p.omega;
  a.a1 = c->p.density * c->p.accel / 9.f;
  a.a2 = c->p.density * c->p.accel / 36.f;
  a.partials1 = sl.partials + (long)c->slot_fill * c->part_stride + part_offset;
  a.partials2 = a.partials1 + c->part_stride;
  const int waves = c->n_strips * band_count;
  typedef void (*fn)(const lbm::Step2Args);
  // [math][nontemporal stores][cells per lane: 0 -> 4, 1 -> 2]
  static const fn table[2][2][2] = {
      {{lbm::step2_stream<0, false, 4>, lbm::step2_stream<0, false, 2>},
       {lbm::step2_stream<0, true, 4>, lbm::step2_stream<0, true, 2>}},
      {{lbm::step2_stream<1, false, 4>, lbm::step2_stream<1, false, 2>},
       {lbm::step2_stream<1, true, 4>, lbm::step2_stream<1, true, 2>}}};
  const fn kernel = table[c->math_mode == LBM_MATH_EXACT ? 0 : 1][c->nts][c->lane_cells == 2 ? 1 : 0];
  void* args[] = {&a};
  return q_kernel(c, stream, reinterpret_cast<const void*>(kernel), dim3(waves), dim3(64), args, done);
}

// k (2..4) timesteps in one pass over the rows [row_first, row_end) of slab s (4 cells per lane), cut into band_count
// bands of band_rows rows that start band_pitch rows apart; partials of step t+j go to slot slot_fill + j
int launch_stepk(lbm_ctx* c, int s, hipStream_t stream, int k, int row_first, int row_end, int band_rows,
                 int band_pitch, int band_count, int part_offset, bool accel_after, hipEvent_t done = nullptr) {
  Slab& sl = c->slab[s];
  if (band_count <= 0) return LBM_SUCCESS;
  lbm::StepKArgs a;
  a.src = sl.lat[c->cur];
  a.dst = sl.lat[c->cur ^ 1];
  a.mask = sl.mask;
  a.plane_stride = c->plane_stride;
  a.row_pitch = c->row_pitch;
  a.pitch = c->pitch;
  a.nx = c->p.nx;
  a.rows = sl.rows;
  a.wrap = (c->halo == HALO_SELF) ? 1 : 0;
  a.band_rows = band_rows;
  a.row_first = row_first;
  a.band_pitch = band_pitch;
  a.row_end = row_end;
  a.n_strips = c->n_strips;
  a.n_bands = band_count;
  a.chunk = c->xcd_chunk;
  a.halo_lanes = c->halo_lanes;
  a.accel_row = sl.accel_row;
  a.accel_row2 = sl.accel_row2;
  a.accel_after = accel_after ? 1 : 0;
  a.omega = c->p.omega;
  a.a1 = c->p.density * c->p.accel / 9.f;
  a.a2 = c->p.density * c->p.accel / 36.f;
  a.partials = sl.partials + (long)c->slot_fill * c->part_stride + part_offset;
  a.slot_stride = c->part_stride;
  int waves = c->n_strips * band_count;
  if (a.chunk > 0) {
    const int chunks = band_count * ceil_div(c->n_strips, a.chunk);
    waves = 8 * ceil_div(chunks, 8) * a.chunk;
  }
  typedef void (*fn)(const lbm::StepKArgs);
  // [math][nontemporal stores][k - 2][prefetch]
  // (K = 4 with the next row prefetched does not fit the register file -- 112 bytes of scratch per lane -- so that
  // request runs the kernel without prefetch: same results)
#define LBM_K_ROW(M, N) {{lbm::stepk_stream<M, N, 4, 2, false>, lbm::stepk_stream<M, N, 4, 2, true>}, \
                         {lbm::stepk_stream<M, N, 4, 3, false>, lbm::stepk_stream<M, N, 4, 3, true>}, \
                         {lbm::stepk_stream<M, N, 4, 4, false>, lbm::stepk_stream<M, N, 4, 4, false>}}
  static const fn table[2][2][3][2] = {{LBM_K_ROW(0, false), LBM_K_ROW(0, true)}, {LBM_K_ROW(1, false), LBM_K_ROW(1, true)}};
#undef LBM_K_ROW
  // exact arithmetic on pairs of cells (v_pk_* instructions): [nontemporal stores][k - 2][prefetch][windows in LDS]
  // (K = 4 prefetches only with two of its three windows in LDS: with fewer it spills, and runs without prefetch)
#define LBM_PK(N, KK, PF) {lbm::stepk_pk<N, KK, (PF && KK < 4), 0>, lbm::stepk_pk<N, KK, (PF && KK < 4), 1>, lbm::stepk_pk<N, KK, PF, (KK > 2 ? 2 : 1)>}
#define LBM_PK_ROW(N) {{LBM_PK(N, 2, false), LBM_PK(N, 2, true)}, {LBM_PK(N, 3, false), LBM_PK(N, 3, true)}, \
                       {LBM_PK(N, 4, false), LBM_PK(N, 4, true)}}
  // [nontemporal stores][k - 2][prefetch][windows in LDS]  (the QUAD form of stepk_pk -- both pairs of a lane in one
  // basic block -- measured the same speed with more registers and is not instantiated: profiles/r02_tuning.md)
  static const fn table_pk[2][3][2][3] = {LBM_PK_ROW(false), LBM_PK_ROW(true)};
#undef LBM_PK_ROW
#undef LBM_PK
  // two cells per lane (one pair): [nontemporal stores][k - 2][prefetch][windows in LDS: 0, 1]
#define LBM_PK1(N, KK, PF) {lbm::stepk_pk<N, KK, (PF && KK < 4), 0, false, 1>, lbm::stepk_pk<N, KK, PF, 1, false, 1>}
#define LBM_PK1_ROW(N) {{LBM_PK1(N, 2, false), LBM_PK1(N, 2, true)}, {LBM_PK1(N, 3, false), LBM_PK1(N, 3, true)}, \
                        {LBM_PK1(N, 4, false), LBM_PK1(N, 4, true)}}
  static const fn table_pk1[2][3][2][2] = {LBM_PK1_ROW(false), LBM_PK1_ROW(true)};
#undef LBM_PK1_ROW
#undef LBM_PK1
  const int lds_windows = c->lds_windows < k ? c->lds_windows : k - 1;
  const bool packed = c->packed != 0;  // the packed kernels have one arithmetic (the exact one) and serve both math modes
  const fn kernel = (packed && c->lane_cells == 2) ? table_pk1[c->nts][k - 2][c->prefetch ? 1 : 0][lds_windows ? 1 : 0]
                    : packed ? table_pk[c->nts][k - 2][c->prefetch ? 1 : 0][lds_windows]
                           : table[c->math_mode == LBM_MATH_EXACT ? 0 : 1][c->nts][k - 2][c->prefetch ? 1 : 0];
  void* args[] = {&a};
  return q_kernel(c, stream, reinterpret_cast<const void*>(kernel), dim3(waves), dim3(64), args, done);
}

// the stream kernel for a k-step pass: the 2-cells-per-lane form exists for k = 2 only (step2_stream)
int launch_pass(lbm_ctx* c, int s, hipStream_t stream, int k, int row_first, int row_end, int band_rows,
                int band_pitch, int band_count, int part_offset, bool accel_after, hipEvent_t done = nullptr) {
  if ((c->lane_cells == 4 && (k > 2 || c->prefetch || c->xcd_chunk || c->use_stepk || c->packed)) ||
      (c->lane_cells == 2 && c->packed))
    return launch_stepk(c, s, stream, k, row_first, row_end, band_rows, band_pitch, band_count, part_offset, accel_after, done);
  return launch_step2(c, s, stream, row_first, row_end, band_rows, band_pitch, band_count, part_offset, accel_after, done);
}

int tile_count_for(const lbm_params* p, int shape) {
  return ceil_div(p->nx, kTileShapes[shape].tw) * ceil_div(p->ny, kTileShapes[shape].th);
}
int tile_count(const lbm_ctx* c) {
  const TileShape& t = kTileShapes[c->tile_shape];
  return ceil_div(c->p.nx, t.tw) * ceil_div(c->slab[0].rows, t.th);
}

// n_steps <= c->tile_steps timesteps of the whole (single, periodic) slab from LDS tiles; partials of step j go
// to slot slot_fill + j
int launch_tile(lbm_ctx* c, hipStream_t stream, int n_steps, bool accel_after) {
  Slab& sl = c->slab[0];
  lbm::TileArgs a;
  a.src = sl.lat[c->cur];
  a.dst = sl.lat[c->cur ^ 1];
  a.mask = sl.mask;
  a.plane_stride = c->plane_stride;
  a.row_pitch = c->row_pitch;
  a.pitch = c->pitch;
  a.nx = c->p.nx;
  a.ny = sl.rows;
  a.n_steps = n_steps;
  a.accel_row = sl.accel_row;
  a.accel_after = accel_after ? 1 : 0;
  a.omega = c->p.omega;
  a.a1 = c->p.density * c->p.accel / 9.f;
  a.a2 = c->p.density * c->p.accel / 36.f;
  a.partials = sl.partials + (long)c->slot_fill * c->part_stride;
  a.slot_stride = c->part_stride;
  const TileShape& t = kTileShapes[c->tile_shape];
  a.tiles_x = ceil_div(c->p.nx, t.tw);
  void* args[] = {&a};
  return q_kernel(c, stream, reinterpret_cast<const void*>(c->math_mode == LBM_MATH_EXACT ? t.exact : t.fast), dim3(tile_count(c)),
                  dim3(t.threads), args);
}

int blocks_for_rows(const lbm_ctx* c, int n_rows) {
  if (n_rows <= 0) return 0;
  return c->vec4 ? ceil_div((long)(c->p.nx / 4) * n_rows, lbm::kBlock)
                 : ceil_div((long)c->p.nx * n_rows, lbm::kBlock);
}

// One halo exchange, enqueued on the comm streams: the `depth` boundary rows at each end of every
// slab's lattice `src` travel, whole (all 9 speeds, as MPI_Waitall/d2q9-bgk.c:225-230 ships
// them), into the halo rows of its ring neighbours' lattices `dst` -- zero copy on both sides,
// because halo rows and boundary rows are contiguous in the row-interleaved layout.
//   my rows [rows-depth, rows)  ->  north neighbour's rows [-depth, 0)
//   my rows [0, depth)          ->  south neighbour's rows [rows_s, rows_s + depth)
// north neighbour of slab s = s+1 (periodic), south = s-1; across processes the ring runs over
// ranks (MPI/d2q9-bgk.c:210-211).
// Synchronous pipeline (slot < 0): src == dst == the current lattice.  Precondition (stream order):
// the kernels that wrote the boundary rows are ordered before this on the comm stream.
// Stale-halo pipeline (slot 0/1): src = the lattice just produced, dst = the lattice the pass after
// next reads.  The comm stream first waits for ev_step of this slab (boundary rows written) and, where
// this slab writes into its neighbours' memory itself (memcpy), of the neighbours (they have finished
// reading the halo rows about to be overwritten); ev_x[slot] marks the arrival.
int exchange_halos(lbm_ctx* c, int depth, int src, int dst, int slot) {
  const long n = (long)depth * c->row_pitch;
  const bool stale = (slot >= 0);
  if (c->halo == HALO_RCCL) {
    if (stale) {
      for (int s = 0; s < c->n_slabs; s++) {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        // a peer's receive is posted behind the peer's own ev_step wait, so nothing lands in halo
        // rows a running pass still reads
        HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sl.ev_step, 0));
      }
    }
    // one thread driving several communicators must group them; with one thread per slab each
    // thread groups its own four operations.  A group once opened is closed on every path: an error inside it must
    // not leave the communicator in group mode.
    RCCL_OR_FAIL(LBM_FAILURE);
    const RcclApi& nc = *rc_api_;
    // while a chunk is being built: the group is captured on the (single) comm stream alone -- a capture with one
    // user stream, its origin -- and enters the chunk as a child-graph node behind the comm stream's dependencies
    const bool building = (c->builder != nullptr);
    if (building) {
      if (c->n_slabs != 1) LBM_FAIL(LBM_FAILURE, "hipGraph chunk: the RCCL transport is built for one communicator per process");
      HIP_TRY(LBM_FAILURE, hipStreamBeginCapture(c->slab[0].comm, hipStreamCaptureModeRelaxed));
    }
    if (!c->team) NCCL_TRY(LBM_FAILURE, nc.GroupStart());
    int rc = for_slabs(c, [&](int s) -> int {
      Slab& sl = c->slab[s];
      float* from = sl.lat[src];
      float* to = sl.lat[dst];
      int me, parts;
      if (c->ranked) { me = c->rank; parts = c->world; } else { me = s; parts = c->n_slabs; }
      // the four operations in the posting order of lbm_halo_plan (the order matters when north == south, 2 parts:
      // the first send pairs with the peer's first receive)
      lbm_halo_op ops[4];
      if (lbm_halo_plan(sl.rows, parts, me, depth, ops) != LBM_SUCCESS) return LBM_FAILURE;
      if (c->team) {
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        NCCL_TRY(LBM_FAILURE, nc.GroupStart());
      }
      ncclResult_t res = ncclSuccess;
      for (int i = 0; i < 4 && res == ncclSuccess; i++) {
        float* base = ops[i].is_send ? from : to;
        float* ptr = base + (long)ops[i].row_first * c->row_pitch;
        const size_t count = (size_t)ops[i].row_count * c->row_pitch;
        res = ops[i].is_send ? nc.Send(ptr, count, ncclFloat, ops[i].peer, sl.nccl, sl.comm)
                             : nc.Recv(ptr, count, ncclFloat, ops[i].peer, sl.nccl, sl.comm);
      }
      if (c->team) {
        const ncclResult_t end = nc.GroupEnd();
        if (res == ncclSuccess) res = end;
      }
      if (res != ncclSuccess) LBM_FAIL(LBM_FAILURE, "RCCL error in the halo exchange: %s", nc.GetErrorString(res));
      return LBM_SUCCESS;
    });
    if (!c->team) {
      const ncclResult_t end = nc.GroupEnd();
      if (rc == LBM_SUCCESS && end != ncclSuccess) { raise_error(__LINE__, "RCCL error: %s (ncclGroupEnd)", nc.GetErrorString(end)); rc = LBM_FAILURE; }
    }
    if (building) {
      hipGraph_t child = nullptr;
      const hipError_t ended = hipStreamEndCapture(c->slab[0].comm, &child);
      if (rc == LBM_SUCCESS && ended != hipSuccess) { raise_error(__LINE__, "HIP error: %s (capture of the RCCL group)", hipGetErrorString(ended)); rc = LBM_FAILURE; }
      if (rc == LBM_SUCCESS) {
        std::vector<hipGraphNode_t>& last = c->builder->last_of(c->slab[0].comm);
        hipGraphNode_t node = nullptr;
        const hipError_t added = hipGraphAddChildGraphNode(&node, c->builder->graph, last.data(), last.size(), child);
        if (added != hipSuccess) { raise_error(__LINE__, "HIP error: %s (hipGraphAddChildGraphNode)", hipGetErrorString(added)); rc = LBM_FAILURE; }
        else last.assign(1, node);
      }
      if (child) (void)hipGraphDestroy(child);  // the node holds its own copy
    }
    if (rc != LBM_SUCCESS) return rc;
    if (stale) {
      for (int s = 0; s < c->n_slabs; s++) {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_x[slot], sl.comm));
      }
    }
    return LBM_SUCCESS;
  }
  if (c->halo == HALO_HOST) {
    // the host's own message passing (MPI in the reference: MPI_Isend / MPI_Irecv / MPI_Waitall, MPI_Waitall/
    // d2q9-bgk.c:225-243): boundary rows to pinned host buffers, the host's exchange callback, halo rows back.
    // The callback blocks the host, so this transport does not overlap the interior rows -- it exists so that the
    // decomposition can run under an MPI-style launcher and be tested with several ranks on one device.
    if (stale) LBM_FAIL(LBM_FAILURE, "the stale-halo mode is not available with the hosted exchange");
    Slab& sl = c->slab[0];
    lbm_halo_op ops[4];
    if (lbm_halo_plan(sl.rows, c->world, c->rank, depth, ops) != LBM_SUCCESS) return LBM_FAILURE;
    float* bufs[4] = {c->host_send[0], c->host_send[1], c->host_recv[0], c->host_recv[1]};
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    for (int i = 0; i < 2; i++)
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(bufs[i], sl.lat[src] + (long)ops[i].row_first * c->row_pitch, (size_t)n * sizeof(float),
                                          hipMemcpyDeviceToHost, sl.comm));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.comm));
    if (c->host_comm.exchange(c->host_comm.user, 4, ops, bufs, (size_t)n) != 0)
      LBM_FAIL(LBM_FAILURE, "the host's halo exchange callback failed");
    for (int i = 2; i < 4; i++)
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(sl.lat[dst] + (long)ops[i].row_first * c->row_pitch, bufs[i], (size_t)n * sizeof(float),
                                          hipMemcpyHostToDevice, sl.comm));
    return LBM_SUCCESS;
  }
  if (c->halo == HALO_MEMCPY) {
    // push model inside one process: slab s copies its boundary rows into its neighbours' halo rows
    // once the neighbours have finished with the previous contents (synchronous: their boundary
    // kernels, ev_boundary, recorded in the previous phase; stale: their whole-slab pass, ev_step);
    // the neighbours' next halo-reading kernels wait for ev_halo / ev_x[slot] of the pushing slabs.
    return for_slabs(c, [&](int s) -> int {
      Slab& sl = c->slab[s];
      const int north = (s + 1) % c->n_slabs, south = (s - 1 + c->n_slabs) % c->n_slabs;
      float* from = sl.lat[src];
      Slab& sn = c->slab[north];
      Slab& ss = c->slab[south];
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      if (stale && q_wait(c, sl.comm, sl.ev_step) != LBM_SUCCESS) return LBM_FAILURE;
      if (q_wait(c, sl.comm, stale ? sn.ev_step : sn.ev_boundary) != LBM_SUCCESS) return LBM_FAILURE;
      if (q_wait(c, sl.comm, stale ? ss.ev_step : ss.ev_boundary) != LBM_SUCCESS) return LBM_FAILURE;
      if (q_copy(c, sn.lat[dst] - n, from + (long)(sl.rows - depth) * c->row_pitch, n * sizeof(float), sl.comm) != LBM_SUCCESS) return LBM_FAILURE;
      if (q_copy(c, ss.lat[dst] + (long)ss.rows * c->row_pitch, from, n * sizeof(float), sl.comm) != LBM_SUCCESS) return LBM_FAILURE;
      return q_record(c, stale ? sl.ev_x[slot] : sl.ev_halo, sl.comm);
    });
  }
  return LBM_SUCCESS;
}

// reduce the buffered per-workgroup partials of the last slot_fill steps into tot_u[step_base...]
int flush_partials(lbm_ctx* c, int step_base) {
  if (c->slot_fill == 0) return LBM_SUCCESS;
  return for_slabs(c, [&](int s) -> int {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    // the boundary rows' partials are written on the comm stream
    const bool split = (c->halo != HALO_SELF && c->halo_mode == LBM_HALO_SYNC);
    if (split) HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, sl.ev_boundary, 0));
    hipLaunchKernelGGL(lbm::reduce_partials, dim3(c->slot_fill), dim3(lbm::kBlock), 0, sl.compute,
                       sl.partials, sl.slot_counts, c->part_stride, sl.tot_u, step_base, (const int*)nullptr);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
    if (split) {
      // the next boundary kernels (comm stream) reuse the partial slots just read
      HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_flush, sl.compute));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sl.ev_flush, 0));
    }
    return LBM_SUCCESS;
  });
}

// the slab threads spin between phases while a run is in flight
struct HotGuard {
  SlabTeam* t;
  explicit HotGuard(SlabTeam* team) : t(team) { if (t) { t->hot.store(true); t->cv.notify_all(); } }
  ~HotGuard() { if (t) t->hot.store(false); }
};

// device time per timestep between ev_t0 (recorded by the caller at the start of the run) and now, on
// the compute streams the step kernels run on; max over slabs
int read_step_timing(lbm_ctx* c, int n_steps, float* kernel_ms) {
  float worst = 0.f;
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_t1, sl.compute));
  }
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipEventSynchronize(sl.ev_t1));
    float ms = 0.f;
    HIP_TRY(LBM_FAILURE, hipEventElapsedTime(&ms, sl.ev_t0, sl.ev_t1));
    if (ms > worst) worst = ms;
  }
  *kernel_ms = worst / (float)n_steps;
  return LBM_SUCCESS;
}

// The timestep loop.  Single slab: one fused launch per pass (one to four timesteps).  Several slabs / ranks
// (the Waitall pattern of MPI_Waitall/d2q9-bgk.c:225-253, restructured for two HIP streams):
//
//   compute stream:  I(0) ─────────────► I(1) ─────────────► I(2) ...     rows that touch no halo row
//                      ▲ waits B(m-1)      ▲
//   comm stream:     X(0) → B(0) → X(1) → B(1) → X(2) → B(2) ...          halo exchange, halo-touching rows
//                            ▲ waits I(m-1)
//
// I(m) and B(m) both read lattice m and write disjoint rows of lattice m+1; B(m) additionally
// needs the halos X(m) (its own stream, in order) and writes the boundary rows X(m+1) sends.  The
// chain of interior kernels is the critical path; exchange and boundary rows hide beside it.
//
// K-step passes across slabs: an output row y reads source rows y-K .. y+K, so only rows [0, K) and
// [rows-K, rows) touch halo rows.  They form two K-row bands (short sweeps: low latency on the comm
// stream); rows [K, rows-K) are the interior region, cut into bands of band_rows.

// Pipeline events in a defined state: "I(-1)" = the lattice is ready (also orders the comm stream after everything
// on the compute stream), "B(-1)" done.  Called at the start of a run and after a graph replay (events recorded
// inside a stream capture cannot be waited for outside it).
int reset_pipeline(lbm_ctx* c) {
  return for_slabs(c, [&](int s) -> int {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_interior[1], sl.compute));
    HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sl.ev_interior[1], 0));
    HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_boundary, sl.comm));
    if (c->halo == HALO_MEMCPY) HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_halo, sl.comm));
    return LBM_SUCCESS;
  });
}

// Pass m of the pipeline: X(m) (with halos), I(m), B(m), and the bookkeeping of the partial slots it fills.
//   tile > 0: that many timesteps of the LDS-tile kernel;  k >= 2: a k-step pass of the stream kernel;  else one step.
//   accel_after: apply the acceleration of the step after this pass (false on the last pass of an lbm_run call).
//   bound_events: bind the pipeline events to the kernels' own completion signals (hipExtLaunchKernel); not inside a
//   stream capture, where hipEventRecord costs nothing (it becomes a graph edge).
int issue_pass(lbm_ctx* c, int m, int tile, int k, bool accel_after, bool bound_events) {
  const bool halo = (c->halo != HALO_SELF);
  const int adv = tile ? tile : (k ? k : 1);
  const int depth = c->fuse2 ? c->pass_steps : 1;  // a K-step pass reads K rows beyond the slab
  if (halo && exchange_halos(c, depth, c->cur, c->cur, -1) != LBM_SUCCESS) return LBM_FAILURE;
  // phase 1: rows that touch no halo row (or the whole slab) on the compute streams
  if (for_slabs(c, [&](int s) -> int {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        if (halo && m > 0 && q_wait(c, sl.compute, sl.ev_boundary) != LBM_SUCCESS) return LBM_FAILURE;  // B(m-1)
        hipEvent_t done = (halo && bound_events) ? sl.ev_interior[m & 1] : nullptr;  // I(m) done
        if (tile) {
          if (launch_tile(c, sl.compute, tile, accel_after) != LBM_SUCCESS) return LBM_FAILURE;
        } else if (k) {
          const int r0 = halo ? k : 0, r1 = halo ? sl.rows - k : sl.rows;
          if (launch_pass(c, s, sl.compute, k, r0, r1, c->band_rows, c->band_rows, ceil_div(r1 - r0, c->band_rows), 0,
                          accel_after, done) != LBM_SUCCESS)
            return LBM_FAILURE;
        } else if (!halo) {
          if (launch_step(c, s, sl.compute, 0, 1, sl.rows, 0, accel_after) != LBM_SUCCESS) return LBM_FAILURE;
        } else {
          if (launch_step(c, s, sl.compute, 1, 1, sl.rows - 2, 0, accel_after, done) != LBM_SUCCESS) return LBM_FAILURE;
        }
        if (halo && !done) return q_record(c, sl.ev_interior[m & 1], sl.compute);
        return LBM_SUCCESS;
      }) != LBM_SUCCESS)
    return LBM_FAILURE;
  // phase 2: halo-touching rows / bands on the comm streams, behind the exchange X(m)
  if (halo &&
      for_slabs(c, [&](int s) -> int {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        if (q_wait(c, sl.comm, sl.ev_interior[(m + 1) & 1]) != LBM_SUCCESS) return LBM_FAILURE;  // I(m-1)
        if (c->halo == HALO_MEMCPY) {
          const int north = (s + 1) % c->n_slabs, south = (s - 1 + c->n_slabs) % c->n_slabs;
          if (q_wait(c, sl.comm, c->slab[north].ev_halo) != LBM_SUCCESS) return LBM_FAILURE;
          if (q_wait(c, sl.comm, c->slab[south].ev_halo) != LBM_SUCCESS) return LBM_FAILURE;
        }
        hipEvent_t bdone = bound_events ? sl.ev_boundary : nullptr;  // B(m) done
        if (k) {
          // rows [0, k) and [rows-k, rows) as two k-row bands in one launch
          const int off = c->n_strips * ceil_div(sl.rows - 2 * k, c->band_rows);
          if (launch_pass(c, s, sl.comm, k, 0, sl.rows, k, sl.rows - k, 2, off, accel_after, bdone) != LBM_SUCCESS) return LBM_FAILURE;
        } else {
          if (launch_step(c, s, sl.comm, 0, sl.rows - 1, 2, sl.blocks_main, accel_after, bdone) != LBM_SUCCESS) return LBM_FAILURE;
        }
        if (!bdone) return q_record(c, sl.ev_boundary, sl.comm);
        return LBM_SUCCESS;
      }) != LBM_SUCCESS)
    return LBM_FAILURE;
  // bookkeeping of the partial slots written by this pass
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    const int fused_waves = halo ? c->n_strips * (ceil_div(sl.rows - 2 * k, c->band_rows) + 2)
                                 : c->n_strips * ceil_div(sl.rows, c->band_rows);
    const int n_part = tile ? tile_count(c) : (k ? fused_waves : sl.blocks_main + sl.blocks_boundary);
    for (int j = 0; j < adv; j++) sl.slot_counts.n[c->slot_fill + j] = n_part;
  }
  c->cur ^= 1;
  c->slot_fill += adv;
  return LBM_SUCCESS;
}

// ---- hipGraph replay of the timestep loop ---------------------------------------------------------------------
// A chunk = an even number of passes (so that it starts and ends on the same lattice buffer) and the reduce of
// their partial sums, built ONCE per lattice parity -- both streams of every slab, the halo exchange (RCCL
// send/recv or device copies) inside the graph -- and replayed with one hipGraphLaunch (BASELINE.json
// configs[4]: "double-buffered halos + hipGraph-captured timestep").  The graph is built node by node from the very
// issue code of the stream pipeline (GraphBuilder: virtual streams and events); only an RCCL group is captured, on
// its single comm stream, and enters as a child graph.  All launch arguments of a chunk are the same
// every time except the index of the chunk's first step in tot_u, which the reduce kernel reads from device memory.
// Every pass of a chunk applies the next step's acceleration, so a chunk is only replayed while at least one more
// timestep follows it in the same lbm_run call.  A chunk begins with its own exchange X(0) and ends with every
// stream joined, i.e. one exchange per chunk is not hidden behind interior rows (1 of 20-32).
int chunk_passes(const lbm_ctx* c, int* steps_per_pass) {
  const int halo = (c->halo != HALO_SELF);
  const int adv = (!halo && c->tile_steps) ? c->tile_steps : (c->fuse2 ? c->pass_steps : 1);
  int passes = kPartSlots / adv;
  const int cap = env_int("LBM_GRAPH_PASSES", 0);  // experiments and tests: shorter chunks
  if (cap > 0 && passes > cap) passes = cap;
  passes -= passes & 1;
  *steps_per_pass = adv;
  return passes;
}

// LBM_GRAPH_DUMP=<path>: the chunk's graph, checked before hipGraphInstantiate sees it -- node and edge
// counts, self-edges, duplicate edges, and a topological sort (Kahn) that reports whether the graph is acyclic;
// the graph itself goes to <path> as a dot file.  Returns false when the graph must not be instantiated.
bool graph_is_sound(hipGraph_t graph, const char* dump_path) {
  size_t n_nodes = 0, n_edges = 0;
  if (hipGraphGetNodes(graph, nullptr, &n_nodes) != hipSuccess || hipGraphGetEdges(graph, nullptr, nullptr, &n_edges) != hipSuccess)
    return true;  // cannot look: leave the verdict to the runtime
  if (dump_path) fprintf(stderr, "lbm_hip graph: %zu nodes, %zu edges reported\n", n_nodes, n_edges);
  std::vector<hipGraphNode_t> nodes(n_nodes), from(n_edges), to(n_edges);
  if (n_nodes && hipGraphGetNodes(graph, nodes.data(), &n_nodes) != hipSuccess) return true;
  if (n_edges && hipGraphGetEdges(graph, from.data(), to.data(), &n_edges) != hipSuccess) return true;
  auto index_of = [&](hipGraphNode_t x) -> long {
    for (size_t i = 0; i < n_nodes; i++) if (nodes[i] == x) return (long)i;
    return -1;
  };
  std::vector<long> a(n_edges), b(n_edges);
  std::vector<int> indeg(n_nodes, 0);
  size_t self_edges = 0, dup_edges = 0, foreign = 0;
  for (size_t e = 0; e < n_edges; e++) {
    a[e] = index_of(from[e]);
    b[e] = index_of(to[e]);
    if (a[e] < 0 || b[e] < 0) { foreign++; continue; }
    if (a[e] == b[e]) self_edges++;
    for (size_t f = 0; f < e; f++) if (a[f] == a[e] && b[f] == b[e]) { dup_edges++; break; }
    indeg[(size_t)b[e]]++;
  }
  // Kahn: every node of an acyclic graph is eventually freed
  std::vector<long> ready;
  for (size_t i = 0; i < n_nodes; i++) if (indeg[i] == 0) ready.push_back((long)i);
  size_t sorted = 0;
  while (!ready.empty()) {
    const long v = ready.back();
    ready.pop_back();
    sorted++;
    for (size_t e = 0; e < n_edges; e++)
      if (a[e] == v && b[e] >= 0 && --indeg[(size_t)b[e]] == 0) ready.push_back(b[e]);
  }
  const bool acyclic = (sorted == n_nodes);
  if (dump_path) {
    fprintf(stderr, "lbm_hip: chunk graph: %zu nodes, %zu edges, %zu self-edges, %zu duplicate edges, %zu edges to foreign nodes, %s\n",
            n_nodes, n_edges, self_edges, dup_edges, foreign, acyclic ? "acyclic" : "CYCLIC");
    if (*dump_path && hipGraphDebugDotPrint(graph, dump_path, hipGraphDebugDotFlagsVerbose) != hipSuccess)
      fprintf(stderr, "lbm_hip: hipGraphDebugDotPrint(%s) failed\n", dump_path);
  }
  return acyclic && self_edges == 0 && foreign == 0;
}

int build_chunk(lbm_ctx* c, hipGraphExec_t* out) {
  const bool halo = (c->halo != HALO_SELF);
  int adv;
  const int passes = chunk_passes(c, &adv);
  if (passes < 2) LBM_FAIL(LBM_FAILURE, "hipGraph chunk: no even number of passes fits");
  for (int s = 1; s < c->n_slabs; s++)
    if (c->slab[s].device != c->slab[0].device) LBM_FAIL(LBM_FAILURE, "hipGraph chunk: the slabs of one graph must share a device");
  const int tile = (!halo && c->tile_steps) ? c->tile_steps : 0;
  const int k = (!tile && c->fuse2) ? c->pass_steps : 0;
  const int saved_cur = c->cur, saved_fill = c->slot_fill;
  GraphBuilder gb;
  HIP_TRY(LBM_FAILURE, hipSetDevice(c->slab[0].device));
  HIP_TRY(LBM_FAILURE, hipGraphCreate(&gb.graph, 0));
  c->builder = &gb;
  c->slot_fill = 0;
  // Nothing is recorded yet: the first waits of pass 0 (I(-1), B(-1), the previous exchange) find no snapshot and
  // add no dependency -- "ready when the chunk starts", which is what the launch stream's order guarantees.
  int rc = LBM_SUCCESS;
  for (int m = 0; m < passes && rc == LBM_SUCCESS; m++) rc = issue_pass(c, m, tile, k, true, false);
  for (int s = 0; s < c->n_slabs && rc == LBM_SUCCESS; s++) {
    Slab& sl = c->slab[s];
    if (halo) rc = q_wait(c, sl.compute, sl.ev_boundary);  // the boundary rows' partials
    const float* partials = sl.partials;
    long stride = c->part_stride;
    double* tot_u = sl.tot_u;
    int zero = 0, fill = c->slot_fill;
    const int* base_dev = sl.flushed_dev;
    int* counter = sl.flushed_dev;
    void* reduce_args[] = {&partials, &sl.slot_counts, &stride, &tot_u, &zero, &base_dev};
    if (rc == LBM_SUCCESS)
      rc = q_kernel(c, sl.compute, reinterpret_cast<const void*>(lbm::reduce_partials), dim3(c->slot_fill), dim3(lbm::kBlock), reduce_args);
    void* advance_args[] = {&counter, &fill};
    if (rc == LBM_SUCCESS)
      rc = q_kernel(c, sl.compute, reinterpret_cast<const void*>(lbm::advance_counter), dim3(1), dim3(1), advance_args);
  }
  c->builder = nullptr;
  c->cur = saved_cur;  // an even number of passes
  c->slot_fill = saved_fill;
  if (rc != LBM_SUCCESS) { (void)hipGraphDestroy(gb.graph); return LBM_FAILURE; }
  const char* dump_path = getenv("LBM_GRAPH_DUMP");
  if ((dump_path && !graph_is_sound(gb.graph, dump_path)) || env_int("LBM_GRAPH_DUMP_ONLY", 0)) {
    (void)hipGraphDestroy(gb.graph);
    LBM_FAIL(LBM_FAILURE, "hipGraph chunk: the graph is not instantiated (unsound, or LBM_GRAPH_DUMP_ONLY)");
  }
  const hipError_t inst = hipGraphInstantiate(out, gb.graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(gb.graph);
  HIP_TRY(LBM_FAILURE, inst);
  return LBM_SUCCESS;
}

// replays as many whole chunks as fit in front of the last timestep of this call; returns the timesteps done
int replay_chunks(lbm_ctx* c, int n_steps, int first_step, int* done) {
  *done = 0;
  int adv;
  const int chunk_steps = chunk_passes(c, &adv) * adv;
  if (chunk_steps <= 0) return LBM_SUCCESS;
  const int n_chunks = (n_steps - 1) / chunk_steps;
  if (n_chunks <= 0 || c->slot_fill != 0) return LBM_SUCCESS;
  Slab& s0 = c->slab[0];
  hipGraphExec_t& exec = s0.chunk_graph[c->cur];
  if (!exec && build_chunk(c, &exec) != LBM_SUCCESS) {
    // e.g. slabs on several devices: go on launch by launch
    fprintf(stderr, "lbm_hip: hipGraph chunk not built (%s); continuing with stream launches\n", g_last_error);
    exec = nullptr;
    c->use_graph = 0;
    return LBM_SUCCESS;
  }
  // everything enqueued so far on the other streams precedes the chunks, which run "on" slab 0's compute stream
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    hipLaunchKernelGGL(lbm::set_counter, dim3(1), dim3(1), 0, sl.compute, sl.flushed_dev, first_step);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
    if (s > 0) {
      HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_step, sl.compute));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(s0.compute, sl.ev_step, 0));
    }
    if (c->halo != HALO_SELF) {
      HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_flush, sl.comm));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(s0.compute, sl.ev_flush, 0));
    }
  }
  HIP_TRY(LBM_FAILURE, hipSetDevice(s0.device));
  for (int k = 0; k < n_chunks; k++) HIP_TRY(LBM_FAILURE, hipGraphLaunch(exec, s0.compute));
  // ... and everything that follows on the other streams comes after them
  if (c->n_slabs > 1 || c->halo != HALO_SELF) {
    HIP_TRY(LBM_FAILURE, hipEventRecord(s0.ev_fork, s0.compute));
    for (int s = 0; s < c->n_slabs; s++) {
      Slab& sl = c->slab[s];
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      if (s > 0) HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, s0.ev_fork, 0));
      if (c->halo != HALO_SELF) HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, s0.ev_fork, 0));
    }
  }
  *done = n_chunks * chunk_steps;
  return LBM_SUCCESS;
}

int run_steps_stale(lbm_ctx* c, int n_steps, float* kernel_ms);

// seam granules: [2 directions][bands][2 slots][nx] + one XCC-id granule per band
size_t resident_gran_bytes(const lbm_ctx* c) {
  return (2UL * c->resident_bands * 2 * c->p.nx + c->resident_bands) * sizeof(uint4);
}

const void* resident_kernel(int nx, int rows, int joint) {
  if (rows == 2) return (nx > 512) ? reinterpret_cast<const void*>(lbm::resident_band<1024, false, 2>)
                                   : reinterpret_cast<const void*>(lbm::resident_band<512, false, 2>);
  if (nx > 512) return reinterpret_cast<const void*>(lbm::resident_band<1024>);
  return joint ? reinterpret_cast<const void*>(lbm::resident_band<512, true>) : reinterpret_cast<const void*>(lbm::resident_band<512>);
}

// The timestep loop of a cache-resident single slab: launches of lbm::resident_band, each advancing up to
// kResidentChunk timesteps with the lattice in registers (SerialCode/d2q9-bgk.c:166-170 as ONE launch), each followed
// by the reduce of its per-band partial sums.  The caller has applied the first step's accelerate_flow.
int run_resident(lbm_ctx* c, int n_steps) {
  Slab& sl = c->slab[0];
  HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
  for (int t = 0; t < n_steps;) {
    const int n = (n_steps - t < kResidentChunk) ? n_steps - t : kResidentChunk;
    lbm::ResidentArgs a;
    a.src = sl.lat[c->cur];
    a.dst = sl.lat[c->cur ^ 1];
    a.mask = sl.mask;
    a.plane_stride = c->plane_stride;
    a.row_pitch = c->row_pitch;
    a.pitch = c->pitch;
    a.nx = c->p.nx;
    a.ny = sl.rows;
    a.n_steps = n;
    a.accel_row = sl.accel_row;
    a.accel_last = (t + n < n_steps) ? 1 : 0;
    a.omega = c->p.omega;
    a.a1 = c->p.density * c->p.accel / 9.f;
    a.a2 = c->p.density * c->p.accel / 36.f;
    a.gran = sl.res_gran;
    a.gran_bytes = (unsigned)resident_gran_bytes(c);
    a.xcd_affinity = env_int("LBM_RESIDENT_XCD", 1) ? 1 : 0;
    a.epoch0 = (unsigned)(c->steps_done + t);
    a.partials = sl.res_part;
    a.status = sl.res_status;
    a.timeout_ticks = c->resident_timeout;
    a.absent_band = env_int("LBM_RESIDENT_ABSENT_BAND", -1);  // tests of the give-up path
#ifdef LBM_RESIDENT_PROFILE
    static long long* prof_dev = nullptr;
    if (!prof_dev) HIP_TRY(LBM_FAILURE, hipMalloc(&prof_dev, 1024 * 8 * sizeof(long long)));
    a.prof = prof_dev;
#endif
    a.group = c->resident_group;
    a.one_xcd = c->resident_one_xcd;
    void* args[] = {&a};
    HIP_TRY(LBM_FAILURE, hipLaunchKernel(resident_kernel(c->p.nx, c->resident_rows, c->resident_joint),
                                         dim3(c->resident_bands / a.group * (a.one_xcd ? 8 : 1)), dim3(c->p.nx * a.group), args, 0, sl.compute));
    hipLaunchKernelGGL(lbm::reduce_band_partials, dim3(n), dim3(64), 0, sl.compute, (const float*)sl.res_part,
                       c->resident_bands, sl.tot_u, c->steps_done + t);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
#ifdef LBM_RESIDENT_PROFILE
    {
      static const char* const phase[8] = {"edges->LDS", "barrier", "shifts(+interior)", "first halo answer", "further polls", "collision", "publish+sum", "extra polls (count)"};
      std::vector<long long> h(c->resident_bands * 8);
      HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
      HIP_TRY(LBM_FAILURE, hipMemcpy(h.data(), prof_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      fprintf(stderr, "resident profile %dx%d rows %d, %d steps (s_memtime ticks per step, wave 0 of each band: mean / min / max over bands)\n", c->p.nx, sl.rows, c->resident_rows, n);
      for (int i = 0; i < 8; i++) {
        double sum = 0, lo = 1e30, hi = 0;
        for (int b = 0; b < c->resident_bands; b++) { const double v = (double)h[b * 8 + i] / n; sum += v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        fprintf(stderr, "  %-20s %9.2f %9.2f %9.2f\n", phase[i], sum / c->resident_bands, lo, hi);
      }
      // the bands that wait least for their neighbours set the pace: who are they, and where does their time go?
      std::vector<int> order(c->resident_bands);
      for (int b = 0; b < c->resident_bands; b++) order[b] = b;
      std::sort(order.begin(), order.end(), [&](int x, int y) { return h[x * 8 + 4] < h[y * 8 + 4]; });
      for (int k = 0; k < 6 && k < c->resident_bands; k++) {
        const int b = order[k];
        fprintf(stderr, "  band %3d:", b);
        for (int i = 0; i < 7; i++) fprintf(stderr, " %8.1f", (double)h[b * 8 + i] / n);
        fprintf(stderr, "\n");
      }
    }
#endif
    c->cur ^= 1;
    t += n;
  }
  // the verdict of these launches travels to the host behind them; lbm_sync looks at it
  HIP_TRY(LBM_FAILURE, hipMemcpyAsync(sl.res_status_host, sl.res_status, sizeof(int), hipMemcpyDeviceToHost, sl.compute));
  c->resident_used = true;
  return LBM_SUCCESS;
}

int run_steps(lbm_ctx* c, int n_steps, float* kernel_ms) {
  if (!c) LBM_FAIL(LBM_FAILURE, "lbm_run: null context");
  if (n_steps < 0) LBM_FAIL(LBM_FAILURE, "lbm_run: negative step count");
  if (kernel_ms) *kernel_ms = 0.f;
  if (n_steps == 0) return LBM_SUCCESS;
  if (c->steps_done + n_steps > c->capacity)
    LBM_FAIL(LBM_FAILURE, "lbm_run: %d steps requested but the av_vels record holds %d (maxIters)",
             c->steps_done + n_steps, c->capacity);
  if (c->halo != HALO_SELF && c->halo_mode != LBM_HALO_SYNC) return run_steps_stale(c, n_steps, kernel_ms);

  const float a1 = c->p.density * c->p.accel / 9.f;
  const float a2 = c->p.density * c->p.accel / 36.f;
  const bool halo = (c->halo != HALO_SELF);
  HotGuard hot_guard(c->team);

  // accelerate_flow() of the first step (later steps: epilogue of the step kernel)
  if (for_slabs(c, [&](int s) -> int {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        if (sl.accel_row >= 0 && sl.accel_row < sl.rows) {
          hipLaunchKernelGGL(lbm::accelerate_row, dim3(ceil_div(c->p.nx, 256)), dim3(256), 0, sl.compute,
                             sl.lat[c->cur], sl.mask, c->plane_stride, c->row_pitch, c->pitch, c->p.nx,
                             sl.accel_row, a1, a2);
          HIP_TRY(LBM_FAILURE, hipGetLastError());
        }
        if (kernel_ms) HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_t0, sl.compute));
        return LBM_SUCCESS;
      }) != LBM_SUCCESS)
    return LBM_FAILURE;

  if (c->resident && n_steps >= c->resident_min_steps) {
    if (run_resident(c, n_steps) != LBM_SUCCESS) return LBM_FAILURE;
    c->steps_done += n_steps;
    return kernel_ms ? read_step_timing(c, n_steps, kernel_ms) : LBM_SUCCESS;
  }

  int flushed_upto = c->steps_done;
  int t_first = 0;
  if (c->use_graph) {
    if (replay_chunks(c, n_steps, flushed_upto, &t_first) != LBM_SUCCESS) return LBM_FAILURE;
    flushed_upto += t_first;
  }
  if (halo && reset_pipeline(c) != LBM_SUCCESS) return LBM_FAILURE;

  // passes launch by launch: pass_steps timesteps per pass where enabled and that many remain, else two, else one
  static const int ext_events = env_int("LBM_EXT_EVENTS", 1);
  const int slots_per_pass = c->tile_steps > c->pass_steps ? c->tile_steps : c->pass_steps;
  int m = 0;  // pass counter (event parity)
  for (int t = t_first; t < n_steps; m++) {
    const int tile = (!halo && c->tile_steps) ? (c->tile_steps < n_steps - t ? c->tile_steps : n_steps - t) : 0;
    const int k = (!tile && c->fuse2 && n_steps - t >= 2) ? (n_steps - t >= c->pass_steps ? c->pass_steps : 2) : 0;
    const int adv = tile ? tile : (k ? k : 1);
    const bool last = (t + adv == n_steps);
    if (issue_pass(c, m, tile, k, !last, ext_events != 0) != LBM_SUCCESS) return LBM_FAILURE;
    t += adv;
    if (c->slot_fill + slots_per_pass > kPartSlots || last) {
      if (flush_partials(c, flushed_upto) != LBM_SUCCESS) return LBM_FAILURE;
      flushed_upto += c->slot_fill;
      c->slot_fill = 0;
    }
  }
  c->steps_done += n_steps;
  // (the last flush made every compute stream wait for its final boundary kernel)

  return kernel_ms ? read_step_timing(c, n_steps, kernel_ms) : LBM_SUCCESS;
}

// ---- freshest-available halo mode ------------------------------------------------------------------------------
int ensure_fresh_buffers(lbm_ctx* c) {
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    if (sl.fresh_stage) continue;
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.fresh_stage, 4 * (size_t)c->row_pitch * sizeof(float)));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.fresh_arrived, 4 * sizeof(unsigned)));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.fresh_id_src, 2 * sizeof(unsigned)));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.fresh_decision, sizeof(int)));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.fresh_log, (size_t)c->capacity));
    HIP_TRY(LBM_FAILURE, hipMemset(sl.fresh_arrived, 0, 4 * sizeof(unsigned)));
    HIP_TRY(LBM_FAILURE, hipMemset(sl.fresh_decision, 0, sizeof(int)));
    HIP_TRY(LBM_FAILURE, hipMemset(sl.fresh_log, 3, (size_t)c->capacity));  // synchronous and first passes: both sides fresh
    for (int i = 0; i < 2; i++) HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_fresh[i], hipEventDisableTiming));
  }
  return LBM_SUCCESS;
}

// F(m): the boundary rows of the lattice `src` (timestep id - 1 of the run, just produced) travel towards the staging
// rows [par] of the ring neighbours, each followed in stream order by `id`.  Same preconditions as the stale exchange
// (exchange_halos, slot >= 0): behind ev_step of this slab and, where this slab writes into its neighbours' memory
// itself, of the neighbours -- their previous pass, the last reader of staging [par] (two passes ago), is over.
// Nobody ever waits for it (LBM_FRESH_FORCE=wait excepted: tests).
int fresh_exchange(lbm_ctx* c, int src, int par, unsigned id) {
  const long n = c->row_pitch;
  // tests: hold about half of the (step, slab) exchanges back by so many microseconds, so that looks miss them
  const int delay_us = env_int("LBM_FRESH_TEST_DELAY_US", 0);
  auto delayed = [&](int s) { return delay_us > 0 && ((((id * 2654435761u) >> 11) ^ (unsigned)s) & 1u) != 0; };
  if (c->halo == HALO_RCCL) {
    for (int s = 0; s < c->n_slabs; s++) {
      Slab& sl = c->slab[s];
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sl.ev_step, 0));
      if (delayed(s)) hipLaunchKernelGGL(lbm::fresh_test_delay, dim3(1), dim3(1), 0, sl.comm, (long long)delay_us * 100);
    }
    RCCL_OR_FAIL(LBM_FAILURE);
    const RcclApi& nc = *rc_api_;
    if (!c->team) NCCL_TRY(LBM_FAILURE, nc.GroupStart());
    int rc = for_slabs(c, [&](int s) -> int {
      Slab& sl = c->slab[s];
      int me, parts;
      if (c->ranked) { me = c->rank; parts = c->world; } else { me = s; parts = c->n_slabs; }
      lbm_halo_op ops[4];
      if (lbm_halo_plan(sl.rows, parts, me, 1, ops) != LBM_SUCCESS) return LBM_FAILURE;
      if (c->team) {
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        NCCL_TRY(LBM_FAILURE, nc.GroupStart());
      }
      ncclResult_t res = ncclSuccess;
      for (int i = 0; i < 4 && res == ncclSuccess; i++) {
        // a receive of halo row -1 (rows) lands in the south (north) staging row instead
        float* ptr = ops[i].is_send ? sl.lat[src] + (long)ops[i].row_first * c->row_pitch
                                    : sl.fresh_stage + ((long)par * 2 + (ops[i].row_first < 0 ? 0 : 1)) * n;
        res = ops[i].is_send ? nc.Send(ptr, (size_t)n, ncclFloat, ops[i].peer, sl.nccl, sl.comm)
                             : nc.Recv(ptr, (size_t)n, ncclFloat, ops[i].peer, sl.nccl, sl.comm);
      }
      if (c->team) {
        const ncclResult_t end = nc.GroupEnd();
        if (res == ncclSuccess) res = end;
      }
      if (res != ncclSuccess) LBM_FAIL(LBM_FAILURE, "RCCL error in the halo exchange: %s", nc.GetErrorString(res));
      return LBM_SUCCESS;
    });
    if (!c->team) {
      const ncclResult_t end = nc.GroupEnd();
      if (rc == LBM_SUCCESS && end != ncclSuccess) { raise_error(__LINE__, "RCCL error: %s (ncclGroupEnd)", nc.GetErrorString(end)); rc = LBM_FAILURE; }
    }
    if (rc != LBM_SUCCESS) return rc;
    for (int s = 0; s < c->n_slabs; s++) {
      Slab& sl = c->slab[s];
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      hipLaunchKernelGGL(lbm::fresh_mark, dim3(1), dim3(1), 0, sl.comm, sl.fresh_arrived + par * 2, sl.fresh_arrived + par * 2 + 1, id);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_fresh[par], sl.comm));
    }
    return LBM_SUCCESS;
  }
  if (c->halo == HALO_MEMCPY) {
    return for_slabs(c, [&](int s) -> int {
      Slab& sl = c->slab[s];
      Slab& sn = c->slab[(s + 1) % c->n_slabs];
      Slab& ss = c->slab[(s - 1 + c->n_slabs) % c->n_slabs];
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sl.ev_step, 0));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, sn.ev_step, 0));
      HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.comm, ss.ev_step, 0));
      if (delayed(s)) hipLaunchKernelGGL(lbm::fresh_test_delay, dim3(1), dim3(1), 0, sl.comm, (long long)delay_us * 100);
      hipLaunchKernelGGL(lbm::fresh_mark, dim3(1), dim3(1), 0, sl.comm, sl.fresh_id_src + par, (unsigned*)nullptr, id);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      // my top row is my north neighbour's south halo row, my row 0 my south neighbour's north halo row
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(sn.fresh_stage + ((long)par * 2 + 0) * n, sl.lat[src] + (long)(sl.rows - 1) * c->row_pitch,
                                          (size_t)n * sizeof(float), hipMemcpyDefault, sl.comm));
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(sn.fresh_arrived + par * 2 + 0, sl.fresh_id_src + par, sizeof(unsigned), hipMemcpyDefault, sl.comm));
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(ss.fresh_stage + ((long)par * 2 + 1) * n, sl.lat[src], (size_t)n * sizeof(float), hipMemcpyDefault, sl.comm));
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(ss.fresh_arrived + par * 2 + 1, sl.fresh_id_src + par, sizeof(unsigned), hipMemcpyDefault, sl.comm));
      HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_fresh[par], sl.comm));
      return LBM_SUCCESS;
    });
  }
  LBM_FAIL(LBM_FAILURE, "the freshest-available halo mode is not available with the hosted exchange");
}

// Stale-halo ("asynchronous") timestep loop: the GPU analogue of the reference's research variant,
// MPI_Testall_OptimizedVersion/d2q9-bgk.c:256-301, which replaces MPI_Waitall by MPI_Testall and relaxes
// the boundary rows with whatever halo contents are there.  Here the staleness is pinned to exactly
// one pass, which keeps the run reproducible: pass m reads its own rows of lattice m but the halo
// rows its neighbours sent from lattice m-1 (pass 0 of every lbm_run call starts from fresh halos).
// Nothing on the compute stream ever waits for an exchange issued in the same pass:
//
//   compute stream:  S(0) ──────► S(1) ──────► S(2) ──────► S(3) ...    whole slab, one launch per pass
//                      │  ▲ X'(0)   │  ▲ X'(1)   │  ▲ X'(2)
//   comm stream:       └► X'(1) ────┴► X'(2) ────┴► X'(3) ...            X'(k): boundary rows of lattice k
//                                                                         -> halo rows S(k+1) reads
//
// X'(k) starts when S(k-1) has written lattice k and has a whole pass to land.  It writes the halo
// rows of the OTHER lattice buffer (the one S(k+1) reads), which S(k-1) finished reading and S(k)
// never touches, so there is no torn read -- unlike the reference, whose Irecv may land mid-row.
// Stale passes always advance ONE timestep, also where the synchronous pipeline uses the two-step kernel:
// then every population that crosses a slab boundary is simply delayed by one step -- nothing is lost or
// duplicated, steady states are unchanged, and the transient stays within 1 % (2 slabs) .. 4 % (8 slabs
// of 16 rows) of the synchronous run on the reference's 128x128 case.  With two steps per pass the
// redundantly relaxed halo-adjacent rows would be computed from stale data on one side of the seam and
// from fresh data on the other, which no longer conserves mass: measured, that variant drifts past the
// 1 % rule with 2 slabs and diverges to NaN after 2172 steps with 8 (profiles/r01_tuning.md).
//
// LBM_HALO_FRESHEST on top of that -- the reference's rule itself, "look once, never wait" (MPI_Testall_Optimized
// Version/d2q9-bgk.c:262-290: post the exchange, relax the interior rows, MPI_Testall, relax the boundary rows with
// whatever is there): the rows of lattice k ALSO travel (F(k), first on the comm stream) towards a staging row per
// side, followed by the step's id; S(k) is cut into interior rows, one look at the ids (fresh_decide: which sides have
// arrived, noted in the log), whole staging rows moved over the one-pass-old halo rows where they have (fresh_adopt),
// boundary rows.  Every halo row is the row of this pass or of the pass before -- never older, never torn -- and
// given the log of decisions the run is reproducible (tests/slab_model.py: run_slabs_freshest).
int run_steps_stale(lbm_ctx* c, int n_steps, float* kernel_ms) {
  const float a1 = c->p.density * c->p.accel / 9.f;
  const float a2 = c->p.density * c->p.accel / 36.f;
  const int depth = 1;  // one timestep per pass (see above): only the adjacent row is read
  const bool freshest = (c->halo_mode == LBM_HALO_FRESHEST);
  // tests: "wait" makes every look find its rows (= the synchronous run), "never" sends none (= the stale mode)
  const char* force_env = freshest ? getenv("LBM_FRESH_FORCE") : nullptr;
  const bool force_wait = force_env && !strcmp(force_env, "wait"), force_never = force_env && !strcmp(force_env, "never");
  if (freshest && c->halo == HALO_HOST) LBM_FAIL(LBM_FAILURE, "the freshest-available halo mode is not available with the hosted exchange");
  if (freshest && ensure_fresh_buffers(c) != LBM_SUCCESS) return LBM_FAILURE;
  HotGuard hot_guard(c->team);

  // accelerate_flow() of the first step, then fresh halos for pass 0 (same lattice) and, from the same
  // rows, the one-pass-old halos of pass 1 (other lattice)
  if (for_slabs(c, [&](int s) -> int {
        Slab& sl = c->slab[s];
        HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
        if (sl.accel_row >= 0 && sl.accel_row < sl.rows) {
          hipLaunchKernelGGL(lbm::accelerate_row, dim3(ceil_div(c->p.nx, 256)), dim3(256), 0, sl.compute,
                             sl.lat[c->cur], sl.mask, c->plane_stride, c->row_pitch, c->pitch, c->p.nx,
                             sl.accel_row, a1, a2);
          HIP_TRY(LBM_FAILURE, hipGetLastError());
        }
        HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_step, sl.compute));
        if (kernel_ms) HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_t0, sl.compute));
        return LBM_SUCCESS;
      }) != LBM_SUCCESS)
    return LBM_FAILURE;
  if (exchange_halos(c, depth, c->cur, c->cur, 1) != LBM_SUCCESS) return LBM_FAILURE;      // read by S(0)
  if (exchange_halos(c, depth, c->cur, c->cur ^ 1, 0) != LBM_SUCCESS) return LBM_FAILURE;  // read by S(1)

  int flushed_upto = c->steps_done;
  for (int m = 0; m < n_steps; m++) {  // pass m = timestep m of this call
    const bool last = (m + 1 == n_steps);
    const int slot = (m + 1) & 1;  // the exchange S(m) consumes: issued during pass m-2 (or the prologue)
    if (for_slabs(c, [&](int s) -> int {
          Slab& sl = c->slab[s];
          HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
          // own exchange: my halo rows have landed (RCCL) and the boundary rows of the lattice this pass
          // overwrites have been read out (the copies X'(m-1) made from it)
          HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, sl.ev_x[slot], 0));
          if (c->halo == HALO_MEMCPY) {
            // push model: my halo rows are written by the neighbours' streams
            const int north = (s + 1) % c->n_slabs, south = (s - 1 + c->n_slabs) % c->n_slabs;
            HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, c->slab[north].ev_x[slot], 0));
            HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, c->slab[south].ev_x[slot], 0));
          }
          if (!freshest) {
            if (launch_step(c, s, sl.compute, 0, 1, sl.rows, 0, !last) != LBM_SUCCESS) return LBM_FAILURE;
          } else {
            const int par = m & 1;
            const unsigned id = (unsigned)(c->steps_done + m) + 1u;
            if (launch_step(c, s, sl.compute, 1, 1, sl.rows - 2, 0, !last) != LBM_SUCCESS) return LBM_FAILURE;
            if (m > 0 && force_wait) {
              if (c->halo == HALO_MEMCPY) {
                const int north = (s + 1) % c->n_slabs, south = (s - 1 + c->n_slabs) % c->n_slabs;
                HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, c->slab[north].ev_fresh[par], 0));
                HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, c->slab[south].ev_fresh[par], 0));
              } else {
                HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, sl.ev_fresh[par], 0));
              }
            }
            hipLaunchKernelGGL(lbm::fresh_decide, dim3(1), dim3(1), 0, sl.compute, (const unsigned*)(sl.fresh_arrived + par * 2), id,
                               m == 0 ? 1 : 0, sl.fresh_decision, sl.fresh_log + c->steps_done + m);
            HIP_TRY(LBM_FAILURE, hipGetLastError());
            if (m > 0) {
              const long n = c->row_pitch;
              float* lat = sl.lat[c->cur];
              hipLaunchKernelGGL(lbm::fresh_adopt, dim3(ceil_div(n, 256)), dim3(256), 0, sl.compute, (const int*)sl.fresh_decision,
                                 (const unsigned*)(sl.fresh_stage + ((long)par * 2 + 0) * n), (const unsigned*)(sl.fresh_stage + ((long)par * 2 + 1) * n),
                                 (unsigned*)(lat - n), (unsigned*)(lat + (long)sl.rows * c->row_pitch), n);
              HIP_TRY(LBM_FAILURE, hipGetLastError());
            }
            if (launch_step(c, s, sl.compute, 0, sl.rows - 1, 2, sl.blocks_main, !last) != LBM_SUCCESS) return LBM_FAILURE;
          }
          HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_step, sl.compute));
          return LBM_SUCCESS;
        }) != LBM_SUCCESS)
      return LBM_FAILURE;
    for (int s = 0; s < c->n_slabs; s++)
      c->slab[s].slot_counts.n[c->slot_fill] = freshest ? c->slab[s].blocks_main + c->slab[s].blocks_boundary : blocks_for_rows(c, c->slab[s].rows);
    c->cur ^= 1;
    c->slot_fill += 1;
    // F(m+1): the rows S(m) just produced, for S(m+1) itself if they get there before its look
    if (!last && freshest && !force_never && fresh_exchange(c, c->cur, (m + 1) & 1, (unsigned)(c->steps_done + m + 1) + 1u) != LBM_SUCCESS)
      return LBM_FAILURE;
    // X'(m+1): the same rows, for S(m+2)
    if (!last && exchange_halos(c, depth, c->cur, c->cur ^ 1, slot) != LBM_SUCCESS) return LBM_FAILURE;
    if (c->slot_fill >= kPartSlots - 1 || last) {
      if (flush_partials(c, flushed_upto) != LBM_SUCCESS) return LBM_FAILURE;
      flushed_upto += c->slot_fill;
      c->slot_fill = 0;
    }
  }
  c->steps_done += n_steps;

  // leave the comm streams ordered before whatever the host enqueues on the compute streams next
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    for (int i = 0; i < 2; i++) HIP_TRY(LBM_FAILURE, hipStreamWaitEvent(sl.compute, sl.ev_x[i], 0));
  }
  return kernel_ms ? read_step_timing(c, n_steps, kernel_ms) : LBM_SUCCESS;
}

void free_slab(Slab& sl) {
  if (hipSetDevice(sl.device) != hipSuccess) return;
  // graphs that captured RCCL operations hold on to the communicator: they go first
  for (int i = 0; i < 2; i++)
    if (sl.chunk_graph[i]) { (void)hipGraphExecDestroy(sl.chunk_graph[i]); sl.chunk_graph[i] = nullptr; }
  if (sl.nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(sl.nccl);
  for (int i = 0; i < 2; i++) if (sl.lat_alloc[i]) (void)hipFree(sl.lat_alloc[i]);
  if (sl.mask_alloc) (void)hipFree(sl.mask_alloc);
  if (sl.partials) (void)hipFree(sl.partials);
  if (sl.tot_u) (void)hipFree(sl.tot_u);
  if (sl.scratch) (void)hipFree(sl.scratch);
  if (sl.reduce_buf) (void)hipFree(sl.reduce_buf);
  if (sl.flushed_dev) (void)hipFree(sl.flushed_dev);
  if (sl.fresh_stage) (void)hipFree(sl.fresh_stage);
  if (sl.fresh_arrived) (void)hipFree(sl.fresh_arrived);
  if (sl.fresh_id_src) (void)hipFree(sl.fresh_id_src);
  if (sl.fresh_decision) (void)hipFree(sl.fresh_decision);
  if (sl.fresh_log) (void)hipFree(sl.fresh_log);
  for (int i = 0; i < 2; i++) if (sl.ev_fresh[i]) (void)hipEventDestroy(sl.ev_fresh[i]);
  if (sl.res_gran) (void)hipFree(sl.res_gran);
  if (sl.res_part) (void)hipFree(sl.res_part);
  if (sl.res_status) (void)hipFree(sl.res_status);
  if (sl.res_status_host) (void)hipHostFree(sl.res_status_host);
  if (sl.ev_boundary) (void)hipEventDestroy(sl.ev_boundary);
  if (sl.ev_halo) (void)hipEventDestroy(sl.ev_halo);
  for (int i = 0; i < 2; i++) if (sl.ev_interior[i]) (void)hipEventDestroy(sl.ev_interior[i]);
  if (sl.ev_flush) (void)hipEventDestroy(sl.ev_flush);
  if (sl.ev_step) (void)hipEventDestroy(sl.ev_step);
  if (sl.ev_fork) (void)hipEventDestroy(sl.ev_fork);
  for (int i = 0; i < 2; i++) if (sl.ev_x[i]) (void)hipEventDestroy(sl.ev_x[i]);
  if (sl.ev_t0) (void)hipEventDestroy(sl.ev_t0);
  if (sl.ev_t1) (void)hipEventDestroy(sl.ev_t1);
  if (sl.compute) (void)hipStreamDestroy(sl.compute);
  if (sl.comm) (void)hipStreamDestroy(sl.comm);
  sl = Slab();
}

bool validate_params(const lbm_params* p) {
  // the cell count must fit the reference's int counters (tot_cells, SerialCode/d2q9-bgk.c:411)
  return p && p->nx >= 1 && p->ny >= 2 && p->max_iters >= 0 && (long)p->nx * (long)p->ny <= 2147483647L;
}

// a device staging buffer that is freed on every way out of the scope that allocated it
struct DeviceTemp {
  void* p = nullptr;
  ~DeviceTemp() { if (p) (void)hipFree(p); }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Build one slab: allocate, build the mask on the device, fill the lattice.
int build_slab(lbm_ctx* c, int s, const ObstacleSource& obst, const float* cells_aos) {
  Slab& sl = c->slab[s];
  const lbm_params& p = c->p;
  HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
  HIP_TRY(LBM_FAILURE, hipStreamCreateWithFlags(&sl.compute, hipStreamNonBlocking));
  HIP_TRY(LBM_FAILURE, hipStreamCreateWithFlags(&sl.comm, hipStreamNonBlocking));
  HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_boundary, hipEventDisableTiming));
  HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_halo, hipEventDisableTiming));
  for (int i = 0; i < 2; i++) HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_interior[i], hipEventDisableTiming));
  HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_flush, hipEventDisableTiming));
  HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_step, hipEventDisableTiming));
  HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_fork, hipEventDisableTiming));
  for (int i = 0; i < 2; i++) HIP_TRY(LBM_FAILURE, hipEventCreateWithFlags(&sl.ev_x[i], hipEventDisableTiming));
  HIP_TRY(LBM_FAILURE, hipEventRecord(sl.ev_halo, sl.comm));
  HIP_TRY(LBM_FAILURE, hipEventCreate(&sl.ev_t0));
  HIP_TRY(LBM_FAILURE, hipEventCreate(&sl.ev_t1));

  // lattices with kHaloRows halo rows below and above the owned rows (zeroed: pitch padding and
  // unused halo rows stay finite); lat[] points at owned row 0
  const size_t lat_bytes = (size_t)(sl.rows + 2 * kHaloRows) * c->row_pitch * sizeof(float);
  for (int i = 0; i < 2; i++) {
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.lat_alloc[i], lat_bytes));
    HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.lat_alloc[i], 0, lat_bytes, sl.compute));
    sl.lat[i] = sl.lat_alloc[i] + (size_t)kHaloRows * c->row_pitch;
  }
  HIP_TRY(LBM_FAILURE, hipMalloc(&sl.partials, (size_t)kPartSlots * c->part_stride * sizeof(float)));
  HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.partials, 0, (size_t)kPartSlots * c->part_stride * sizeof(float), sl.compute));
  HIP_TRY(LBM_FAILURE, hipMalloc(&sl.tot_u, (size_t)(c->capacity > 0 ? c->capacity : 1) * sizeof(double)));
  HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.tot_u, 0, (size_t)(c->capacity > 0 ? c->capacity : 1) * sizeof(double), sl.compute));
  HIP_TRY(LBM_FAILURE, hipMalloc(&sl.scratch, 2 * kSumBlocks * sizeof(double)));
  HIP_TRY(LBM_FAILURE, hipMalloc(&sl.flushed_dev, sizeof(int)));
  if (c->resident) {
    // granules start at tag 0 = "nothing"; tags are global step indices + 1, so they never need clearing again
    const size_t gran_bytes = resident_gran_bytes(c);
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.res_gran, gran_bytes));
    HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.res_gran, 0, gran_bytes, sl.compute));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.res_part, (size_t)kResidentChunk * c->resident_bands * sizeof(float)));
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.res_status, sizeof(int)));
    HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.res_status, 0, sizeof(int), sl.compute));
    HIP_TRY(LBM_FAILURE, hipHostMalloc(&sl.res_status_host, sizeof(int)));
    *sl.res_status_host = 0;
  }
  if (c->ranked) HIP_TRY(LBM_FAILURE, hipMalloc(&sl.reduce_buf, (size_t)(c->capacity > 0 ? c->capacity : 1) * sizeof(double)));

  // obstacle mask: uint8 (rows + 2*kMaskHalo) x pitch with the (periodic) neighbour rows beyond the slab, which a
  // multi-step pass relaxes redundantly.  Built on the device: from the reference's host type (int, SerialCode/
  // d2q9-bgk.c:541) uploaded row range by row range through a bounded staging buffer, or expanded from a small tile.
  {
    const int mrows = sl.rows + 2 * kMaskHalo;
    const long mask_cells = (long)mrows * c->pitch;
    HIP_TRY(LBM_FAILURE, hipMalloc(&sl.mask_alloc, (size_t)mask_cells));
    HIP_TRY(LBM_FAILURE, hipMemsetAsync(sl.mask_alloc, 0, (size_t)mask_cells, sl.compute));
    sl.mask = sl.mask_alloc + (size_t)kMaskHalo * c->pitch;
    if (obst.kind == OBST_TILE) {
      const size_t tn = (size_t)obst.tile_nx * obst.tile_ny;
      std::vector<unsigned char> t8(tn);
      for (size_t i = 0; i < tn; i++) t8[i] = obst.data[i] ? 1 : 0;
      DeviceTemp tile_buf;
      HIP_TRY(LBM_FAILURE, hipMalloc(&tile_buf.p, tn));
      unsigned char* tile_dev = tile_buf.as<unsigned char>();
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(tile_dev, t8.data(), tn, hipMemcpyHostToDevice, sl.compute));
      hipLaunchKernelGGL(lbm::mask_from_tile, dim3(ceil_div((long)p.nx * mrows, 256)), dim3(256), 0, sl.compute, tile_dev,
                         obst.tile_nx, obst.tile_ny, sl.mask_alloc, p.nx, c->pitch, sl.row_first - kMaskHalo, mrows, p.ny);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    } else {
      long chunk_rows = (32L << 20) / ((long)p.nx * sizeof(int));
      if (chunk_rows < 1) chunk_rows = 1;
      DeviceTemp stage_buf;
      HIP_TRY(LBM_FAILURE, hipMalloc(&stage_buf.p, (size_t)chunk_rows * p.nx * sizeof(int)));
      int* stage = stage_buf.as<int>();
      for (int r = 0; r < mrows;) {
        // source row of mask row r, and how many rows from there are contiguous in the source
        long src_row;
        int run = mrows - r;
        if (obst.kind == OBST_ROWS) {
          src_row = (long)(sl.row_first - c->row_first) + r;  // the caller's array starts kMaskHalo rows below its first row
        } else {
          const int g = ((sl.row_first - kMaskHalo + r) % p.ny + p.ny) % p.ny;
          src_row = g;
          if (run > p.ny - g) run = p.ny - g;
        }
        if (run > chunk_rows) run = (int)chunk_rows;
        HIP_TRY(LBM_FAILURE, hipMemcpyAsync(stage, obst.data + (size_t)src_row * p.nx, (size_t)run * p.nx * sizeof(int),
                                            hipMemcpyHostToDevice, sl.compute));
        hipLaunchKernelGGL(lbm::mask_from_int, dim3(ceil_div((long)p.nx * run, 256)), dim3(256), 0, sl.compute, stage,
                           sl.mask_alloc + (size_t)r * c->pitch, p.nx, c->pitch, run);
        HIP_TRY(LBM_FAILURE, hipGetLastError());
        HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));  // the staging buffer is reused
        r += run;
      }
    }
    // fluid cells of the owned rows (the reference counts them while parsing, MPI_Waitall/d2q9-bgk.c:794-804)
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(sl.scratch);
    HIP_TRY(LBM_FAILURE, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), sl.compute));
    hipLaunchKernelGGL(lbm::count_blocked, dim3(ceil_div((long)c->pitch * sl.rows, 256 * 16)), dim3(256), 0, sl.compute,
                       sl.mask, (long)c->pitch * sl.rows, cnt);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
    unsigned long long blocked = 0;
    HIP_TRY(LBM_FAILURE, hipMemcpyAsync(&blocked, cnt, sizeof(blocked), hipMemcpyDeviceToHost, sl.compute));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    sl.fluid_cells = (long)p.nx * sl.rows - (long)blocked;
  }

  // lattice: equilibrium or the caller's cells
  if (!cells_aos) {
    const float r0 = p.density * 4.f / 9.f;  // SerialCode/d2q9-bgk.c:546-548
    const float r1 = p.density / 9.f;
    const float r2 = p.density / 36.f;
    hipLaunchKernelGGL(lbm::init_equilibrium, dim3(ceil_div((long)p.nx * sl.rows, 256)), dim3(256), 0,
                       sl.compute, sl.lat[0], c->plane_stride, c->row_pitch, p.nx, sl.rows, r0, r1, r2);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
  } else {
    // upload in chunks of rows through a staging buffer, transposing AoS -> SoA on the device
    const int chunk_rows = (int)(((64L << 20) / ((long)p.nx * lbm::kQ * sizeof(float))) > 0
                                     ? ((64L << 20) / ((long)p.nx * lbm::kQ * sizeof(float)))
                                     : 1);
    DeviceTemp stage_buf;
    HIP_TRY(LBM_FAILURE, hipMalloc(&stage_buf.p, (size_t)chunk_rows * p.nx * lbm::kQ * sizeof(float)));
    float* stage = stage_buf.as<float>();
    for (int r0 = 0; r0 < sl.rows; r0 += chunk_rows) {
      const int nr = (sl.rows - r0 < chunk_rows) ? sl.rows - r0 : chunk_rows;
      const size_t n = (size_t)nr * p.nx * lbm::kQ;
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(stage, cells_aos + (size_t)(sl.row_first - (obst.local_cells ? c->row_first : 0) + r0) * p.nx * lbm::kQ,
                                          n * sizeof(float), hipMemcpyHostToDevice, sl.compute));
      hipLaunchKernelGGL(lbm::aos_to_soa, dim3(ceil_div((long)n, 256)), dim3(256), 0, sl.compute, stage,
                         sl.lat[0], c->plane_stride, c->row_pitch, p.nx, r0, nr);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    }
  }
  HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
  return LBM_SUCCESS;
}

// Which step kernel advances a decomposition of the grid into `parts` row slabs (in one process or over ranks), and how
// many timesteps it takes per pass -- from global numbers only, so every rank of a multi-process run decides alike.
// create_common and the host-only query lbm_plan_halo_depth share it.
struct StreamPlan { bool vec4; int fuse2, lane_cells, pass_steps; };
StreamPlan plan_stream(const lbm_params* params, int parts, bool halo_on, int math_mode) {
  StreamPlan pl;
  // 4 cells per lane need nx % 4 == 0; tiny single-slab grids are latency-bound and run faster with one
  // cell per lane (4x the waves, a quarter of the dependent arithmetic per lane: 128^2 3.2 vs 5.0 us per
  // step, 256^2 3.8 vs 5.2; from 512^2 on the 4-cell kernel wins).  Asking for the stream kernel,
  // which exists in the 4- and 2-cell forms only, implies vec4; so do halos.
  pl.vec4 = (params->nx % 4 == 0) &&
            env_int("LBM_VEC4", ((long)params->nx * params->ny >= 128L * 1024 || env_int("LBM_FUSE2", 0) == 1 || halo_on) ? 1 : 0);
  const int min_rows = params->ny / (parts > 0 ? parts : 1);  // the thinnest slab of a balanced partition
  const long min_cells = (long)params->nx * min_rows;
  // Across slabs / ranks a pass costs one exchange and ~10 runtime calls per slab whatever it computes, so
  // several timesteps per pass always pay there (1024^2 over 2/4/8 slabs on one device: 45/74/97 us per step
  // vs 70/113/125 one-step; 2048^2 over 8: 96 vs 261).
  pl.fuse2 = (pl.vec4 && env_int("LBM_FUSE2", (min_cells >= 300L * 1024 || halo_on) ? 1 : 0)) ? 1 : 0;
  pl.lane_cells = env_int("LBM_LANE_CELLS", min_cells >= 7L * 512 * 1024 ? 4 : 2) == 2 ? 2 : 4;  // from 3.5 Mi cells
  // Timesteps per pass of the stream kernel.  The two-step kernel at 8192^2 is bound by DRAM traffic (round-2 PMC:
  // 5.4-5.8 TB/s at the memory controllers whatever the band height or the arithmetic), so the 4-cell form runs more
  // steps per pass: K = 3 (stepk_stream, 2 waves per SIMD, next row prefetched) 0.345 vs 0.47-0.49 ms per step, at
  // which point it is bound by VALU issue again (K = 4 with scalar arithmetic: 0.36); with the collision on PAIRS of
  // cells (stepk_pk: v_pk_* instructions, 108 instead of 155 lane-instructions per update) K = 4 pays: 0.275-0.285.
  // The packed kernel exists for the exact arithmetic only; the 2-cell form for K = 2 only.
  // The two-cell form (one pair per lane, twice the waves: mid-size grids) takes two halo lanes per side beyond two
  // steps and runs K = 3 as the packed kernel (124 VGPRs, 4 waves per SIMD): 1024^2 9.4 vs 10.7 us (K = 2), 1280^2
  // 12.4 vs 15.2 (four-cell K = 4), 1536^2 15.2 vs 20.9, 1792^2 20.2 vs 22.3; from 2048^2 the four-cell form wins
  // (24.9 vs 25.6-27.5).
  // FAST math (reciprocal + FMA, scalar) is the faster arithmetic only in the one-step and LDS-tile kernels.  The
  // multi-step stream kernels run the packed EXACT collision in both modes: it is faster than the scalar fast form
  // (8192^2: 0.27 vs 0.345 ms per step) and at K = 4 it already sits at the DRAM bound of its access pattern (5.8 GB per
  // launch at 5.4-5.8 TB/s), so a packed fast form could not be faster -- and exact results meet the fast mode's
  // tolerance trivially.  LBM_PACKED=0 selects the scalar kernels (fast math: K = 3 / 2).
  (void)math_mode;
  const bool exact_packed = env_int("LBM_PACKED", 1) != 0;
  pl.pass_steps = env_int("LBM_PASS_STEPS", pl.lane_cells == 4 ? (exact_packed ? 4 : 3) : (exact_packed ? 3 : 2));
  if (pl.pass_steps < 2 || pl.pass_steps > kHaloRows) pl.pass_steps = 2;
  if (pl.lane_cells != 4 && !exact_packed) pl.pass_steps = 2;  // the scalar two-cell kernel (step2_stream) is two-step
  // across slabs a K-step pass needs slabs of at least 2K rows (the stream kernel at all: 4); a periodic slab at least K
  if (halo_on && min_rows < 2 * pl.pass_steps) pl.pass_steps = 2;
  if (halo_on && min_rows < 4) pl.fuse2 = 0;
  if (min_rows < pl.pass_steps) pl.pass_steps = 2;
  return pl;
}

lbm_ctx* create_common(const lbm_params* params, const ObstacleSource& obst, const float* cells_aos,
                       int n_slabs, int math_mode, int rank, int world, const void* unique_id,
                       int device, const lbm_host_comm* host_comm = nullptr) {
  if (!validate_params(params)) LBM_FAIL(nullptr, "lbm_create: invalid parameters");
  if (!obst.data) LBM_FAIL(nullptr, "lbm_create: obstacles is NULL");
  if (obst.kind == OBST_TILE && (obst.tile_nx < 1 || obst.tile_ny < 1))
    LBM_FAIL(nullptr, "lbm_create: invalid obstacle tile %dx%d", obst.tile_nx, obst.tile_ny);
  if (math_mode != LBM_MATH_EXACT && math_mode != LBM_MATH_FAST)
    LBM_FAIL(nullptr, "lbm_create: unknown math mode %d", math_mode);
  if (n_slabs < 1 || n_slabs > kMaxSlabs) LBM_FAIL(nullptr, "lbm_create: n_gpus must be 1..%d", kMaxSlabs);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    LBM_FAIL(nullptr, "lbm_create: no HIP device available (this library has no CPU path)");

  lbm_ctx* c = new lbm_ctx();
  c->p = *params;
  c->math_mode = math_mode;
  c->rank = rank;
  c->world = world;
  c->capacity = params->max_iters;
  c->pitch = (int)round_up(params->nx, 64);
  c->plane_stride = c->pitch + env_int("LBM_PLANE_PAD_FLOATS", 0) / 4 * 4;
  c->row_pitch = 9 * c->plane_stride;
  c->neigh = env_int("LBM_NEIGH", 0);
  if (c->neigh < 0 || c->neigh > 2) c->neigh = 0;
  // nontemporal stores pay once the two lattices no longer fit the 256 MiB Infinity Cache
  // (measured: +4 % at 4096^2 and above, -2..-20 % at 2048^2 and below; profiles/r01_tuning.md)
  const double lattice_pair_bytes = 2.0 * 36.0 * (double)params->nx * (double)params->ny;
  c->nts = env_int("LBM_NTS", lattice_pair_bytes > 512.0 * 1024 * 1024 ? 1 : 0) ? 1 : 0;
  c->snake = env_int("LBM_SNAKE", 0) ? 1 : 0;
  // hipGraph replay pays where the loop is bound by the host's launch rate (~3.5 us per launch): measured
  // 128^2 3.11 vs 3.52 us per step, 128x256 3.22 vs 3.53; no difference from 256^2 on
  c->use_graph = env_int("LBM_GRAPH", (long)params->nx * params->ny < 64L * 1024 ? 1 : 0) ? 1 : 0;
  c->n_slabs = n_slabs;

  // rows of this context, then of each slab
  if (lbm_partition_rows(params->ny, world, rank, &c->row_first, &c->row_count) != LBM_SUCCESS) {
    delete c;
    return nullptr;
  }
  const bool force_halo = env_int("LBM_FORCE_HALO", 0) != 0;
  c->ranked = (unique_id != nullptr) || (host_comm != nullptr);
  c->hosted = (host_comm != nullptr);
  if (c->hosted) c->host_comm = *host_comm;
  if (world > 1 || (c->ranked && force_halo)) c->halo = c->hosted ? HALO_HOST : HALO_RCCL;
  else if (n_slabs > 1 || force_halo) {
    const char* h = getenv("LBM_HALO");
    bool distinct = (n_slabs <= ndev);
    c->halo = (h && !strcmp(h, "memcpy")) ? HALO_MEMCPY
              : (h && !strcmp(h, "rccl")) ? HALO_RCCL
              : (distinct ? HALO_RCCL : HALO_MEMCPY);
  } else c->halo = HALO_SELF;

  {
    const char* hm = getenv("LBM_HALO_MODE");
    if (hm && !strcmp(hm, "stale")) {
      c->halo_mode = LBM_HALO_STALE;
      static bool warned = false;
      if (!warned && c->halo != HALO_SELF && rank == 0) {
        warned = true;
        fprintf(stderr, "lbm_hip: LBM_HALO_MODE=stale is EXPERIMENTAL: halo rows one pass old; results differ from the "
                        "synchronous run (measured up to 4.7 %% on av_vels mid-transient, outside check.py's 1 %% rule)\n");
      }
    } else if (hm && !strcmp(hm, "freshest") && !c->hosted) {
      c->halo_mode = LBM_HALO_FRESHEST;
      static bool warned = false;
      if (!warned && c->halo != HALO_SELF && rank == 0) {
        warned = true;
        fprintf(stderr, "lbm_hip: LBM_HALO_MODE=freshest is EXPERIMENTAL: every halo row is this step's or the step before's, "
                        "whichever has arrived; results differ from the synchronous run and from run to run\n");
      }
    }
  }

  const bool halo_on = (c->halo != HALO_SELF);
  const StreamPlan plan = plan_stream(params, world * n_slabs, halo_on, math_mode);
  c->vec4 = plan.vec4;

  int max_blocks = 0;
  for (int s = 0; s < n_slabs; s++) {
    Slab& sl = c->slab[s];
    int first = 0, count = c->row_count;
    if (n_slabs > 1 && lbm_partition_rows(c->row_count, n_slabs, s, &first, &count) != LBM_SUCCESS) {
      delete c;
      return nullptr;
    }
    sl.device = c->ranked ? device : (s % ndev);
    sl.row_first = c->row_first + first;
    sl.rows = count;
    const int lid = params->ny - 2;  // SerialCode/d2q9-bgk.c:223
    // slab-local index of the lid row; with several slabs it may be one of MY halo rows (-kMaskHalo..-1 or
    // rows..rows+kMaskHalo-1), which a multi-step pass relaxes redundantly and must accelerate like its owner does
    sl.accel_row = sl.accel_row2 = lbm::kNoRow;
    for (int shift = -1; shift <= 1; shift++) {
      const int local = lid + shift * params->ny - sl.row_first;
      const bool owned = (local >= 0 && local < sl.rows);
      const bool in_halo = (c->halo != HALO_SELF) && ((local < 0 && local >= -kMaskHalo) || (local >= sl.rows && local < sl.rows + kMaskHalo));
      if (owned || in_halo) {
        if (sl.accel_row == lbm::kNoRow || owned) { if (sl.accel_row != lbm::kNoRow) sl.accel_row2 = sl.accel_row; sl.accel_row = local; }
        else sl.accel_row2 = local;
      }
    }
    if (c->halo == HALO_SELF) {
      sl.blocks_main = blocks_for_rows(c, sl.rows);
      sl.blocks_boundary = 0;
    } else {
      if (sl.rows < 2) {
        raise_error(__LINE__, "lbm_create: a slab needs at least 2 rows");
        delete c;
        return nullptr;
      }
      sl.blocks_main = blocks_for_rows(c, sl.rows - 2);
      sl.blocks_boundary = blocks_for_rows(c, 2);
    }
    if (sl.blocks_main + sl.blocks_boundary > max_blocks) max_blocks = sl.blocks_main + sl.blocks_boundary;
  }
  // ---- which kernel, and its geometry (all measured on MI355X; profiles/r01_tuning.md) --------
  //   single periodic slab below 300 Ki cells (round 2: the packed two-cell stream kernel wins from 576^2 on: 6.0 vs 7.2 us,
  //                        640^2 7.2 vs 8.9, 704^2 7.2 vs 9.1; 512^2 5.8 vs 5.4): LDS tiles, 4 or 3 timesteps per launch (step_tile; set further
  //                        down).  With halos: always several timesteps per pass (fewer exchanges).
  //   (one timestep per pass, step_vec4 / step_scalar: the odd last step of a run, widths that are not a multiple
  //                        of 4, LBM_FUSE2=0; it was the default up to 1.5 Mi cells until the two-step kernel stopped
  //                        computing |u| on its warm-up rows: 768^2 9.8 vs 11.3 us, 1024^2 12.35 vs 13.23, 1152^2 15.3 vs 18.1)
  //   0.3 .. 3.5 Mi cells : THREE timesteps per pass, 2 cells per lane (one pair, two halo lanes per side: 124 VGPRs,
  //                        4 waves/SIMD, twice the waves of the 4-cell form; 1024^2 9.4 us vs 10.7 two-step)
  //   >= 3.5 Mi cells    : FOUR timesteps per pass on pairs of cells, 4 cells per lane (16-byte accesses; us per step,
  //                        this form | 2-cell two-step: 1024^2 15.1 | 10.7, 1280^2 15.2 | 17.2, 1536^2 20.9 | 22.1,
  //                        1792^2 22.1 | 28.2; three-step scalar | two-step: 2048^2 31.8 | 35.4, 3072^2 59.0 | 76.0,
  //                        4096^2 93 | 129, 8192^2 340 | 492; four-step packed: 2048^2 24.9, 4096^2 75.3, 8192^2 277-285)
  // LBM_FUSE2, LBM_LANE_CELLS, LBM_BAND_ROWS override.  Ranks decide from global numbers only, so
  // every rank of a multi-process run takes the same path.
  c->fuse2 = plan.fuse2;
  c->lane_cells = plan.lane_cells;
  c->pass_steps = plan.pass_steps;
  c->halo_lanes = ceil_div(c->pass_steps, c->lane_cells);
  c->n_strips = ceil_div(params->nx / c->lane_cells > 0 ? params->nx / c->lane_cells : 1, 64 - 2 * c->halo_lanes);
  // Packed arithmetic (exact mode, 4 cells per lane): on.  With K = 4 two of the three sliding windows live in LDS
  // (18 KB per wave), which leaves registers to prefetch the next row (216 VGPRs): us per step, this form | packed
  // without prefetch / LDS | scalar K = 3: 16384^2 1091 | 1097 | 1355, 12288^2 640 | 652 | 838, 6144^2 180 | 187 | 233,
  // 4096^2 75.3 | 78.1 | 94.7, 3072^2 45.6 | 45.3 | 59.0, 2048^2 24.9 | 26.3 | 31.9; a rank's share through the halo
  // pipeline 8192x1024 45.1 | 46.7 | 55.3, 8192x2048 77.5 | 81.4 | 98.9, 8192x4096 149 | 152 | 187.
  c->packed = env_int("LBM_PACKED", 1) ? 1 : 0;  // both math modes (see plan_stream)
  c->lds_windows = env_int("LBM_LDS_WINDOWS", (c->packed && c->pass_steps == 4) ? 2 : 0);
  if (c->lds_windows < 0 || c->lds_windows > 2 || !c->packed) c->lds_windows = 0;
  // scalar K = 4 with prefetch spills (245 + 36 VGPRs); the packed K = 4 needs its LDS windows for it
  c->prefetch = env_int("LBM_PREFETCH", (c->lane_cells == 4 && (c->pass_steps == 3 || (c->pass_steps == 4 && c->lds_windows == 2))) ? 1 : 0) ? 1 : 0;
  // strips per XCD chunk: a whole band of strips, for slabs of many rounds of waves only (16384^2, K = 4: 1.033 ms per
  // step with 67-strip chunks, 1.088 with 34, 1.107 without; K = 3: 12288^2 0.793 vs 0.832).  Elsewhere the band height
  // packs the waves tightly into rounds (below) and the few empty workgroups of the chunked order spill into an
  // extra round (4096^2: 0.135 vs 0.093; 8192^2: 0.298 vs 0.288).
  const bool many_rounds = (long)c->n_strips * ceil_div(c->row_count / n_slabs, 24) >= 16L * 1024;
  c->xcd_chunk = env_int("LBM_XCD_CHUNK", (c->pass_steps >= 3 && many_rounds) ? c->n_strips : 0);
  if (c->xcd_chunk < 0 || c->xcd_chunk > c->n_strips) c->xcd_chunk = 0;
  c->use_stepk = env_int("LBM_STEPK", 0) ? 1 : 0;

  // Band height.  A wave sweeps band_rows + 2 rows.
  //   4-cell form (256 CUs x 12 waves resident): short bands, by row width -- measured optimum 7 rows at 8192 cells
  //   per row (8192^2: 0.477-0.480 ms vs 0.481-0.484 at 6, 0.495 at 4; same in the halo pipeline), 4-5 rows for
  //   narrower and for wider rows (7168^2: 0.384 at 4 vs 0.411 at 7; 6144^2: 0.277 at 5 vs 0.303 at 7; 4096^2
  //   0.131-0.132 at 4-7; 12288: 5; 16384^2: 2.01 at 4 vs 2.23 at 6).  Fitting whole rounds of resident waves
  //   does NOT pay here (4096^2: 0.146 with the round model's 23-row bands vs 0.131; 8192x2048: 0.139 vs 0.126).
  //   2-cell form (mid-size grids, 256 x 20 waves resident): a slab that fits in a few rounds is quantised by
  //   them -- pick the height that fills k rounds exactly (1536^2: 4 rows = 0.98 rounds 24.0 us, 8 rows 25.3 us).
  {
    int pick = (params->nx <= 7168) ? 5 : (params->nx <= 8192 ? 7 : (params->nx <= 12288 ? 5 : 4));
    const long resident = 256L * 4 * 5;  // 2-cell waves resident at once
    const long slab_rows = (n_slabs > 1 || world > 1) ? (c->row_count / n_slabs) - 4 : c->row_count;
    const long rows_eff = slab_rows > 1 ? slab_rows : 1;
    if (c->lane_cells == 2 && (long)c->n_strips * ceil_div(rows_eff, 8) < 5 * resident) {
      const long lo = 3;
      long best_cost = -1;
      for (int k = 1; k <= 4; k++) {
        long b = (rows_eff * c->n_strips + k * resident - 1) / (k * resident);
        if (b < lo) b = lo;
        if (b > 32) b = 32;
        const long rounds = ((long)c->n_strips * ceil_div(rows_eff, b) + resident - 1) / resident;
        const long cost = rounds * (b + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; pick = (int)b; }
      }
    }
    if (c->lane_cells == 2 && c->pass_steps >= 3) {
      // two-cell packed kernel: these sizes are bound by latency, and the best height is the one that spreads the
      // slab over one round of two waves per SIMD (2048 waves): 768^2 3, 1024^2 5, 1152^2 6, 1280^2 7-8, 1536^2 10,
      // 1792^2 14-16 rows (profiles/r02_tuning.md)
      const long interior = (n_slabs > 1 || world > 1 || halo_on) ? rows_eff + 4 - 2 * c->pass_steps : rows_eff;
      long b = ((interior > 1 ? interior : 1) * c->n_strips + 2047) / 2048;
      pick = (int)(b < 3 ? 3 : (b > 64 ? 64 : b));
    }
    if (c->lane_cells == 4 && c->pass_steps >= 3) {
      // K >= 3 (2 waves per SIMD, bound by instruction issue): the waves run in rounds of 2048 and every wave of a round
      // takes band + 2(K-1) row iterations, so the cost of a band height is rounds x iterations (8192^2, K = 4: 46 rows
      // = 6086 waves = 2.97 rounds 0.274 ms per step; 55 rows = 2.47 rounds 0.288; 58 rows 0.300; 64 rows 0.309;
      // 24 / 32 rows 0.293 / 0.296; K = 3: 44 rows = 3.10 rounds 0.389, 46 rows 0.350).  Round 3: heights up to 160
      // rows -- ONE round of 2040 waves at 8192^2 (137 rows: 6 warm-up rows per 137 instead of per 46) 0.2691 vs
      // 0.2757 at 46, 0.2736 at 69 (two rounds), 0.284 at 92, 0.334 at 119, 0.321 at 180 (same box).
      const long interior = (n_slabs > 1 || world > 1 || halo_on) ? rows_eff + 4 - 2 * c->pass_steps : rows_eff;
      const long r_int = interior > 1 ? interior : 1;
      const int warm = 2 * (c->pass_steps - 1);
      if ((long)c->n_strips * ceil_div(r_int, 24) >= 16L * 1024) {
        pick = 32;  // many rounds (XCD-chunked order): flat in the height, 16384^2 24 / 32 / model (48) = 1.078 / 1.077 / 1.098
      } else {
        // rounds of 2048 resident waves; a last round that fills at most half of the slots leaves one wave per SIMD,
        // which then runs at nearly twice the speed
        double best = -1.0;
        const int b_max = env_int("LBM_BAND_MAX", 160);
        for (int b = 8; b <= b_max; b++) {
          const long waves = (long)c->n_strips * ceil_div(r_int, b);
          const long full = waves / 2048, rest = waves % 2048;
          double rounds = (double)full + (rest == 0 ? 0.0 : (rest > 1024 ? 1.0 : 0.6));
          if (rounds < 1.0) rounds = 1.0;  // a lone wave on a SIMD hides no latency
          const double cost = rounds * (b + warm);
          if (best < 0.0 || cost < best) { best = cost; pick = b; }
        }
      }
    }
    c->band_rows = env_int("LBM_BAND_ROWS", pick);
  }
  if (c->band_rows < 1) c->band_rows = 1;
  for (int s = 0; s < n_slabs; s++) {
    const int waves = c->n_strips * (ceil_div(c->slab[s].rows, c->band_rows) + 2);
    if (c->fuse2 && waves > max_blocks) max_blocks = waves;
  }
  // LDS-tile kernel (several timesteps per launch) for small single-slab grids: LBM_TILE_STEPS overrides
  if (!halo_on) {
    // measured (us per step; one-step kernels | 16x8 tiles, 4 steps per launch | 32x16 tiles, 3 steps per launch):
    //   128^2 3.14 | 2.09 | -      256^2 3.84 | 2.82 | 3.54    384^2 5.65 | 4.19 | 5.43    448^2 6.14 | 5.44 | 5.26
    //   512^2 6.56 | 6.02 | 5.34   640^2 9.38 | 8.34 | 8.89    768^2 11.28 | 11.22 | 10.14  896^2 12.70 | 14.7 | 13.8
    //   1024^2 13.34 | 18.8 | 15.1 -- from there the redundant halo work costs more than the launches it saves
    // (asking for one of the other kernels by LBM_FUSE2 / LBM_VEC4 takes the tile kernel out of the default)
    const bool other_kernel_requested = getenv("LBM_FUSE2") || getenv("LBM_VEC4");
    const long cells = (long)params->nx * params->ny;
    const int dflt_shape = (cells <= 200L * 1024) ? 0 : 3;
    const int dflt_steps = (other_kernel_requested || cells >= 300L * 1024) ? 0 : kTileShapes[dflt_shape].kmax;
    c->tile_steps = env_int("LBM_TILE_STEPS", dflt_steps);
    c->tile_shape = env_int("LBM_TILE_SHAPE", getenv("LBM_TILE_STEPS") ? (c->tile_steps > 4 ? 1 : 0) : dflt_shape);
    if (c->tile_shape < 0 || c->tile_shape >= kTileShapeCount) c->tile_shape = 0;
    if (c->tile_steps < 0 || c->tile_steps > kTileShapes[c->tile_shape].kmax) c->tile_steps = kTileShapes[c->tile_shape].kmax;
    if (c->tile_steps && tile_count_for(params, c->tile_shape) > max_blocks) max_blocks = tile_count_for(params, c->tile_shape);
    // a graph chunk is kPartSlots timesteps in an even number of passes
  }
  c->part_stride = round_up(max_blocks, 64);

  // Resident kernel (lbm::resident_band): one launch advances up to kResidentChunk timesteps with the lattice in
  // registers, bands of 4 rows x the full width per workgroup, seam rows through L2 granules.  For single periodic
  // slabs whose bands are all co-resident (at most one workgroup per CU of the device) and whose rows are one lane
  // per cell wide: the reference's own data sets (128x128 ... 1024x1024).  Both math modes run it (its arithmetic is
  // the exact one, which meets the fast mode's tolerance and is the faster kernel at these sizes).  Asking for
  // another kernel by any of the selection knobs leaves it off unless LBM_RESIDENT=1 says otherwise.
  if (!halo_on && n_slabs == 1) {
    static const char* const selectors[] = {"LBM_FUSE2", "LBM_VEC4", "LBM_TILE_STEPS", "LBM_TILE_SHAPE", "LBM_LANE_CELLS", "LBM_PASS_STEPS",
                                            "LBM_PACKED", "LBM_BAND_ROWS", "LBM_GRAPH", "LBM_STEPK", "LBM_LDS_WINDOWS", "LBM_PREFETCH",
                                            "LBM_XCD_CHUNK", "LBM_NEIGH", "LBM_SNAKE", "LBM_NTS"};
    bool other_kernel = false;
    for (const char* name : selectors) other_kernel = other_kernel || getenv(name) != nullptr;
    const int nx = params->nx, ny = params->ny;
    int cus = 0;
    const int dev = c->slab[0].device;
    if (hipSetDevice(dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    // rows per band: 2 (one pair per lane) where every band still gets a CU of its own, else 4 (us per step, 4 | 2 rows:
    // 128^2 1.88 | 1.54, 128x256 1.93 | 1.59, 256^2 2.03 | 1.62, 512^2 2.73 | 2.17, 1024x512 4.35 | 3.43)
    int rows = env_int("LBM_RESIDENT_ROWS", 0);
    if (rows != 2 && rows != 4) rows = (ny % 2 == 0 && ny / 2 <= cus && ny >= 4) ? 2 : 4;
    const bool shape_ok = nx % 64 == 0 && nx >= 64 && nx <= 1024 && ny % rows == 0 && ny >= 2 * rows;
    if (shape_ok && env_int("LBM_RESIDENT", other_kernel ? 0 : 1)) {
      int per_cu = 0;
      const int joint = (rows == 4 && nx <= 512 && env_int("LBM_RESIDENT_JOINT", nx <= 256 ? 1 : 0)) ? 1 : 0;
      if (cus > 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_kernel(nx, rows, joint), nx, 0) == hipSuccess &&
          per_cu >= 1 && ny / rows <= cus) {
        c->resident = 1;
        c->resident_rows = rows;
        c->resident_bands = ny / rows;
        c->resident_joint = joint;
        // Grids of at most 128 waves (two-row bands): everything on ONE XCD, one wave per SIMD -- workgroups of four
        // waves (1, 2 or 4 bands side by side), at most one per CU of the XCD; 8 x the workgroups are launched and
        // those not dealt to the first XCD leave at once.  Then no seam crosses the fabric (hand-off 0.29 instead of
        // 0.63 us, tools/hop_flavours.hip): 128^2 1.10 vs 1.36 us per step, 64x128 1.06 vs 1.34, 128x64 1.05 vs 1.42
        // (two bands per workgroup alone: no change; one XCD with two workgroups per CU: none either).
        {
          const int waves_per_band = nx / 64, cus_per_xcd = cus / 8;
          int group = 1, one_xcd = 0;
          if (rows == 2 && waves_per_band <= 4) {
            for (int g = 1; g * waves_per_band <= 4 && !one_xcd; g *= 2)
              if (c->resident_bands % g == 0 && c->resident_bands / g <= cus_per_xcd) { group = g; one_xcd = 1; }
          }
          one_xcd = env_int("LBM_RESIDENT_ONE_XCD", one_xcd) ? 1 : 0;
          group = env_int("LBM_RESIDENT_GROUP", one_xcd ? group : 1);
          if (group < 1 || c->resident_bands % group != 0 || nx * group > (nx > 512 ? 1024 : 512)) group = 1;
          c->resident_group = group;
          c->resident_one_xcd = one_xcd;
        }
        // a launch costs about 20 us before its first step (lattice into registers, back out, reduce, status copy);
        // measured wall time of one lbm_run(n) + sync, per-pass kernels | resident (tools/resident_crossover.py):
        // 128^2 n = 4 24.5 | 27.0, n = 8 32.9 | 32.4, n = 16 50.0 | 44.4; 256^2 n = 4 28.8 | 28.4, n = 8 41.0 | 35.4;
        // 1024^2 n = 4 59.8 | 50.4, n = 8 96.2 | 69.1
        const long cells = (long)nx * ny;
        c->resident_min_steps = env_int("LBM_RESIDENT_MIN_STEPS", cells >= 48L * 1024 ? 4 : 8);
        if (c->resident_min_steps < 1) c->resident_min_steps = 1;
        c->resident_timeout = (long long)env_int("LBM_RESIDENT_TIMEOUT_MS", 2000) * 100000LL;  // wall_clock64(): 100 MHz
      }
    }
  }

  for (int s = 0; s < n_slabs; s++)
    if (build_slab(c, s, obst, cells_aos) != LBM_SUCCESS) {
      lbm_destroy(c);
      return nullptr;
    }

  if (c->hosted) {
    for (int i = 0; i < 2; i++) {
      const size_t bytes = (size_t)kHaloRows * c->row_pitch * sizeof(float);
      if (hipHostMalloc(&c->host_send[i], bytes) != hipSuccess || hipHostMalloc(&c->host_recv[i], bytes) != hipSuccess) {
        raise_error(__LINE__, "lbm_create_rank_hosted: cannot allocate the pinned exchange buffers");
        lbm_destroy(c);
        return nullptr;
      }
    }
  } else if (c->ranked) {
    // one process per GPU: the communicator spans the ranks (also used for the av_vels reduce)
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RcclApi* nc = rccl();
    ncclResult_t res = ncclSuccess;
    if (!nc || hipSetDevice(c->slab[0].device) != hipSuccess ||
        (res = nc->CommInitRank(&c->slab[0].nccl, world, id, rank)) != ncclSuccess) {
      raise_error(__LINE__, "lbm_create_rank: ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                  nc ? nc->GetErrorString(res) : g_rccl_error);
      lbm_destroy(c);
      return nullptr;
    }
  } else if (c->halo == HALO_RCCL) {
    ncclComm_t comms[kMaxSlabs];
    int devs[kMaxSlabs];
    for (int s = 0; s < n_slabs; s++) devs[s] = c->slab[s].device;
    RcclApi* nc = rccl();
    ncclResult_t res = ncclSuccess;
    if (!nc || (res = nc->CommInitAll(comms, n_slabs, devs)) != ncclSuccess) {
      raise_error(__LINE__, "lbm_create: ncclCommInitAll failed: %s (set LBM_HALO=memcpy when slabs share a device)",
                  nc ? nc->GetErrorString(res) : g_rccl_error);
      lbm_destroy(c);
      return nullptr;
    }
    for (int s = 0; s < n_slabs; s++) c->slab[s].nccl = comms[s];
  }
  // global number of fluid cells (av_velocity's divisor): the slabs' device-side counts, summed over the ranks
  // (the reference counts on rank 0 while parsing, MPI_Waitall/d2q9-bgk.c:794-804)
  {
    long long fluid = 0;
    for (int s = 0; s < n_slabs; s++) fluid += c->slab[s].fluid_cells;
    if (c->hosted && world > 1) {
      double v = (double)fluid;  // exact: a cell count is below 2^31
      if (c->host_comm.allreduce_sum(c->host_comm.user, &v, 1) != 0) {
        raise_error(__LINE__, "lbm_create_rank_hosted: the host's all-reduce callback failed");
        lbm_destroy(c);
        return nullptr;
      }
      fluid = (long long)(v + 0.5);
    } else if (c->ranked && world > 1) {
      Slab& sl = c->slab[0];
      long long* dev = reinterpret_cast<long long*>(sl.scratch);
      if (hipSetDevice(sl.device) != hipSuccess ||
          hipMemcpy(dev, &fluid, sizeof(fluid), hipMemcpyHostToDevice) != hipSuccess ||
          !rccl() || g_rccl.AllReduce(dev, dev, 1, ncclInt64, ncclSum, sl.nccl, sl.comm) != ncclSuccess ||
          hipStreamSynchronize(sl.comm) != hipSuccess ||
          hipMemcpy(&fluid, dev, sizeof(fluid), hipMemcpyDeviceToHost) != hipSuccess) {
        raise_error(__LINE__, "lbm_create_rank: all-reduce of the fluid-cell count failed");
        lbm_destroy(c);
        return nullptr;
      }
    }
    c->fluid_cells = (int)fluid;
  }
  // One issuing thread per slab when one process drives several slabs on DISTINCT devices (LBM_GPUS=n on a multi-GPU
  // node): a pass enqueues ~10 runtime calls per slab, 25-30 us on one thread -- more than an 8-GPU pass of 8192^2
  // takes on the devices.  With several slabs on ONE device it is slower (the runtime serialises calls to a device:
  // 65 vs 53 us per step for 2 slabs), so there it stays opt-in.  LBM_THREADS=0/1 overrides.
  bool distinct_devices = n_slabs > 1;
  for (int a = 0; a < n_slabs; a++)
    for (int b = a + 1; b < n_slabs; b++)
      if (c->slab[a].device == c->slab[b].device) distinct_devices = false;
  const bool want_team = n_slabs > 1 && env_int("LBM_THREADS", distinct_devices ? 1 : 0);
  // hipGraph replay of the halo pipeline (both streams of every slab, RCCL send/recv or device copies inside the
  // capture) exists (capture_chunk) but is OFF unless LBM_GRAPH=1: measured on MI355X / ROCm 7.2 it buys nothing
  // (host issue 11.1 vs 11.6 us per step for a rank with RCCL self-exchange at 256^2: the runtime still enqueues every
  // node) and hipGraphInstantiate overflows its stack on the larger pipelines (3+ slabs with device-copy halos, a
  // rank's 20-pass chunk at 8192x1024) -- profiles/r02_tuning.md.  The device-copy transport never uses it.
  if (c->halo != HALO_SELF && (!getenv("LBM_GRAPH") || c->halo == HALO_HOST)) c->use_graph = 0;
  if (want_team) {
    c->use_graph = 0;
    c->team = new SlabTeam();
    c->team->start(n_slabs);
  }
  return c;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

void lbm_set_error_mode(int mode) { g_error_mode = (mode == LBM_ERRORS_RETURN) ? LBM_ERRORS_RETURN : LBM_ERRORS_DIE; }
const char* lbm_last_error(void) { return g_last_error; }
const char* lbm_version(void) { return "lbm_hip 0.1 gfx950"; }

int lbm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lbm_partition_rows(int ny, int parts, int index, int* first, int* count) {
  if (parts < 1 || index < 0 || index >= parts || ny < 1)
    LBM_FAIL(LBM_FAILURE, "lbm_partition_rows: bad arguments (ny=%d parts=%d index=%d)", ny, parts, index);
  const int base = ny / parts, rem = ny % parts;
  const int cnt = base + (index < rem ? 1 : 0);
  const int fst = index * base + (index < rem ? index : rem);
  if (parts > 1 && base < 2)  // some part (not necessarily this one) would be too thin
    LBM_FAIL(LBM_FAILURE, "lbm_partition_rows: %d rows over %d parts leaves a part with fewer than 2 rows", ny, parts);
  if (first) *first = fst;
  if (count) *count = cnt;
  return LBM_SUCCESS;
}

int lbm_halo_plan(int rows, int parts, int index, int depth, lbm_halo_op out[4]) {
  if (!out || rows < 1 || parts < 1 || index < 0 || index >= parts || depth < 1 || depth > rows)
    LBM_FAIL(LBM_FAILURE, "lbm_halo_plan: bad arguments (rows=%d parts=%d index=%d depth=%d)", rows, parts, index, depth);
  // ring with periodic wrap (MPI/d2q9-bgk.c:210-211): north = the part above, south = the part below
  const int north = (index + 1) % parts, south = (index - 1 + parts) % parts;
  out[0] = {1, north, rows - depth, depth};  // my top rows     -> north's rows [-depth, 0)
  out[1] = {1, south, 0, depth};             // my bottom rows  -> south's rows [rows_s, rows_s + depth)
  out[2] = {0, south, -depth, depth};        // my south halo  <-  south's top rows
  out[3] = {0, north, rows, depth};          // my north halo  <-  north's bottom rows
  return LBM_SUCCESS;
}

int lbm_plan_halo_depth(const lbm_params* params, int parts, int math_mode) {
  if (!validate_params(params) || parts < 1 || (math_mode != LBM_MATH_EXACT && math_mode != LBM_MATH_FAST))
    LBM_FAIL(0, "lbm_plan_halo_depth: bad arguments");
  const StreamPlan pl = plan_stream(params, parts, true, math_mode);
  return pl.fuse2 ? pl.pass_steps : 1;
}

lbm_ctx* lbm_create(const lbm_params* params, const int* obstacles, const float* cells_aos,
                    int n_gpus, int math_mode) {
  const ObstacleSource obst = {OBST_GLOBAL, obstacles, 0, 0, false};
  return create_common(params, obst, cells_aos, n_gpus, math_mode, 0, 1, nullptr, 0);
}

lbm_ctx* lbm_create_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                          const float* cells_aos, int n_gpus, int math_mode) {
  const ObstacleSource obst = {OBST_TILE, tile, tile_nx, tile_ny, false};
  return create_common(params, obst, cells_aos, n_gpus, math_mode, 0, 1, nullptr, 0);
}

int lbm_rccl_unique_id(void* id_out) {
  if (!id_out) LBM_FAIL(LBM_FAILURE, "lbm_rccl_unique_id: NULL output");
  static_assert(sizeof(ncclUniqueId) == LBM_RCCL_ID_BYTES, "RCCL unique id size");
  ncclUniqueId id;
  RCCL_OR_FAIL(LBM_FAILURE);
  NCCL_TRY(LBM_FAILURE, rc_api_->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return LBM_SUCCESS;
}

int lbm_rccl_info(const lbm_ctx* c, lbm_rccl_status* out) {
  if (!out) LBM_FAIL(LBM_FAILURE, "lbm_rccl_info: NULL output");
  memset(out, 0, sizeof(*out));
  // without a context: bind the library (as the first multi-GPU create would) and describe it; with one: describe
  // what the context uses, binding nothing on behalf of a context that never needed RCCL
  RcclApi* nc = nullptr;
  if (!c) {
    nc = rccl();
    if (!nc) LBM_FAIL(LBM_FAILURE, "RCCL is not available: %s", g_rccl_error);
  } else {
    for (int s = 0; s < c->n_slabs; s++) if (c->slab[s].nccl) out->n_comms++;
    if (out->n_comms > 0) nc = rccl();
  }
  if (!nc) return LBM_SUCCESS;
  out->loaded = 1;
  strncpy(out->library, nc->path, sizeof(out->library) - 1);
  NCCL_TRY(LBM_FAILURE, nc->GetVersion(&out->version));
  if (c && out->n_comms > 0) {
    for (int s = 0; s < c->n_slabs; s++)
      if (c->slab[s].nccl) {
        NCCL_TRY(LBM_FAILURE, nc->CommCount(c->slab[s].nccl, &out->nranks));
        NCCL_TRY(LBM_FAILURE, nc->CommUserRank(c->slab[s].nccl, &out->rank));
        break;
      }
  }
  return LBM_SUCCESS;
}

static bool rank_args_ok(int rank, int world_size, const void* unique_id) {
  if (world_size < 1 || rank < 0 || rank >= world_size) {
    raise_error(__LINE__, "lbm_create_rank: bad rank %d of %d", rank, world_size);
    return false;
  }
  if (!unique_id) {
    raise_error(__LINE__, "lbm_create_rank: unique_id is NULL");
    return false;
  }
  return true;
}

lbm_ctx* lbm_create_rank(const lbm_params* params, const int* obstacles, const float* cells_aos,
                         int rank, int world_size, const void* unique_id, int device, int math_mode) {
  if (!rank_args_ok(rank, world_size, unique_id)) return nullptr;
  const ObstacleSource obst = {OBST_GLOBAL, obstacles, 0, 0, false};
  return create_common(params, obst, cells_aos, 1, math_mode, rank, world_size, unique_id, device);
}

lbm_ctx* lbm_create_rank_rows(const lbm_params* params, const int* obstacle_rows, const float* cells_rows_aos,
                              int rank, int world_size, const void* unique_id, int device, int math_mode) {
  if (!rank_args_ok(rank, world_size, unique_id)) return nullptr;
  const ObstacleSource obst = {OBST_ROWS, obstacle_rows, 0, 0, true};
  return create_common(params, obst, cells_rows_aos, 1, math_mode, rank, world_size, unique_id, device);
}

static bool hosted_args_ok(int rank, int world_size, const lbm_host_comm* comm) {
  if (world_size < 1 || rank < 0 || rank >= world_size) {
    raise_error(__LINE__, "lbm_create_rank_hosted: bad rank %d of %d", rank, world_size);
    return false;
  }
  if (!comm || !comm->exchange || !comm->allreduce_sum) {
    raise_error(__LINE__, "lbm_create_rank_hosted: the exchange and all-reduce callbacks are required");
    return false;
  }
  return true;
}

lbm_ctx* lbm_create_rank_hosted(const lbm_params* params, const int* obstacles, const float* cells_aos,
                                int rank, int world_size, const lbm_host_comm* comm, int device, int math_mode) {
  if (!hosted_args_ok(rank, world_size, comm)) return nullptr;
  const ObstacleSource obst = {OBST_GLOBAL, obstacles, 0, 0, false};
  return create_common(params, obst, cells_aos, 1, math_mode, rank, world_size, nullptr, device, comm);
}

lbm_ctx* lbm_create_rank_hosted_rows(const lbm_params* params, const int* obstacle_rows, const float* cells_rows_aos,
                                     int rank, int world_size, const lbm_host_comm* comm, int device, int math_mode) {
  if (!hosted_args_ok(rank, world_size, comm)) return nullptr;
  const ObstacleSource obst = {OBST_ROWS, obstacle_rows, 0, 0, true};
  return create_common(params, obst, cells_rows_aos, 1, math_mode, rank, world_size, nullptr, device, comm);
}

lbm_ctx* lbm_create_rank_hosted_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                                      int rank, int world_size, const lbm_host_comm* comm, int device, int math_mode) {
  if (!hosted_args_ok(rank, world_size, comm)) return nullptr;
  const ObstacleSource obst = {OBST_TILE, tile, tile_nx, tile_ny, false};
  return create_common(params, obst, nullptr, 1, math_mode, rank, world_size, nullptr, device, comm);
}

lbm_ctx* lbm_create_rank_tiled(const lbm_params* params, const int* tile, int tile_nx, int tile_ny,
                               int rank, int world_size, const void* unique_id, int device, int math_mode) {
  if (!rank_args_ok(rank, world_size, unique_id)) return nullptr;
  const ObstacleSource obst = {OBST_TILE, tile, tile_nx, tile_ny, false};
  return create_common(params, obst, nullptr, 1, math_mode, rank, world_size, unique_id, device);
}

void lbm_destroy(lbm_ctx* c) {
  if (!c) return;
  if (c->team) {
    c->team->shutdown();
    delete c->team;
    c->team = nullptr;
  }
  for (int s = 0; s < c->n_slabs; s++) {
    if (c->slab[s].compute) {
      (void)hipSetDevice(c->slab[s].device);
      (void)hipStreamSynchronize(c->slab[s].compute);
      (void)hipStreamSynchronize(c->slab[s].comm);
    }
  }
  for (int s = 0; s < c->n_slabs; s++) free_slab(c->slab[s]);
  for (int i = 0; i < 2; i++) {
    if (c->host_send[i]) (void)hipHostFree(c->host_send[i]);
    if (c->host_recv[i]) (void)hipHostFree(c->host_recv[i]);
  }
  delete c;
}

int lbm_get_info(const lbm_ctx* c, lbm_info* out) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_get_info: NULL argument");
  out->n_slabs = c->n_slabs;
  out->row_first = c->row_first;
  out->row_count = c->row_count;
  out->fluid_cells = c->fluid_cells;
  out->steps_done = c->steps_done;
  out->math_mode = c->math_mode;
  out->world_rank = c->rank;
  out->world_size = c->world;
  const bool stale = (c->halo != HALO_SELF && c->halo_mode != LBM_HALO_SYNC);
  out->steps_per_launch = (c->tile_steps && c->halo == HALO_SELF) ? c->tile_steps : ((c->fuse2 && !stale) ? c->pass_steps : 1);
  out->halo_mode = c->halo_mode;
  const bool stream_kernel = c->fuse2 && !stale && !(c->tile_steps && c->halo == HALO_SELF);
  out->band_rows = stream_kernel ? c->band_rows : 0;
  out->lane_cells = stream_kernel ? c->lane_cells : 0;
  out->nontemporal = c->nts;
  {
    int adv = 1;
    const int passes = chunk_passes(c, &adv);
    out->graph_steps = (c->use_graph && !stale) ? passes * adv : 0;
  }
  out->resident_steps = c->resident ? kResidentChunk : 0;
  out->resident_min_steps = c->resident ? c->resident_min_steps : 0;
  out->resident_rows = c->resident ? c->resident_rows : 0;
  out->resident_group = c->resident ? c->resident_group : 0;
  out->resident_one_xcd = c->resident ? c->resident_one_xcd : 0;
  return LBM_SUCCESS;
}

int lbm_set_halo_mode(lbm_ctx* c, int mode) {
  if (!c) LBM_FAIL(LBM_FAILURE, "lbm_set_halo_mode: null context");
  if (mode != LBM_HALO_SYNC && mode != LBM_HALO_STALE && mode != LBM_HALO_FRESHEST) LBM_FAIL(LBM_FAILURE, "lbm_set_halo_mode: unknown mode %d", mode);
  if (mode == LBM_HALO_FRESHEST && c->halo == HALO_HOST) LBM_FAIL(LBM_FAILURE, "lbm_set_halo_mode: the freshest-available mode is not available with the hosted exchange");
  c->halo_mode = mode;
  return LBM_SUCCESS;
}

int lbm_run(lbm_ctx* c, int n_steps) { return run_steps(c, n_steps, nullptr); }

int lbm_run_timed(lbm_ctx* c, int n_steps, float* kernel_ms_per_step) {
  if (!kernel_ms_per_step) LBM_FAIL(LBM_FAILURE, "lbm_run_timed: NULL output");
  return run_steps(c, n_steps, kernel_ms_per_step);
}

int lbm_sync(lbm_ctx* c) {
  if (!c) LBM_FAIL(LBM_FAILURE, "lbm_sync: null context");
  for (int s = 0; s < c->n_slabs; s++) {
    HIP_TRY(LBM_FAILURE, hipSetDevice(c->slab[s].device));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(c->slab[s].compute));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(c->slab[s].comm));
  }
  if (c->resident_used) {
    // did every workgroup of the resident kernel get its neighbours' rows in time?
    const int status = *c->slab[0].res_status_host;
    c->resident_used = false;
    if (status != 0)
      LBM_FAIL(LBM_FAILURE, "the resident kernel gave up waiting for a neighbouring band after %.0f ms (status %d): its %d workgroups "
               "were not all running at once -- is another process using the device?  The lattice of this context is no "
               "longer valid; LBM_RESIDENT=0 selects the launch-per-pass kernels, LBM_RESIDENT_TIMEOUT_MS moves the bound",
               (double)c->resident_timeout / 1e5, status, c->resident_bands);
  }
  return LBM_SUCCESS;
}

int lbm_read_halo_log(lbm_ctx* c, unsigned char* out, int n) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_read_halo_log: NULL argument");
  if (n < 0 || n > c->steps_done) LBM_FAIL(LBM_FAILURE, "lbm_read_halo_log: %d steps requested, %d recorded", n, c->steps_done);
  if (n == 0) return LBM_SUCCESS;
  if (lbm_sync(c) != LBM_SUCCESS) return LBM_FAILURE;
  std::vector<unsigned char> part((size_t)n);
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    if (sl.fresh_log) {
      HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
      HIP_TRY(LBM_FAILURE, hipMemcpy(part.data(), sl.fresh_log, (size_t)n, hipMemcpyDeviceToHost));
    } else {
      part.assign((size_t)n, 3);  // the mode was never used: every halo row was the row of its step
    }
    for (int t = 0; t < n; t++) out[(size_t)t * c->n_slabs + s] = part[(size_t)t];
  }
  return LBM_SUCCESS;
}

int lbm_read_av_vels(lbm_ctx* c, float* out, int n) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_read_av_vels: NULL argument");
  if (n < 0 || n > c->steps_done) LBM_FAIL(LBM_FAILURE, "lbm_read_av_vels: %d steps requested, %d recorded", n, c->steps_done);
  if (n == 0) return LBM_SUCCESS;
  if (lbm_sync(c) != LBM_SUCCESS) return LBM_FAILURE;
  std::vector<double> total((size_t)n, 0.0), part((size_t)n);
  for (int s = 0; s < c->n_slabs; s++) {
    HIP_TRY(LBM_FAILURE, hipSetDevice(c->slab[s].device));
    HIP_TRY(LBM_FAILURE, hipMemcpy(part.data(), c->slab[s].tot_u, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    for (int t = 0; t < n; t++) total[(size_t)t] += part[(size_t)t];
  }
  if (c->hosted) {
    // the reference's MPI_Reduce(av_vels, SUM) (MPI/d2q9-bgk.c:298-309) through the host's own all-reduce
    if (c->world > 1 && c->host_comm.allreduce_sum(c->host_comm.user, total.data(), n) != 0)
      LBM_FAIL(LBM_FAILURE, "the host's all-reduce callback failed");
  } else if (c->ranked) {
    // the reference's MPI_Reduce(av_vels, SUM) (MPI/d2q9-bgk.c:298-309), as an all-reduce
    Slab& sl = c->slab[0];
    double* tmp = sl.reduce_buf;  // allocated once at create: nothing to leak on an error return
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipMemcpy(tmp, total.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    RCCL_OR_FAIL(LBM_FAILURE);
    NCCL_TRY(LBM_FAILURE, rc_api_->AllReduce(tmp, tmp, (size_t)n, ncclDouble, ncclSum, sl.nccl, sl.comm));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.comm));
    HIP_TRY(LBM_FAILURE, hipMemcpy(total.data(), tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  }
  const float cells = (float)c->fluid_cells;
  for (int t = 0; t < n; t++) out[t] = (float)total[(size_t)t] / cells;  // SerialCode/d2q9-bgk.c:457
  return LBM_SUCCESS;
}

int lbm_read_cells(lbm_ctx* c, float* cells_aos) {
  if (!c || !cells_aos) LBM_FAIL(LBM_FAILURE, "lbm_read_cells: NULL argument");
  if (lbm_sync(c) != LBM_SUCCESS) return LBM_FAILURE;
  const int nx = c->p.nx;
  long chunk_rows = (64L << 20) / ((long)nx * lbm::kQ * sizeof(float));
  if (chunk_rows < 1) chunk_rows = 1;
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    DeviceTemp stage_buf;
    HIP_TRY(LBM_FAILURE, hipMalloc(&stage_buf.p, (size_t)chunk_rows * nx * lbm::kQ * sizeof(float)));
    float* stage = stage_buf.as<float>();
    for (int r0 = 0; r0 < sl.rows; r0 += (int)chunk_rows) {
      const int nr = (sl.rows - r0 < chunk_rows) ? sl.rows - r0 : (int)chunk_rows;
      const size_t n = (size_t)nr * nx * lbm::kQ;
      hipLaunchKernelGGL(lbm::soa_to_aos, dim3(ceil_div((long)n, 256)), dim3(256), 0, sl.compute,
                         sl.lat[c->cur], stage, c->plane_stride, c->row_pitch, nx, r0, nr);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      HIP_TRY(LBM_FAILURE, hipMemcpyAsync(cells_aos + (size_t)(sl.row_first - c->row_first + r0) * nx * lbm::kQ, stage,
                                          n * sizeof(float), hipMemcpyDeviceToHost, sl.compute));
      HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    }
  }
  return LBM_SUCCESS;
}

int lbm_read_final_state(lbm_ctx* c, float* u_x, float* u_y, float* u_mag, float* pressure) {
  if (!c || !u_x || !u_y || !u_mag || !pressure) LBM_FAIL(LBM_FAILURE, "lbm_read_final_state: NULL argument");
  if (lbm_sync(c) != LBM_SUCCESS) return LBM_FAILURE;
  const int nx = c->p.nx;
  long chunk_rows = (16L << 20) / ((long)nx * sizeof(float));
  if (chunk_rows < 1) chunk_rows = 1;
  float* outs[4] = {u_x, u_y, u_mag, pressure};
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    DeviceTemp stage_buf;
    const size_t chunk_cells = (size_t)chunk_rows * nx;
    HIP_TRY(LBM_FAILURE, hipMalloc(&stage_buf.p, 4 * chunk_cells * sizeof(float)));
    float* stage = stage_buf.as<float>();
    for (int r0 = 0; r0 < sl.rows; r0 += (int)chunk_rows) {
      const int nr = (sl.rows - r0 < chunk_rows) ? sl.rows - r0 : (int)chunk_rows;
      const size_t n = (size_t)nr * nx;
      hipLaunchKernelGGL(lbm::final_state, dim3(ceil_div((long)n, 256)), dim3(256), 0, sl.compute,
                         sl.lat[c->cur], sl.mask, c->plane_stride, c->row_pitch, c->pitch, nx, r0, nr, c->p.density,
                         stage, stage + chunk_cells, stage + 2 * chunk_cells, stage + 3 * chunk_cells);
      HIP_TRY(LBM_FAILURE, hipGetLastError());
      const size_t off = (size_t)(sl.row_first - c->row_first + r0) * nx;
      for (int k = 0; k < 4; k++)
        HIP_TRY(LBM_FAILURE, hipMemcpyAsync(outs[k] + off, stage + k * chunk_cells, n * sizeof(float),
                                            hipMemcpyDeviceToHost, sl.compute));
      HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    }
  }
  return LBM_SUCCESS;
}

// global sums of |u| over fluid cells and of density over all cells
static int lattice_totals(lbm_ctx* c, double* speed, double* mass) {
  if (lbm_sync(c) != LBM_SUCCESS) return LBM_FAILURE;
  double tot[2] = {0.0, 0.0};
  std::vector<double> h(2 * kSumBlocks);
  for (int s = 0; s < c->n_slabs; s++) {
    Slab& sl = c->slab[s];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    hipLaunchKernelGGL(lbm::lattice_sums, dim3(kSumBlocks), dim3(lbm::kBlock), 0, sl.compute, sl.lat[c->cur],
                       sl.mask, c->plane_stride, c->row_pitch, c->pitch, c->p.nx, sl.rows, sl.scratch,
                       sl.scratch + kSumBlocks);
    HIP_TRY(LBM_FAILURE, hipGetLastError());
    HIP_TRY(LBM_FAILURE, hipMemcpyAsync(h.data(), sl.scratch, 2 * kSumBlocks * sizeof(double),
                                        hipMemcpyDeviceToHost, sl.compute));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.compute));
    for (int i = 0; i < kSumBlocks; i++) { tot[0] += h[i]; tot[1] += h[kSumBlocks + i]; }
  }
  if (c->hosted) {
    if (c->world > 1 && c->host_comm.allreduce_sum(c->host_comm.user, tot, 2) != 0)
      LBM_FAIL(LBM_FAILURE, "the host's all-reduce callback failed");
  } else if (c->ranked) {
    Slab& sl = c->slab[0];
    HIP_TRY(LBM_FAILURE, hipSetDevice(sl.device));
    HIP_TRY(LBM_FAILURE, hipMemcpy(sl.scratch, tot, 2 * sizeof(double), hipMemcpyHostToDevice));
    RCCL_OR_FAIL(LBM_FAILURE);
    NCCL_TRY(LBM_FAILURE, rc_api_->AllReduce(sl.scratch, sl.scratch, 2, ncclDouble, ncclSum, sl.nccl, sl.comm));
    HIP_TRY(LBM_FAILURE, hipStreamSynchronize(sl.comm));
    HIP_TRY(LBM_FAILURE, hipMemcpy(tot, sl.scratch, 2 * sizeof(double), hipMemcpyDeviceToHost));
  }
  *speed = tot[0];
  *mass = tot[1];
  return LBM_SUCCESS;
}

int lbm_av_velocity(lbm_ctx* c, float* out) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_av_velocity: NULL argument");
  double speed, mass;
  if (lattice_totals(c, &speed, &mass) != LBM_SUCCESS) return LBM_FAILURE;
  *out = (float)speed / (float)c->fluid_cells;
  return LBM_SUCCESS;
}

int lbm_total_density(lbm_ctx* c, double* out) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_total_density: NULL argument");
  double speed, mass;
  if (lattice_totals(c, &speed, &mass) != LBM_SUCCESS) return LBM_FAILURE;
  *out = mass;
  return LBM_SUCCESS;
}

int lbm_calc_reynolds(lbm_ctx* c, float* out) {
  if (!c || !out) LBM_FAIL(LBM_FAILURE, "lbm_calc_reynolds: NULL argument");
  float av;
  if (lbm_av_velocity(c, &av) != LBM_SUCCESS) return LBM_FAILURE;
  const float viscosity = 1.f / 6.f * (2.f / c->p.omega - 1.f);  // SerialCode/d2q9-bgk.c:639
  *out = av * c->p.reynolds_dim / viscosity;                     // :641
  return LBM_SUCCESS;
}

}  // extern "C"
