// tune_step.hip -- standalone timing harness for variants of the fused step kernel (gfx950).
//
// Not part of the product: it exists to pick launch geometry, load/store flavour and plane
// padding by measurement on the real chip (cdna_hip_programming.md section 5.4, rule 24:
// variants interleaved in ONE process).  Results are recorded under profiles/.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/tune_step.hip -o tools/tune_step
//   tools/tune_step [nx ny rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../lbm-asynchronous_amd/csrc/lbm_kernels.hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

using namespace lbm;

// plain float4 copy of the same byte volume: the streaming ceiling on this chip
__global__ __launch_bounds__(kBlock) void copy4(const float4* __restrict__ s, float4* __restrict__ d, long n) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) d[i] = s[i];
}


// decomposition probe: a 9-plane copy that adds the step kernel's access features one at a time
// ROWS: pull planes 2,5,6 / 4,7,8 from the rows below / above; EDGE: the 6 neighbour dword loads;
// MASK: the uchar4 mask load; SUM: the per-workgroup reduction + partial store
template <bool ROWS, bool EDGE, bool MASK, bool SUM>
__global__ __launch_bounds__(kBlock) void probe(const StepArgs a) {
  const int quads_x = a.nx >> 2;
  const long q = (long)blockIdx.x * kBlock + threadIdx.x;
  const long n_quads = (long)quads_x * a.n_rows;
  float acc = 0.f;
  if (q < n_quads) {
    const int row = (int)(q / quads_x);
    const int x0 = (int)(q - (long)row * quads_x) << 2;
    const long ps = a.plane_stride;
    const int rs = ROWS ? ((row == 0) ? a.rows - 1 : row - 1) : row;
    const int rn = ROWS ? ((row == a.rows - 1) ? 0 : row + 1) : row;
    const float* c = a.src + (long)row * a.row_pitch;
    const float* sb = a.src + (long)rs * a.row_pitch;
    const float* nb = a.src + (long)rn * a.row_pitch;
    float4 v[kQ];
    v[0] = *reinterpret_cast<const float4*>(c + x0);
    v[1] = *reinterpret_cast<const float4*>(c + 1 * ps + x0);
    v[3] = *reinterpret_cast<const float4*>(c + 3 * ps + x0);
    v[2] = *reinterpret_cast<const float4*>(sb + 2 * ps + x0);
    v[5] = *reinterpret_cast<const float4*>(sb + 5 * ps + x0);
    v[6] = *reinterpret_cast<const float4*>(sb + 6 * ps + x0);
    v[4] = *reinterpret_cast<const float4*>(nb + 4 * ps + x0);
    v[7] = *reinterpret_cast<const float4*>(nb + 7 * ps + x0);
    v[8] = *reinterpret_cast<const float4*>(nb + 8 * ps + x0);
    if constexpr (EDGE) {
      const int xw = (x0 == 0) ? a.nx - 1 : x0 - 1;
      const int xe = (x0 + 4 == a.nx) ? 0 : x0 + 4;
      v[1].x += c[1 * ps + xw];  v[3].w += c[3 * ps + xe];
      v[5].x += sb[5 * ps + xw]; v[6].w += sb[6 * ps + xe];
      v[7].w += nb[7 * ps + xe]; v[8].x += nb[8 * ps + xw];
    }
    if constexpr (MASK) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(a.mask + (long)row * a.pitch + x0);
      if (m.x | m.y | m.z | m.w) v[0].x = -v[0].x;
    }
    float* d = a.dst + (long)row * a.row_pitch + x0;
#pragma unroll
    for (int k = 0; k < kQ; k++) *reinterpret_cast<float4*>(d + k * ps) = v[k];
    acc = v[0].x;
  }
  if constexpr (SUM) {
    const float total = block_sum(acc);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = total;
  }
}

__global__ void fill_const(float* p, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

typedef void (*kern_t)(const StepArgs);
struct Variant {
  const char* name;
  kern_t k;
  int block = kBlock;
  int snake = 0;
};

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 8192;
  const int ny = argc > 2 ? atoi(argv[2]) : 8192;
  const int rounds = argc > 3 ? atoi(argv[3]) : 3;
  const int iters = getenv("TUNE_ITERS") ? atoi(getenv("TUNE_ITERS")) : 20;
  const long cells = (long)nx * ny;
  const long max_pad = 1 << 20;
  const long ps_max = cells + max_pad + 2048L * ny / 9;  // also covers row-interleaved pads up to 2048 floats
  float *A, *B, *partials;
  unsigned char* mask;
  CK(hipMalloc(&A, ps_max * 9 * sizeof(float)));
  CK(hipMalloc(&B, (ps_max * 9 + (1 << 22)) * sizeof(float)));
  CK(hipMalloc(&mask, cells));
  CK(hipMalloc(&partials, 1 << 20));
  {  // walls every 1024 cells like the tiled obstacle map of the benchmark grid
    std::vector<unsigned char> m((size_t)cells, 0);
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++)
        m[(size_t)y * nx + x] = ((x % 1024) == 0 || (y % 1024) == 0 || (x % 1024) == 1023 || (y % 1024) == 1023 || (x % 1024) == 341);
    CK(hipMemcpy(mask, m.data(), (size_t)cells, hipMemcpyHostToDevice));
  }
  auto fill = [&](float* L, long ps) {
    hipLaunchKernelGGL(init_equilibrium, dim3((cells + 255) / 256), dim3(256), 0, 0, L, ps, (long)nx, nx, ny,
                       0.1f * 4.f / 9.f, 0.1f / 9.f, 0.1f / 36.f);
  };

  std::vector<Variant> vars = {
      {"exact loads nt       ", step_vec4<0, 0, true, 256, false>, 256},
      {"exact loads          ", step_vec4<0, 0, false, 256, false>, 256},
      {"fast  loads          ", step_vec4<1, 0, false, 256, false>, 256},
      {"copy  loads          ", step_vec4<2, 0, false, 256, false>, 256},
      {"exact loads b128     ", step_vec4<0, 0, false, 128, false>, 128},
      {"exact loads b64      ", step_vec4<0, 0, false, 64, false>, 64},
      {"copy  loads b64      ", step_vec4<2, 0, false, 64, false>, 64},
  };
  std::vector<long> pads = {0, 320, 1088, 8256};
  std::vector<int> occs = {0};  // 0 = no cap; k = at most k workgroups (4 waves each) per CU
  if (getenv("TUNE_OCC")) {
    occs.clear();
    for (char* t = strtok(getenv("TUNE_OCC"), ","); t; t = strtok(nullptr, ",")) occs.push_back(atoi(t));
  }
  const bool interleave = getenv("TUNE_LAYOUT") && !strcmp(getenv("TUNE_LAYOUT"), "rows");
  long boff = 0;  // extra offset (floats) of lattice B's base: decorrelates source and destination
  if (argc > 4 && !strcmp(argv[4], "sweep")) {
    // tools/tune_step nx ny rounds sweep <first> <step> <count> [boff]: one flavour, many pads
    pads.clear();
    const long first = atol(argv[5]), step = atol(argv[6]), count = atol(argv[7]);
    for (long i = 0; i < count; i++) pads.push_back(first + i * step);
    if (argc > 8) boff = atol(argv[8]);
    vars.resize(1);
  } else if (argc > 4) {
    pads.clear();
    for (int i = 4; i < argc; i++) pads.push_back(atol(argv[i]));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  printf("# grid %dx%d, %d iters per measurement, algorithmic 72 B/cell\n", nx, ny, iters);
  {
    const long n4 = cells * 9 / 4;
    for (int r = 0; r < rounds; r++) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(copy4, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4*)A, (float4*)B, n4);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("copy4 ceiling: %.4f ms  %.1f GB/s\n", ms / iters, 72.0 * cells / (ms / iters * 1e-3) / 1e9);
    }
  }
  for (long pad : pads) {
    const long ps = cells + pad;
    if (pad > (interleave ? 2048 : max_pad)) { printf("pad %ld too large, skipped\n", pad); continue; }
    if (interleave) {
      const long n = 9L * (nx + pad) * ny;
      hipLaunchKernelGGL(fill_const, dim3((n + 255) / 256), dim3(256), 0, 0, A, n, 0.0111f);
      hipLaunchKernelGGL(fill_const, dim3((n + 255) / 256), dim3(256), 0, 0, B + boff, n, 0.0111f);
    } else {
      fill(A, ps);
      fill(B + boff, ps);
    }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; r++) {
      for (auto& v : vars) for (int occ : occs) {
        const size_t dyn_lds = occ > 0 ? (size_t)(160 * 1024 / occ - 64) : 0;
        StepArgs a;
        memset(&a, 0, sizeof(a));
        a.mask = mask; a.plane_stride = interleave ? nx + pad : ps; a.pitch = nx;
        a.row_pitch = interleave ? 9L * (nx + pad) : nx; a.nx = nx; a.rows = ny; a.row_first = 0;
        a.row_stride = 1; a.n_rows = ny; a.omega = 1.85f; a.wrap = 1;
        a.a1 = 0.1f * 0.01f / 9.f; a.a2 = 0.1f * 0.01f / 36.f; a.accel_row = ny - 2; a.partials = partials;
        const int grid = (int)(((long)(nx / 4) * ny + v.block - 1) / v.block);
        float* L[2] = {A, B + boff};
        for (int i = 0; i < 2; i++) {
          a.src = L[i & 1]; a.dst = L[(i & 1) ^ 1]; a.reverse = v.snake ? (i & 1) : 0;
          hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.block), dyn_lds, 0, a);
        }
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) {
          a.src = L[i & 1]; a.dst = L[(i & 1) ^ 1]; a.reverse = v.snake ? (i & 1) : 0;
          hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.block), dyn_lds, 0, a);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / iters;
        char nm[64];
        snprintf(nm, sizeof(nm), "%s occ%d", v.name, occ);
        printf("pad %6ld  %-24s  %.4f ms  %7.1f GB/s  %6.0f MLUPS\n", pad, nm, per,
               72.0 * cells / (per * 1e-3) / 1e9, cells / (per * 1e-3) / 1e6);
      }
    }
  }
  return 0;
}
