// hop_flavours.hip -- the cross-XCD hand-off of tools/hop_latency.hip again, over memory kinds (hipMalloc, fine-grained,
// uncached) and cache-policy bits of the store and of the load (aux of the raw-buffer builtins on gfx950: 1 = sc0,
// 2 = nt, 16 = sc1): is there a cheaper way across the fabric than "sc1 store, sc1 load"?
//   hipcc --offload-arch=gfx950 -O2 tools/hop_flavours.hip -o tools/hop_flavours && tools/hop_flavours
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef int vec4i __attribute__((ext_vector_type(4)));

template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(uint4* cells, int partner_xor, int rounds, long long* ticks, int* xcc_out, int active) {
  const int b = blockIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
  if (threadIdx.x == 0) xcc_out[b] = (int)xcc;
  if (b >= active) return;
  const int p = b ^ partner_xor;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(cells, 0, 1 << 20, 0x00020000);
  const unsigned mine = (unsigned)b * 1024u + threadIdx.x * 16u, theirs = (unsigned)p * 1024u + threadIdx.x * 16u;
  const bool leader = (b & partner_xor) == 0;
  long long t0 = 0;
  for (int r = 1; r <= rounds; r++) {
    if (r == 11) t0 = wall_clock64();
    const vec4i v = {r, r, r, r};
    if (leader) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, ST);
    bool seen = false;
    for (unsigned spins = 0; spins < (1u << 18) && !seen; spins++) {
      asm volatile("" ::: "memory");
      const vec4i g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)theirs, 0, LD);
      seen = __all(g.w == r);
    }
    if (!seen) {
      if (threadIdx.x == 0) ticks[b] = -1;
      return;
    }
    if (!leader) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, ST);
  }
  if (threadIdx.x == 0) ticks[b] = wall_clock64() - t0;
}

// the same game with an agent-scope atomic exchange of one 64-bit word {value, tag} per lane as the "store" (atomics are
// performed at the point of coherence), read by an agent-scope atomic load
__global__ __launch_bounds__(64) void pingpong_atomic(unsigned long long* cells, int partner_xor, int rounds, long long* ticks, int* xcc_out, int active) {
  const int b = blockIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
  if (threadIdx.x == 0) xcc_out[b] = (int)xcc;
  if (b >= active) return;
  const int p = b ^ partner_xor;
  unsigned long long* mine = cells + (size_t)b * 128 + threadIdx.x;
  unsigned long long* theirs = cells + (size_t)p * 128 + threadIdx.x;
  const bool leader = (b & partner_xor) == 0;
  long long t0 = 0;
  for (int r = 1; r <= rounds; r++) {
    if (r == 11) t0 = wall_clock64();
    const unsigned long long v = ((unsigned long long)r << 32) | (unsigned)r;
    if (leader) (void)__hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool seen = false;
    for (unsigned spins = 0; spins < (1u << 18) && !seen; spins++) {
      const unsigned long long g = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seen = __all((int)(g >> 32) == r);
    }
    if (!seen) {
      if (threadIdx.x == 0) ticks[b] = -1;
      return;
    }
    if (!leader) (void)__hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) ticks[b] = wall_clock64() - t0;
}

template <int ST, int LD>
void run(const char* kind, uint4* cells, long long* ticks, int* xcc, int px) {
  const int blocks = 256, rounds = 1010, active = 16;
  (void)hipMemset(cells, 0, 1 << 20);
  (void)hipMemset(ticks, 0, blocks * sizeof(long long));
  if (ST < 0) hipLaunchKernelGGL(pingpong_atomic, dim3(blocks), dim3(64), 0, 0, (unsigned long long*)cells, px, rounds, ticks, xcc, active);
  else hipLaunchKernelGGL((pingpong<(ST < 0 ? 0 : ST), LD>), dim3(blocks), dim3(64), 0, 0, cells, px, rounds, ticks, xcc, active);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(1); }
  std::vector<long long> t(blocks);
  std::vector<int> x(blocks);
  (void)hipMemcpy(t.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  (void)hipMemcpy(x.data(), xcc, blocks * sizeof(int), hipMemcpyDeviceToHost);
  std::vector<double> same, cross;
  for (int b = 0; b < active; b++) {
    const double hop_ns = t[b] < 0 ? 1e9 : (double)t[b] * 10.0 / (rounds - 10) / 2.0;
    (x[b] == x[b ^ px] ? same : cross).push_back(hop_ns);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-12s store aux %2d load aux %2d pair b^%d: same-XCD %2zu pairs %7.0f ns   cross-XCD %2zu pairs %7.0f ns\n", kind, ST, LD, px,
         same.size(), med(same), cross.size(), med(cross));
  fflush(stdout);
}

int main() {
  long long* ticks;
  int* xcc;
  (void)hipMalloc(&ticks, 256 * sizeof(long long));
  (void)hipMalloc(&xcc, 256 * sizeof(int));
  for (int kind = 0; kind < 3; kind++) {
    uint4* cells = nullptr;
    const char* name = kind == 0 ? "hipMalloc" : (kind == 1 ? "fine-grained" : "uncached");
    hipError_t e = kind == 0 ? hipMalloc(&cells, 1 << 20)
                             : hipExtMallocWithFlags((void**)&cells, 1 << 20, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", name, hipGetErrorString(e)); continue; }
    for (int px : {1, 8}) {
      run<-1, 16>(name, cells, ticks, xcc, px);
      run<16, 16>(name, cells, ticks, xcc, px);
      run<17, 17>(name, cells, ticks, xcc, px);
      run<16, 17>(name, cells, ticks, xcc, px);
      run<18, 16>(name, cells, ticks, xcc, px);
      run<19, 17>(name, cells, ticks, xcc, px);
      run<0, 16>(name, cells, ticks, xcc, px);
      run<0, 0>(name, cells, ticks, xcc, px);
      run<1, 1>(name, cells, ticks, xcc, px);
      run<2, 2>(name, cells, ticks, xcc, px);
    }
    (void)hipFree(cells);
  }
  return 0;
}
