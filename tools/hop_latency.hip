// hop_latency.hip -- what one in-launch hand-off between two workgroups costs on MI355X, by placement and store flavour.
// Pairs of workgroups play ping-pong with one 16-byte {v, v, v, tag} granule each way (the seam granule of
// lbm::resident_band); half a round trip is the "hop" that sits on the critical path of every resident timestep.
//   hipcc --offload-arch=gfx950 -O2 tools/hop_latency.hip -o tools/hop_latency && tools/hop_latency
// Workgroup b pairs with b ^ 1 (neighbouring dispatch slots: different XCDs under round-robin dealing) or with b ^ 8
// (same XCD); the XCC ids actually observed are reported.  Store: sc1 (write-through) or plain (stays in the XCD's L2);
// the load is sc1 (L1-bypassing) always.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef int vec4i __attribute__((ext_vector_type(4)));

template <bool PLAIN_STORE>
__global__ __launch_bounds__(64) void pingpong(uint4* cells, int partner_xor, int rounds, long long* ticks, int* xcc_out, int idle_blocks) {
  const int b = blockIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
  if (threadIdx.x == 0) xcc_out[b] = (int)xcc;
  if (b >= idle_blocks) return;  // only the first pairs play: an otherwise idle chip
  const int p = b ^ partner_xor;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(cells, 0, 1 << 20, 0x00020000);
  const unsigned mine = (unsigned)b * 1024u + threadIdx.x * 16u, theirs = (unsigned)p * 1024u + threadIdx.x * 16u;
  const bool leader = (b & partner_xor) == 0;
  long long t0 = 0;
  for (int r = 1; r <= rounds; r++) {
    if (r == 11) t0 = wall_clock64();
    if (leader) {
      const vec4i v = {r, r, r, r};
      if (PLAIN_STORE) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, 16);
    }
    bool seen = false;
    for (unsigned spins = 0; spins < (1u << 20) && !seen; spins++) {
      asm volatile("" ::: "memory");  // the load must be issued again in every spin
      const vec4i g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)theirs, 0, 16);
      seen = __all(g.w == r);
    }
    if (!seen) {  // never became visible (e.g. a plain store read from another XCD): report and leave, bounded
      if (threadIdx.x == 0) ticks[b] = -1;
      return;
    }
    if (!leader) {
      const vec4i v = {r, r, r, r};
      if (PLAIN_STORE) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)mine, 0, 16);
    }
  }
  if (threadIdx.x == 0) ticks[b] = wall_clock64() - t0;
}

// Tear test: are the four dwords of a naturally aligned 16-byte sc1 store ever seen apart by a 16-byte sc1 load of
// another workgroup?  Writers (even workgroups) store {r, r, r, r} with r counting up as fast as they can, no handshake;
// readers (odd workgroups, on another XCD for b ^ 1, on the same for b ^ 8) load the same granules as fast as they can
// and count every load whose four dwords are not all equal.  lbm::resident_band's seam granules {v, v, v, tag} rely on
// the answer being "never" (an aligned 16-byte access lies inside one 64-byte memory request); this is the evidence.
template <bool PLAIN_STORE>
__global__ __launch_bounds__(64) void tear_test(uint4* cells, int partner_xor, int iterations, unsigned long long* torn, unsigned long long* reads, unsigned long long* changes) {
  const int b = blockIdx.x;
  const bool writer = (b & partner_xor) == 0;
  const int w = writer ? b : (b ^ partner_xor);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(cells, 0, 1 << 20, 0x00020000);
  const unsigned at = (unsigned)w * 1024u + threadIdx.x * 16u;
  unsigned long long bad = 0, seen = 0, changed = 0;
  int prev = 0;
  for (int r = 1; r <= iterations; r++) {
    if (writer) {
      const vec4i v = {r, r, r, r};
      if (PLAIN_STORE) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)at, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)at, 0, 16);
    } else {
      asm volatile("" ::: "memory");
      const vec4i g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)at, 0, 16);
      seen++;
      if (!(g.x == g.y && g.y == g.z && g.z == g.w)) bad++;
      if (g.w != prev) { changed++; prev = g.w; }   // how many different states of the granule this lane saw
    }
  }
  if (!writer) { atomicAdd(torn, bad); atomicAdd(reads, seen); atomicAdd(changes, changed); }
}

int main() {
  const int blocks = 256, rounds = 2010;
  uint4* cells;
  long long* ticks;
  int* xcc;
  (void)hipMalloc(&cells, 1 << 20);
  (void)hipMalloc(&ticks, blocks * sizeof(long long));
  (void)hipMalloc(&xcc, blocks * sizeof(int));
  for (int plain = 0; plain < 2; plain++)
    for (int px : {1, 8})
      for (int active : {16, 256}) {
        (void)hipMemset(cells, 0, 1 << 20);
        (void)hipMemset(ticks, 0, blocks * sizeof(long long));
        if (plain) hipLaunchKernelGGL(pingpong<true>, dim3(blocks), dim3(64), 0, 0, cells, px, rounds, ticks, xcc, active);
        else hipLaunchKernelGGL(pingpong<false>, dim3(blocks), dim3(64), 0, 0, cells, px, rounds, ticks, xcc, active);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<long long> t(blocks);
        std::vector<int> x(blocks);
        (void)hipMemcpy(t.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipMemcpy(x.data(), xcc, blocks * sizeof(int), hipMemcpyDeviceToHost);
        std::vector<double> same, cross;
        for (int b = 0; b < active; b++) {
          const double hop_ns = t[b] < 0 ? 1e9 : (double)t[b] * 10.0 / (rounds - 10) / 2.0;  // 100 MHz ticks; two hops per round; 1e9 = never seen
          (x[b] == x[b ^ px] ? same : cross).push_back(hop_ns);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("store %-5s pair b^%d, %3d workgroups playing: same-XCD pairs %3zu: hop %6.0f ns   cross-XCD pairs %3zu: hop %6.0f ns\n",
               plain ? "plain" : "sc1", px, active, same.size(), med(same), cross.size(), med(cross));
      }
  unsigned long long* counters;
  (void)hipMalloc(&counters, 3 * sizeof(unsigned long long));
  for (int plain = 0; plain < 2; plain++)
    for (int px : {1, 8}) {
      if (plain && px == 1) continue;  // plain stores are never seen across XCDs at all
      (void)hipMemset(cells, 0, 1 << 20);
      (void)hipMemset(counters, 0, 3 * sizeof(unsigned long long));
      if (plain) hipLaunchKernelGGL(tear_test<true>, dim3(blocks), dim3(64), 0, 0, cells, px, 400000, counters, counters + 1, counters + 2);
      else hipLaunchKernelGGL(tear_test<false>, dim3(blocks), dim3(64), 0, 0, cells, px, 400000, counters, counters + 1, counters + 2);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      unsigned long long h[3];
      (void)hipMemcpy(h, counters, sizeof(h), hipMemcpyDeviceToHost);
      printf("tear test, store %-5s pair b^%d (%s XCD): %llu 16-byte loads against concurrent stores, %llu of them saw a new state, %llu torn\n",
             plain ? "plain" : "sc1", px, px == 1 ? "other" : "same", h[1], h[2], h[0]);
    }
  return 0;
}
