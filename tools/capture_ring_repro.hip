// capture_ring_repro.hip -- minimal reproducer of the hipStreamEndCapture stack overflow that round 2 logged as a
// "hipGraphInstantiate SIGSEGV" (ROCm 7.2, libamdhip64.so.7; profiles/r03_graph_capture_defect.md).
//
//   hipcc --offload-arch=gfx950 tools/capture_ring_repro.hip -o /tmp/capture_ring_repro
//   /tmp/capture_ring_repro 2     -> "ended: no error, N nodes"
//   /tmp/capture_ring_repro 3     -> SIGSEGV inside hipStreamEndCapture (unbounded recursion)
//
// n side streams join a capture that begins on an origin stream, then wait for each other's events around a ring
// (stream i waits for its two neighbours, as the comm streams of n slabs with device-copy halos do).  The captured graph
// is acyclic -- every edge runs from an older node to a newer one -- but the runtime's bookkeeping is not:
// hipStreamWaitEvent re-parents the waiting stream to the event's stream and lists it in that stream's
// parallelCaptureStreams_ unless the event's stream's CURRENT parent is the waiter; with three or more streams the
// parent pointers rotate, two streams end up in each other's list, and hip::Stream::EndCapture(), which calls itself
// for every listed stream before clearing its own list, never returns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void touch(int* p) { if (p) *p += 1; }

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3;
  hipStream_t origin;
  CHECK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
  std::vector<hipStream_t> side(n);
  std::vector<hipEvent_t> done(n);
  hipEvent_t fork;
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int i = 0; i < n; i++) {
    CHECK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  }
  CHECK(hipStreamBeginCapture(origin, hipStreamCaptureModeRelaxed));
  CHECK(hipEventRecord(fork, origin));
  for (int i = 0; i < n; i++) {
    CHECK(hipStreamWaitEvent(side[i], fork, 0));
    hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, side[i], (int*)nullptr);
    CHECK(hipEventRecord(done[i], side[i]));
  }
  for (int round = 0; round < 2; round++)
    for (int i = 0; i < n; i++) {
      CHECK(hipStreamWaitEvent(side[i], done[(i + 1) % n], 0));
      CHECK(hipStreamWaitEvent(side[i], done[(i + n - 1) % n], 0));
      hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, side[i], (int*)nullptr);
      CHECK(hipEventRecord(done[i], side[i]));
    }
  for (int i = 0; i < n; i++) CHECK(hipStreamWaitEvent(origin, done[i], 0));
  printf("ending the capture of %d ring streams\n", n);
  fflush(stdout);
  hipGraph_t graph = nullptr;
  const hipError_t end = hipStreamEndCapture(origin, &graph);
  size_t nodes = 0;
  if (graph) (void)hipGraphGetNodes(graph, nullptr, &nodes);
  printf("ended: %s, %zu nodes\n", hipGetErrorString(end), nodes);
  return 0;
}
