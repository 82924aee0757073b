// What does a boundary between two DEPENDENT dispatches cost on one stream, and does a captured hipGraph replay it cheaper?
// Chains of n kernels (tiny: one wave; wide: 1 200 blocks of 256 threads doing a few loads), launched (a) one by one, (b) as an
// instantiated graph.  hipcc --offload-arch=gfx950 -O3 scripts/micro/graph_bench.hip -o scripts/micro/graph_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_spin(long long ticks) {  // holds the stream while the host enqueues what is measured
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_tiny(unsigned* p) {
  if (threadIdx.x == 0) p[0] += 1;
}
__global__ __launch_bounds__(256) void k_wide(const float* __restrict__ a, float* __restrict__ b, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i] * 1.0001f + 1.f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int n = 307200, chain = 20, reps = 200;
  unsigned* cnt; float *a, *b;
  CK(hipMalloc(&cnt, 256)); CK(hipMemset(cnt, 0, 256));
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int wide = 0; wide < 2; ++wide) {
    auto enqueue = [&]() {
      for (int k = 0; k < chain; ++k) {
        if (wide) hipLaunchKernelGGL(k_wide, dim3((n + 255) / 256), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n);
        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, cnt);
      }
    };
    // (a) plain launches
    for (int r = 0; r < 20; ++r) enqueue();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) enqueue();
    CK(hipStreamSynchronize(s));
    double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    // (b) captured graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    // (c) the same two with the host out of the picture: everything is enqueued behind a 20 ms spinner, device time by events
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_plain = 0.f, ms_graph = 0.f;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 2000000LL);
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 50; ++r) enqueue();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_plain, e0, e1));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 2000000LL);
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("%s kernels, chain of %d: %.2f us per kernel launched one by one, %.2f us as a graph; queued behind a spinner (device side only): %.2f / %.2f us\n",
           wide ? "wide (1200 x 256)" : "tiny (1 wave)", chain, us_plain, us_graph, ms_plain * 1000.0 / (50 * chain), ms_graph * 1000.0 / (50 * chain));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
