// Can the tracker's resident kernels be COOPERATIVE launches (hipLaunchCooperativeKernel: all blocks co-resident by contract)?
// (1) what a cooperative launch costs beside a plain one, back to back on one stream; (2) whether it still overlaps a kernel on
// another stream (the frame step's prep stream runs beside the tracker) or waits for an idle device.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/coop_bench.hip -o scripts/micro/coop_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void k_spin(long long ticks, long long* stamp) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}
__global__ __launch_bounds__(512) void k_resident(long long* stamp, unsigned* counter, int rounds) {
  // a grid-wide arrive-and-wait per round, the way the tracker's kernels do it (agent-scope atomics, no cooperative-groups call)
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[0] = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * gridDim.x) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  long long* stamps; unsigned* counter;
  CK(hipMalloc(&stamps, 64)); CK(hipMalloc(&counter, 256));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int coop = 0; CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
  printf("cooperative launch supported: %d\n", coop);
  const int blocks = 200, rounds = 20, reps = 200;
  long long* st = nullptr; int rr = rounds;
  void* args[] = {&st, &counter, &rr};
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 3; ++w) {
      CK(hipMemsetAsync(counter, 0, 4, sb));
      if (mode) CK(hipLaunchCooperativeKernel((const void*)k_resident, dim3(blocks), dim3(512), args, 0, sb));
      else hipLaunchKernelGGL(k_resident, dim3(blocks), dim3(512), 0, sb, st, counter, rounds);
    }
    CK(hipStreamSynchronize(sb));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(counter, 0, 4, sb));
      if (mode) CK(hipLaunchCooperativeKernel((const void*)k_resident, dim3(blocks), dim3(512), args, 0, sb));
      else hipLaunchKernelGGL(k_resident, dim3(blocks), dim3(512), 0, sb, st, counter, rounds);
    }
    CK(hipStreamSynchronize(sb));
    printf("%s launch of a 200 x 512 resident kernel with %d grid-wide rounds: %.2f us per launch (with its 4-byte memset)\n", mode ? "cooperative" : "plain", rounds,
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps);
  }
  // device side only: 50 launches queued behind a 20 ms spinner, timed by events on the stream
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sb, 2000000LL, (long long*)nullptr);
    CK(hipEventRecord(e0, sb));
    for (int r = 0; r < 50; ++r) {
      CK(hipMemsetAsync(counter, 0, 4, sb));
      if (mode) CK(hipLaunchCooperativeKernel((const void*)k_resident, dim3(blocks), dim3(512), args, 0, sb));
      else hipLaunchKernelGGL(k_resident, dim3(blocks), dim3(512), 0, sb, st, counter, rounds);
    }
    CK(hipEventRecord(e1, sb));
    CK(hipStreamSynchronize(sb));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s, queued behind a spinner (device side only): %.2f us per launch\n", mode ? "cooperative" : "plain", ms * 1000.0 / 50);
  }
  // overlap: 40 fat blocks spin for 300 us on stream a; 20 us later the resident kernel is launched on stream b
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(stamps, 0, 64)); CK(hipMemset(counter, 0, 4));
    long long* sa_st = stamps; long long* sb_st = stamps + 2;
    hipLaunchKernelGGL(k_spin, dim3(40), dim3(1024), 0, sa, 30000LL, sa_st);
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    void* a2[] = {&sb_st, &counter, &rr};
    if (mode) CK(hipLaunchCooperativeKernel((const void*)k_resident, dim3(blocks), dim3(512), a2, 0, sb));
    else hipLaunchKernelGGL(k_resident, dim3(blocks), dim3(512), 0, sb, sb_st, counter, rounds);
    CK(hipDeviceSynchronize());
    long long h[4]; CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
    printf("%s: spinner ran %.1f us; the resident kernel started %.1f us after the spinner's start and ran %.1f us -> %s\n", mode ? "cooperative" : "plain",
           (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, (h[3] - h[2]) * 0.01, (h[2] < h[1]) ? "OVERLAPS the other stream's kernel" : "WAITED for the other stream's kernel");
  }
  return 0;
}
