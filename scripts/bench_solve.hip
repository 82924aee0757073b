// Microbenchmark of the scalar Gauss-Newton update (gn_step_core) on one lane, as the resident level kernels run it:
// state in LDS, sums in LDS, `reps` dependent calls; prints ns per call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Idensemonoslam_amd/csrc scripts/bench_solve.hip -o scripts/_solve/bench_solve
#include "../densemonoslam_amd/csrc/track.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
namespace dms {
__global__ __launch_bounds__(512) void k_solve_bench(GnLocal* g, const float* sums, int rgbSize, int sigma, SolveArgs q, KPre kp, int reps, int side,
                                                     long long* ticks, float* out) {
  __shared__ GnLocal s;
  __shared__ float s_sums[64];
  if (threadIdx.x < 64) s_sums[threadIdx.x] = sums[threadIdx.x];
  if (threadIdx.x == 0) s = *g;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    if (threadIdx.x == 0) {
      q.icp = 1;
      q.rgb = 1;
      gn_step_core(s, s_sums, s_sums + 32, rgbSize, sigma, q, &kp, side != 0 || r == reps - 1);
    }
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<float*>(&s)[threadIdx.x];
}

// The same call with FILL kilobytes of other straight-line code executed between two calls (as the pixel passes are in the
// level kernels): shows from which footprint on the solve's instructions have to be fetched again.
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))
template <int FILL>
__global__ __launch_bounds__(512) void k_solve_bench_fill(GnLocal* g, const float* sums, int rgbSize, int sigma, SolveArgs q, KPre kp, int reps,
                                                          long long* ticks, float* out) {
  __shared__ GnLocal s;
  __shared__ float s_sums[64];
  if (threadIdx.x < 64) s_sums[threadIdx.x] = sums[threadIdx.x];
  if (threadIdx.x == 0) s = *g;
  __syncthreads();
  long long acc = 0;
  float f = (float)threadIdx.x;
  for (int r = 0; r < reps; ++r) {
#define FILL4K R1024(asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));)
    if constexpr (FILL >= 4) { FILL4K }
    if constexpr (FILL >= 8) { FILL4K }
    if constexpr (FILL >= 12) { FILL4K }
    if constexpr (FILL >= 16) { FILL4K }
    if constexpr (FILL >= 20) { FILL4K }
    if constexpr (FILL >= 24) { FILL4K }
    if constexpr (FILL >= 28) { FILL4K }
    if constexpr (FILL >= 32) { FILL4K }
    if constexpr (FILL >= 40) { FILL4K FILL4K }
    if constexpr (FILL >= 48) { FILL4K FILL4K }
    if constexpr (FILL >= 56) { FILL4K FILL4K }
    if constexpr (FILL >= 64) { FILL4K FILL4K }
    __syncthreads();
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
      q.icp = 1;
      q.rgb = 1;
      gn_step_core(s, s_sums, s_sums + 32, rgbSize, sigma, q, &kp, r == reps - 1);
    }
    __syncthreads();
    acc += wall_clock64() - t0;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = acc;
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<float*>(&s)[threadIdx.x] + f;
}
}  // namespace dms
int main(int argc, char** argv) {
  using namespace dms;
  const int reps = argc > 1 ? atoi(argv[1]) : 200, blocks = argc > 2 ? atoi(argv[2]) : 1;
  GnLocal h{};
  for (int i = 0; i < 16; ++i) h.resultRt[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 9; ++i) h.Rprev[i] = h.Rprev_inv[i] = h.Rcurr[i] = (i % 4 == 0) ? 1.f : 0.f;
  float sums[64] = {0};
  // J^T J | J^T r of 500 random rows, upper triangle, (i, i..6) per row i
  double A[7][7] = {{0}};
  srand(5);
  for (int n = 0; n < 500; ++n) {
    double row[7];
    for (int k = 0; k < 7; ++k) row[k] = (rand() / (double)RAND_MAX - 0.5) * (k == 6 ? 0.01 : 1.0);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) A[i][j] += row[i] * row[j];
  }
  int sh = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      sums[sh] = (float)A[i][j];
      sums[32 + sh] = (float)(A[i][j] * 0.5);
      ++sh;
    }
  sums[27] = 0.02f;
  sums[28] = 500.f;
  SolveArgs q{};
  q.icpWeight = 10.f;
  q.fx = q.fy = 528.f;
  q.cx = 320.f;
  q.cy = 240.f;
  KPre kp;
  kp.fx = kp.fy = 528.0;
  kp.cx = 320.0;
  kp.cy = 240.0;
  kp.ifx = kp.ify = 1.0 / 528.0;
  GnLocal* d;
  float *ds, *dout;
  long long* dt;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&ds, sizeof(sums));
  hipMalloc(&dout, blocks * 64 * 4);
  hipMalloc(&dt, 8);
  hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
  hipMemcpy(ds, sums, sizeof(sums), hipMemcpyHostToDevice);
  for (int side = 0; side < 2; ++side)
    for (int it = 0; it < 3; ++it) {
      hipLaunchKernelGGL(k_solve_bench, dim3(blocks), dim3(512), 0, 0, d, ds, 400, 123456, q, kp, reps, side, dt, dout);
      long long t = 0;
      hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
      printf("side %d blocks %d: %.1f ns per call (%d dependent calls)\n", side, blocks, t * 10.0 / reps, reps);
    }
#define RUNFILL(F)                                                                                                            \
  for (int it = 0; it < 2; ++it) {                                                                                            \
    hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);                                                                       \
    hipLaunchKernelGGL(k_solve_bench_fill<F>, dim3(blocks), dim3(512), 0, 0, d, ds, 400, 123456, q, kp, 10, dt, dout);        \
    long long t = 0;                                                                                                          \
    hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);                                                                              \
    if (it) printf("fill %2d KB between calls: %.1f ns per call\n", F, t * 10.0 / 10);                                        \
  }
  RUNFILL(0) RUNFILL(8) RUNFILL(16) RUNFILL(24) RUNFILL(32) RUNFILL(40) RUNFILL(48) RUNFILL(56) RUNFILL(64)
  float o[64];
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("resultRt[3] as floats %g %g, krkinv region %g\n", o[6], o[7], o[40]);
  return 0;
}
