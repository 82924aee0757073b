// Stand-alone timing of the scalar section (gn_scalar.hpp: gn_step_combined) as the level kernels run it: one lane of a
// 512-thread block, state and the combined system in LDS, 200 dependent calls.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -mllvm -disable-machine-licm -DVAR=0 scripts/bench_solve_parts.hip -o bs0
// Round 3 on MI355X: 1.28 us per call.  With parts replaced in a scratch copy of gn_scalar.hpp (VAR hooks, not committed): no
// LDL^T at all 0.92 (the factorisation + substitutions are 0.36 us), exp map replaced by a stub 1.09 (0.19 us), hardware
// reciprocal instead of the six IEEE divisions 1.34 (no gain: not latency-bound on the divisions), no projection parameters
// 1.26.  ~800 executed instructions at ~4 cycles each: the section is bound by its instruction count on one wave.
#include "../densemonoslam_amd/csrc/gn_scalar.hpp"
#include <cstdio>
#include <cstdlib>
namespace dms {
__global__ __launch_bounds__(512) void k_bench(sc::GnLocal* g, const double* comb, sc::SolveArgs q, sc::KPre kp, int reps, long long* ticks, float* out) {
  __shared__ sc::GnLocal s;
  __shared__ double s_comb[28];
  if (threadIdx.x < 27) s_comb[threadIdx.x] = comb[threadIdx.x];
  if (threadIdx.x == 0) s = *g;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    if (threadIdx.x == 0) {
      q.icp = 1; q.rgb = 1;
      sc::gn_step_combined(s, s_comb, 0.02f, 500.f, 400, 123456, q, kp, r == reps - 1);
    }
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  if (threadIdx.x < 64) out[threadIdx.x] = reinterpret_cast<float*>(&s)[threadIdx.x];
}
}
int main() {
  using namespace dms;
  sc::GnLocal h{};
  for (int i = 0; i < 16; ++i) h.resultRt[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 9; ++i) h.Rprev[i] = h.Rprev_inv[i] = h.Rcurr[i] = (i % 4 == 0) ? 1.f : 0.f;
  double A[7][7] = {{0}}, comb[27];
  srand(5);
  for (int n = 0; n < 500; ++n) {
    double row[7];
    for (int k = 0; k < 7; ++k) row[k] = (rand() / (double)RAND_MAX - 0.5) * (k == 6 ? 0.01 : 1.0);
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) A[i][j] += row[i] * row[j];
  }
  int sh = 0;
  for (int i = 0; i < 6; ++i) for (int j = i; j < 7; ++j) comb[sh++] = A[i][j] * 100.5;
  sc::SolveArgs q{}; q.icpWeight = 10.f; q.fx = q.fy = 528.f; q.cx = 320.f; q.cy = 240.f;
  sc::KPre kp = sc::kpre_of(528.f, 528.f, 320.f, 240.f, 0);
  sc::GnLocal* d; double* dc; float* dout; long long* dt;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dc, sizeof(comb)); hipMalloc(&dout, 256); hipMalloc(&dt, 8);
  hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(dc, comb, sizeof(comb), hipMemcpyHostToDevice);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(512), 0, 0, d, dc, q, kp, 200, dt, dout);
    long long t = 0; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    printf("%.1f ns per call\n", t * 10.0 / 200);
  }
  float o[64]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); printf("check %g %g\n", o[6], o[40]);
  return 0;
}
