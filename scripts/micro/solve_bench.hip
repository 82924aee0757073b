// Micro-benchmark of the tracker's scalar Gauss-Newton solve (gn_scalar.hpp gn_step_combined) as the resident kernel runs it:
// one lane of one wave, state in LDS.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I densemonoslam_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "common.hpp"
#include "smallmath.hpp"
#include "gn_scalar.hpp"
using namespace dms;


// instrumented copy of sc::gn_step_combined: core-clock stamps between its sections
__device__ __forceinline__ void gn_step_stamped(sc::GnLocal& L, const double* comb, const sc::SolveArgs& q, const sc::KPre& kpre, long long* acc) {
  using namespace sc;
  long long t = clock64(), u;
#define STAMP(i) u = clock64(); acc[i] += u - t; t = u;
  double A[36], b[6], x[6];
  {
    int shift = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 7; ++j) {
        const double v = comb[shift++];
        if (j == 6) b[i] = v; else A[j * 6 + i] = A[i * 6 + j] = v;
      }
  }
  STAMP(0)
  if (!ldlt_spd<6>(A, b, x)) ldlt_pivoted<double, 6>(A, b, x, 1.0 / 1.7976931348623157e308);
  STAMP(1)
  const double rvec[3] = {x[3], x[4], x[5]};
  double R[9];
  rodrigues(rvec, R);
  STAMP(2)
  double nr[16];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double v = fmad(R[i * 3 + 2], L.resultRt[2 * 4 + j], fmad(R[i * 3 + 1], L.resultRt[1 * 4 + j], R[i * 3 + 0] * L.resultRt[0 * 4 + j]));
      if (j == 3) v += x[i];
      nr[i * 4 + j] = v;
    }
  }
  nr[12] = 0.0; nr[13] = 0.0; nr[14] = 0.0; nr[15] = 1.0;
  STAMP(3)
  float Ro[9], to[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Ro[i * 3 + j] = (float)nr[i * 4 + j];
    to[i] = (float)nr[i * 4 + 3];
  }
  float RoT[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) RoT[i * 3 + j] = Ro[j * 3 + i];
  float ti[3];
  mul3vf(RoT, to, ti);
  ti[0] = -ti[0]; ti[1] = -ti[1]; ti[2] = -ti[2];
  float Rprev[9], Rc[9], tc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rprev[i] = L.Rprev[i];
  mul3f(Rprev, RoT, Rc);
  mul3vf(Rprev, ti, tc);
  L.iters_run += 1;
#pragma unroll
  for (int i = 0; i < 16; ++i) L.resultRt[i] = nr[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) L.Rcurr[i] = Rc[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) L.tcurr[i] = tc[i] + L.tprev[i];
  STAMP(4)
  gn_params(nr, kpre, L.krkinv, L.kt);
  STAMP(5)
}

__global__ void k_solve(const double* comb_in, int iters, double* out, long long* clocks) {
  __shared__ sc::GnLocal s;
  __shared__ double s_comb[28];
  __shared__ sc::KPre s_k[2];
  if (threadIdx.x == 0) {
    memset(&s, 0, sizeof(s));
    for (int i = 0; i < 16; ++i) s.resultRt[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 9; ++i) s.Rprev[i] = s.Rprev_inv[i] = s.Rcurr[i] = (i % 4 == 0) ? 1.f : 0.f;
    for (int i = 0; i < 27; ++i) s_comb[i] = comb_in[i];
    s_k[0] = sc::kpre_of(528.f, 528.f, 320.f, 240.f, 0);
    s_k[1] = sc::kpre_of(528.f, 528.f, 320.f, 240.f, 1);
  }
  __syncthreads();
  long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x == 0) {
      sc::SolveArgs q;
      q.icp = 1; q.rgb = 1; q.rgbOnly = 0; q.icpWeight = 10.f; q.level = 0; q.first_iter = it == 0; q.next_level = 0; q.level_below = 0;
      q.fx = 528.f; q.fy = 528.f; q.cx = 320.f; q.cy = 240.f;
      if (clocks[15]) gn_step_stamped(s, s_comb, q, s_k[0], acc); else sc::gn_step_combined(s, s_comb, 1.0f, 1000.f, 1000, 5, q, s_k[0], false);
      s_comb[6] += 1e-9;  // keep the iterations from being hoisted
    }
    __syncthreads();
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; ++i) out[i] = s.resultRt[i];
    clocks[0] = t1 - t0;
    for (int i = 0; i < 6; ++i) clocks[1 + i] = acc[i];
  }
}

int main() {
  double comb[27];
  // a well-conditioned SPD system: diagonally dominant A, small b
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) comb[k++] = j == 6 ? 1e-3 * (i + 1) : (i == j ? 100.0 + i : 0.5 / (1 + i + j));
  double *d_comb, *d_out; long long* d_clk;
  hipMalloc(&d_comb, sizeof(comb)); hipMalloc(&d_out, 16 * 8); hipMalloc(&d_clk, 16 * 8); hipMemset(d_clk, 0, 128);
  hipMemcpy(d_comb, comb, sizeof(comb), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 4; ++rep) {
    long long flag = rep >= 2; hipMemcpy(d_clk + 15, &flag, 8, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(512), 0, 0, d_comb, iters, d_out, d_clk);
    hipDeviceSynchronize();
    long long clk; double out[16]; long long c[8];
    hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost); hipMemcpy(c, d_clk, 64, hipMemcpyDeviceToHost);
    if (flag) printf("  core clocks per iteration: load %lld  ldlt %lld  rodrigues %lld  nr %lld  pose(float)+stores %lld  gn_params %lld\n", c[1] / iters, c[2] / iters, c[3] / iters, c[4] / iters, c[5] / iters, c[6] / iters); hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    printf("solve: %.3f us per iteration (wall_clock64 at 100 MHz), resultRt[3] = %.17g\n", clk * 0.01 / iters, out[3]);
  }
  return 0;
}
