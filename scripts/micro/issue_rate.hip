// What ONE wave can issue on gfx950 (the tracker's scalar section is one lane of one wave while the block's other waves wait at a
// barrier): cycles per vector instruction for dependent and independent chains, one lane and all lanes, one block shape and another.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math scripts/micro/issue_rate.hip -o scripts/micro/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k_chain(double* out, long long* clk, double a0, double b0, int n) {
  if (threadIdx.x == 0) {
    double a = a0, b = b0, c = a0 * 0.5, d = b0 * 0.25, e = a0 + 1, f = b0 + 2, g = a0 + 3, h = b0 + 4;
    float fa = (float)a0, fb = (float)b0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      if (MODE == 0) { a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); }
      if (MODE == 1) { a = __builtin_fma(a, b, 1e-9); c = __builtin_fma(c, b, 1e-9); d = __builtin_fma(d, b, 1e-9); e = __builtin_fma(e, b, 1e-9); f = __builtin_fma(f, b, 1e-9); g = __builtin_fma(g, b, 1e-9); h = __builtin_fma(h, b, 1e-9); a = __builtin_fma(a, b, 1e-9); }
      if (MODE == 2) { a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a = 1.0 / a; a += 1e-12; }
      if (MODE == 3) { fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); fa = __builtin_fmaf(fa, fb, 1e-9f); }
      if (MODE == 4) { a = 1.0 / a; c = 1.0 / c; d = 1.0 / d; e = 1.0 / e; a += 1e-12; c += 1e-12; d += 1e-12; e += 1e-12; }
    }
    long long t1 = clock64();
    out[0] = a + c + d + e + f + g + h + fa;
    clk[MODE] = t1 - t0;
  }
  __syncthreads();
}

// dependent fp64 fma chain / integer multiply-add chain with LANES lanes of wave 0 active in a block of blockDim.x threads
template <int LANES>
__global__ void k_lanes(double* out, long long* clk, double a0, double b0, int n, int slot) {
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  int x = threadIdx.x;
  if (threadIdx.x < LANES) {
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
      a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9);
      a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9); a = __builtin_fma(a, b, 1e-9);
    }
    long long t1 = clock64();
    for (int i = 0; i < n; ++i) { x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; x = x * 3 + 1; }
    long long t2 = clock64();
    if (threadIdx.x == 0) { clk[8 + slot * 2] = t1 - t0; clk[8 + slot * 2 + 1] = t2 - t1; }
    out[8 + threadIdx.x] = a + x;
  }
  __syncthreads();
}

int main() {
  double* o; long long* c;
  if (hipMalloc(&o, 8192) != hipSuccess || hipMalloc(&c, 512) != hipSuccess || hipMemset(c, 0, 512) != hipSuccess) return 1;
  const int n = 4000;
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n);
    hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n);
    hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n);
    hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n);
    hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n);
    hipLaunchKernelGGL(k_lanes<1>, dim3(1), dim3(64), 0, 0, o, c, 0.999, 1.0000001, n, 0);
    hipLaunchKernelGGL(k_lanes<64>, dim3(1), dim3(64), 0, 0, o, c, 0.999, 1.0000001, n, 1);
    hipLaunchKernelGGL(k_lanes<1>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n, 2);
    hipLaunchKernelGGL(k_lanes<64>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n, 3);
    hipLaunchKernelGGL(k_lanes<512>, dim3(1), dim3(512), 0, 0, o, c, 0.999, 1.0000001, n, 4);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    long long h[64];
    if (hipMemcpy(h, c, 512, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    printf("one lane of a 512-thread block, cycles per instruction: dependent fma_f64 %.1f | 7 independent fma_f64 chains %.1f | dependent 1/x (f64) %.1f per division | "
           "dependent fma_f32 %.1f | 4 independent 1/x %.1f per division\n",
           h[0] / (8.0 * n), h[1] / (8.0 * n), h[2] / (8.0 * n), h[3] / (8.0 * n), h[4] / (4.0 * n));
    const char* nm[] = {"1 lane, 64-thread block", "64 lanes, 64-thread block", "1 lane, 512-thread block", "64 lanes, 512-thread block", "all 512 threads (2 waves per SIMD)"};
    for (int s = 0; s < 5; ++s) printf("  %-36s dependent fma_f64 %.2f cycles per instruction   int mad %.2f\n", nm[s], h[8 + s * 2] / (8.0 * n), h[8 + s * 2 + 1] / (8.0 * n));
  }
  return 0;
}
