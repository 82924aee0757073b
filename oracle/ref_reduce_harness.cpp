// Test infrastructure, compiled only by oracle/ref_build.sh into oracle/_ref/libref_reduce.so.
//
// extern "C" wrappers around the REFERENCE's tracking steps (elasticfusion/Core/src/Cuda/cudafuncs.cuh:70-117:
// icpStep, rgbStep, so3Step, computeRgbResidual), so that a Python script on the GPU box can run the reference's own
// kernels on host arrays and record what they return.  Everything that computes is the reference's; this file only moves
// bytes: host array -> the reference's DeviceArray2D (containers/device_array.hpp:194-206) -> the reference's function with
// the argument list RGBDOdometry.cpp:443-539 uses -> host.  Scratch buffers are sized as RGBDOdometry.cpp:44-49 sizes them.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "cudafuncs.cuh"

namespace {
template <class T>
void up(DeviceArray2D<T>& d, const void* host, int rows, int cols) {
  d.upload(host, (size_t)cols * sizeof(T), rows, cols);
}
mat33 m33(const float* p) {
  mat33 m;
  std::memcpy(&m.data[0], p, sizeof(mat33));
  return m;
}
float3 f3(const float* p) {
  float3 v = {p[0], p[1], p[2]};
  return v;
}
}  // namespace

extern "C" {

int ref_sizes(int* out4) {
  out4[0] = (int)sizeof(JtJJtrSE3);
  out4[1] = (int)sizeof(JtJJtrSO3);
  out4[2] = (int)sizeof(DataTerm);
  out4[3] = (int)sizeof(mat33);
  return 0;
}

// vmap / nmap: 3 planes stacked along rows (3*rows x cols floats), RGBDOdometry.cpp:97-101
int ref_icpStep(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                const float* tprev, const float* cam4, const float* vmap_g_prev, const float* nmap_g_prev, int rows, int cols,
                float distThres, float angleThres, int threads, int blocks, float* A36, float* b6, float* residual2) {
  DeviceArray2D<float> vc, nc, vp, np_;
  up(vc, vmap_curr, rows * 3, cols);
  up(nc, nmap_curr, rows * 3, cols);
  up(vp, vmap_g_prev, rows * 3, cols);
  up(np_, nmap_g_prev, rows * 3, cols);
  DeviceArray<JtJJtrSE3> sum, out;
  sum.create(1024);
  out.create(1);
  CameraModel intr(cam4[0], cam4[1], cam4[2], cam4[3]);
  icpStep(m33(Rcurr), f3(tcurr), vc, nc, m33(Rprev_inv), f3(tprev), intr, vp, np_, distThres, angleThres, sum, out, A36, b6,
          residual2, threads, blocks);
  return 0;
}

// corres_out: rows*cols DataTerm records as the kernel wrote them (linear index k = y*cols + x, reduce.cu:838)
int ref_computeRgbResidual(float minScale, const short* dIdx, const short* dIdy, const float* lastDepth, const float* nextDepth,
                           const unsigned char* lastImage, const unsigned char* nextImage, int rows, int cols,
                           float maxDepthDelta, const float* kt3, const float* krkinv9, int threads, int blocks,
                           void* corres_out, int* sigmaSum, int* count) {
  DeviceArray2D<short> dx, dy;
  DeviceArray2D<float> ld, nd;
  DeviceArray2D<unsigned char> li, ni;
  up(dx, dIdx, rows, cols);
  up(dy, dIdy, rows, cols);
  up(ld, lastDepth, rows, cols);
  up(nd, nextDepth, rows, cols);
  up(li, lastImage, rows, cols);
  up(ni, nextImage, rows, cols);
  DeviceArray2D<DataTerm> corres;
  corres.create(rows, cols);
  DeviceArray<int2> sumResidual;
  sumResidual.create(1024);
  int s = 0, c = 0;
  computeRgbResidual(minScale, dx, dy, ld, nd, li, ni, corres, sumResidual, maxDepthDelta, f3(kt3), m33(krkinv9), s, c, threads,
                     blocks);
  // the kernels index the (pitched) allocation LINEARLY (corresImg.data[k], k = y * cols + x: reduce.cu:838, :560), whatever
  // its pitch: the records are taken out the same way
  if (hipMemcpy(corres_out, corres.ptr(), (size_t)rows * cols * sizeof(DataTerm), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  *sigmaSum = s;
  *count = c;
  return 0;
}

// corres: rows*cols DataTerm records (as returned above); cloud: rows x cols float3
int ref_rgbStep(const void* corres_in, float sigma, const float* cloud3, float fx, float fy, const short* dIdx, const short* dIdy,
                float sobelScale, int rows, int cols, int threads, int blocks, float* A36, float* b6) {
  DeviceArray2D<DataTerm> corres;
  corres.create(rows, cols);
  if (hipMemcpy(corres.ptr(), corres_in, (size_t)rows * cols * sizeof(DataTerm), hipMemcpyHostToDevice) != hipSuccess) return -3;  // linear, as the kernel reads it
  DeviceArray2D<float3> cloud;
  up(cloud, cloud3, rows, cols);
  DeviceArray2D<short> dx, dy;
  up(dx, dIdx, rows, cols);
  up(dy, dIdy, rows, cols);
  DeviceArray<JtJJtrSE3> sum, out;
  sum.create(1024);
  out.create(1);
  rgbStep(corres, sigma, cloud, fx, fy, dx, dy, sobelScale, sum, out, A36, b6, threads, blocks);
  return 0;
}

int ref_so3Step(const unsigned char* lastImage, const unsigned char* nextImage, const float* imageBasis9, const float* kinv9,
                const float* krlr9, int rows, int cols, int threads, int blocks, float* A9, float* b3, float* residual2) {
  DeviceArray2D<unsigned char> li, ni;
  up(li, lastImage, rows, cols);
  up(ni, nextImage, rows, cols);
  DeviceArray<JtJJtrSO3> sum, out;
  sum.create(1024);
  out.create(1);
  so3Step(li, ni, m33(imageBasis9), m33(kinv9), m33(krlr9), sum, out, A9, b3, residual2, threads, blocks);
  return 0;
}

// ---- the same four steps behind the signatures of the restatement (oracle/orc.h: orc_so3Step, orc_computeRgbResidual,
// orc_icpStep, orc_rgbStep), so that oracle/orc_odometry.c can run its host loop around the reference's device code
// (orc_odometry_set_step_hooks; tests/golden/make_ref_tracker_golden.py).  The restatement's DataTerm keeps `valid` in an int
// where the reference has a bool and three bytes of padding: records are normalised on the way back.
void ref_hook_so3Step(const unsigned char* lastImage, const unsigned char* nextImage, const float* imageBasis, const float* kinv,
                      const float* krlr, int rows, int cols, float* A, float* b, float* residual) {
  ref_so3Step(lastImage, nextImage, imageBasis, kinv, krlr, rows, cols, 128, 64, A, b, residual);
}
void ref_hook_computeRgbResidual(float minScale, const short* dIdx, const short* dIdy, const float* lastDepth, const float* nextDepth,
                                 const unsigned char* lastImage, const unsigned char* nextImage, void* corres, float maxDepthDelta,
                                 const float* kt, const float* krkinv, int rows, int cols, int* sigmaSum, int* count) {
  ref_computeRgbResidual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, rows, cols, maxDepthDelta, kt, krkinv, 256, 336,
                         corres, sigmaSum, count);
  unsigned char* p = (unsigned char*)corres;
  for (size_t k = 0; k < (size_t)rows * cols; ++k, p += 16) {
    const int v = p[12] != 0;
    if (!v) std::memset(p, 0, 16);
    std::memcpy(p + 12, &v, 4);
  }
}
void ref_hook_icpStep(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                      const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                      float distThres, float angleThres, int rows, int cols, float* A, float* b, float* residual) {
  const float cam[4] = {fx, fy, cx, cy};
  ref_icpStep(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, rows, cols, distThres, angleThres, 128,
              112, A, b, residual);
}
void ref_hook_rgbStep(const void* corres, float sigma, const float* cloud3, float fx, float fy, const short* dIdx, const short* dIdy,
                      float sobelScale, int rows, int cols, float* A, float* b) {
  ref_rgbStep(corres, sigma, cloud3, fx, fy, dIdx, dIdy, sobelScale, rows, cols, 128, 112, A, b);
}

// Timing of the reference's Gauss-Newton inner loop as RGBDOdometry.cpp:425-541 runs it: per iteration computeRgbResidual,
// icpStep, rgbStep, each with its own launches, device synchronisation and download (that is how the reference's functions
// are written); inputs uploaded once, host clock around `iters` iterations (the host-side Eigen solve between them is not
// included: Eigen is not in this image).  so3_iters > 0 times so3Step alone the same way.  Returns microseconds per iteration.
int ref_time_iterations(float minScale, const short* dIdx, const short* dIdy, const float* lastDepth, const float* nextDepth,
                        const unsigned char* lastImage, const unsigned char* nextImage, const float* cloud3, const float* vmap_curr,
                        const float* nmap_curr, const float* vmap_g_prev, const float* nmap_g_prev, int rows, int cols,
                        const float* cam4, const float* kt3, const float* krkinv9, int iters, int so3_iters, double* us_gn,
                        double* us_so3) {
  DeviceArray2D<short> dx, dy;
  DeviceArray2D<float> ld, nd, vc, nc, vp, np_;
  DeviceArray2D<unsigned char> li, ni;
  DeviceArray2D<float3> cloud;
  up(dx, dIdx, rows, cols);
  up(dy, dIdy, rows, cols);
  up(ld, lastDepth, rows, cols);
  up(nd, nextDepth, rows, cols);
  up(li, lastImage, rows, cols);
  up(ni, nextImage, rows, cols);
  up(cloud, cloud3, rows, cols);
  up(vc, vmap_curr, rows * 3, cols);
  up(nc, nmap_curr, rows * 3, cols);
  up(vp, vmap_g_prev, rows * 3, cols);
  up(np_, nmap_g_prev, rows * 3, cols);
  DeviceArray2D<DataTerm> corres;
  corres.create(rows, cols);
  DeviceArray<int2> sumResidual;
  sumResidual.create(1024);
  DeviceArray<JtJJtrSE3> sum, out;
  sum.create(1024);
  out.create(1);
  DeviceArray<JtJJtrSO3> sum3, out3;
  sum3.create(1024);
  out3.create(1);
  CameraModel intr(cam4[0], cam4[1], cam4[2], cam4[3]);
  const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
  float A[36], b[6], res[2], A3[9], b3[3];
  int s = 0, c = 0;
  for (int warm = 0; warm < 2; ++warm) {
    const int n = warm ? iters : 3;
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      computeRgbResidual(minScale, dx, dy, ld, nd, li, ni, corres, sumResidual, 0.07f, f3(kt3), m33(krkinv9), s, c, 256, 336);
      icpStep(m33(I9), f3(z3), vc, nc, m33(I9), f3(z3), intr, vp, np_, 0.10f, 0.342f, sum, out, A, b, res, 128, 112);
      rgbStep(corres, sqrtf((float)(c > 0 ? c : 1)), cloud, intr.fx, intr.fy, dx, dy, 0.125f, sum, out, A, b, 128, 112);
    }
    hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    *us_gn = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
  }
  *us_so3 = 0.0;
  for (int warm = 0; warm < 2 && so3_iters > 0; ++warm) {
    const int n = warm ? so3_iters : 3;
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) so3Step(li, ni, m33(krkinv9), m33(I9), m33(krkinv9), sum3, out3, A3, b3, res, 128, 64);
    hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    *us_so3 = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
  }
  return 0;
}

}  // extern "C"
