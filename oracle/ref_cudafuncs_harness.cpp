// Test infrastructure, compiled only by oracle/ref_build.sh into oracle/_ref/libref_cudafuncs.so.
//
// extern "C" wrappers around the REFERENCE's pyramid / preparation operators and NID scores
// (elasticfusion/Core/src/Cuda/cudafuncs.cuh:119-184 -> cudafuncs.cu:57-757, 1086-1157, 1513-1916), so that a Python script on
// the GPU box can run the reference's own kernels on host arrays and record what they return.  Everything that computes is
// the reference's; this file only moves bytes: host array -> the reference's DeviceArray / DeviceArray2D
// (containers/device_array.hpp) -> the reference's function with the argument list its callers use
// (Utils/RGBDOdometry.cpp:118-292,412, KeyFrame.h:46-219, MutualInformation.cpp:154-213) -> host.
//
// Several of the reference's kernels leave parts of their output unwritten (createVMap: the y / z planes of a pixel without
// depth, cudafuncs.cu:125; tranformMaps: the y / z of a NaN vertex, :215-221; resizeMap: the y / z of a NaN result, :466).
// The reference's create() is a no-op for an allocation that already has the right size (device_memory.cpp:218-243), so the
// wrappers hand over destination arrays that hold ZEROS: the unwritten parts then read back as 0 on every run, the same
// convention the restatement's numpy wrappers use (oracle/orc.py createVMap / resizeMap start from zeros).
// tranformMaps is called in place, as every caller of the reference does (RGBDOdometry.cpp:203, KeyFrame.h:56,129,158,218).
//
// NOT wrapped: imageBGRToIntensity (cudafuncs.cu:641-669) - a legacy texture-reference sampler; gfx950 has no image
// instructions, HIP marks tex2D unavailable there, and ref_build.sh removes exactly those lines before compiling.
#include <cstdint>
#include <cstring>

#include "cudafuncs.cuh"

namespace {
template <class T>
void up(DeviceArray2D<T>& d, const void* host, int rows, int cols) {
  d.upload(host, (size_t)cols * sizeof(T), rows, cols);
}
template <class T>
void zeros(DeviceArray2D<T>& d, int rows, int cols) {
  d.create(rows, cols);
  hipMemset2D(d.ptr(), d.step(), 0, (size_t)cols * sizeof(T), rows);
}
template <class T>
int down(const DeviceArray2D<T>& d, void* host, int rows, int cols) {
  if (d.rows() != rows || d.cols() != cols) return -2;  // the reference sized its output differently from the caller's idea
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  d.download(host, (size_t)cols * sizeof(T));
  return 0;
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

int ref_cf_pyrDown(const uint16_t* src, int rows, int cols, uint16_t* dst) {
  DeviceArray2D<unsigned short> s, d;
  up(s, src, rows, cols);
  pyrDown(s, d, 0);
  return down(d, dst, rows / 2, cols / 2);
}

// vmap: 3 planes stacked along rows (3*rows x cols)
int ref_cf_createVMap(float fx, float fy, float cx, float cy, const uint16_t* depth, int rows, int cols, float* vmap, float cutoff) {
  DeviceArray2D<unsigned short> d;
  up(d, depth, rows, cols);
  DeviceArray2D<float> v;
  zeros(v, rows * 3, cols);
  CameraModel intr(fx, fy, cx, cy);
  createVMap(intr, d, v, cutoff, 0);
  return down(v, vmap, rows * 3, cols);
}

int ref_cf_createNMap(const float* vmap, int rows, int cols, float* nmap) {
  DeviceArray2D<float> v, n;
  up(v, vmap, rows * 3, cols);
  zeros(n, rows * 3, cols);
  createNMap(v, n, 0);
  return down(n, nmap, rows * 3, cols);
}

// in place, as the reference's callers use it; nmap may be NULL (the vertex-only overload, cudafuncs.cu:296)
int ref_cf_tranformMaps(float* vmap, float* nmap, int rows, int cols, const float* R9, const float* t3) {
  DeviceArray2D<float> v, n;
  up(v, vmap, rows * 3, cols);
  if (nmap) {
    up(n, nmap, rows * 3, cols);
    tranformMaps(v, n, m33(R9), f3(t3), v, n);
    int rc = down(n, nmap, rows * 3, cols);
    if (rc) return rc;
  } else {
    tranformMaps(v, m33(R9), f3(t3), v);
  }
  return down(v, vmap, rows * 3, cols);
}

// v4 / n4: rows x cols x 4 floats (RGBA32F texels copied linearly, RGBDOdometry.cpp:148-158); n4 may be NULL (:396)
int ref_cf_copyMaps(const float* v4, const float* n4, int rows, int cols, float* vd, float* nd) {
  DeviceArray<float> vs, ns;
  vs.upload(v4, (size_t)rows * cols * 4);
  DeviceArray2D<float> v, n;
  zeros(v, rows * 3, cols);
  if (n4) {
    ns.upload(n4, (size_t)rows * cols * 4);
    zeros(n, rows * 3, cols);
    copyMaps(vs, ns, v, n, 0);
    int rc = down(n, nd, rows * 3, cols);
    if (rc) return rc;
  } else {
    copyMaps(vs, v);
  }
  return down(v, vd, rows * 3, cols);
}

int ref_cf_resizeMap(const float* m, int rows, int cols, float* out, int normalize) {
  DeviceArray2D<float> s, d;
  up(s, m, rows * 3, cols);
  zeros(d, (rows / 2) * 3, cols / 2);
  if (normalize)
    resizeNMap(s, d, 0);
  else
    resizeVMap(s, d, 0);
  return down(d, out, (rows / 2) * 3, cols / 2);
}

int ref_cf_pyrDownGaussF(const float* src, int rows, int cols, float* dst) {
  DeviceArray2D<float> s, d;
  up(s, src, rows, cols);
  pyrDownGaussF(s, d);
  return down(d, dst, rows / 2, cols / 2);
}

int ref_cf_pyrDownUcharGauss(const unsigned char* src, int rows, int cols, unsigned char* dst) {
  DeviceArray2D<unsigned char> s, d;
  up(s, src, rows, cols);
  pyrDownUcharGauss(s, d, 0);
  return down(d, dst, rows / 2, cols / 2);
}

// the DeviceArray<float> overload (RGBA32F texels, cudafuncs.cu:610; RGBDOdometry.cpp:213)
int ref_cf_verticesToDepth(const float* v4, int rows, int cols, float* dst, float cutoff) {
  DeviceArray<float> vs;
  vs.upload(v4, (size_t)rows * cols * 4);
  DeviceArray2D<float> d;
  zeros(d, rows, cols);
  verticesToDepth(vs, d, cutoff);
  return down(d, dst, rows, cols);
}

// the DeviceArray2D<float> overload (3-plane vertex map, cudafuncs.cu:632)
int ref_cf_verticesToDepth2D(const float* vmap, int rows, int cols, float* dst, float cutoff) {
  DeviceArray2D<float> v, d;
  up(v, vmap, rows * 3, cols);
  zeros(d, rows, cols);
  verticesToDepth(v, d, cutoff);
  return down(d, dst, rows, cols);
}

int ref_cf_computeDerivativeImages(const unsigned char* img, int rows, int cols, short* dx, short* dy) {
  DeviceArray2D<unsigned char> s;
  up(s, img, rows, cols);
  DeviceArray2D<short> gx, gy;
  zeros(gx, rows, cols);  // the caller owns them in the reference too (RGBDOdometry.cpp:78-79)
  zeros(gy, rows, cols);
  computeDerivativeImages(s, gx, gy);
  int rc = down(gx, dx, rows, cols);
  return rc ? rc : down(gy, dy, rows, cols);
}

// cloud: rows x cols float3
int ref_cf_projectToPointCloud(const float* depth, int rows, int cols, float* cloud, float fx, float fy, float cx, float cy, int level) {
  DeviceArray2D<float> d;
  up(d, depth, rows, cols);
  DeviceArray2D<float3> c;
  zeros(c, rows, cols);
  CameraModel intr(fx, fy, cx, cy);
  projectToPointCloud(d, c, intr, level);
  return down(c, cloud, rows, cols);
}

int ref_cf_computeNIDImg(const unsigned char* img_kf, const unsigned char* img_kf_old, const float* dmap_kf, const float* dmap_kf_old,
                         const unsigned char* img_curr, int rows, int cols, int num_bins, float* nid) {
  DeviceArray2D<unsigned char> a, b, c;
  DeviceArray2D<float> da, db;
  up(a, img_kf, rows, cols);
  up(b, img_kf_old, rows, cols);
  up(c, img_curr, rows, cols);
  up(da, dmap_kf, rows, cols);
  up(db, dmap_kf_old, rows, cols);
  float v = 0.f;
  computeNIDImg(a, b, da, db, c, v, num_bins, false);
  *nid = v;
  return 0;
}

int ref_cf_computeNIDDepth(const float* dmap_kf, const float* dmap_kf_old, const float* dmap_curr, int rows, int cols, int num_bins,
                           float max_depth, float* nid) {
  DeviceArray2D<float> a, b, c;
  up(a, dmap_kf, rows, cols);
  up(b, dmap_kf_old, rows, cols);
  up(c, dmap_curr, rows, cols);
  float v = 0.f;
  computeNIDDepth(a, b, c, v, num_bins, max_depth, false);
  *nid = v;
  return 0;
}

}  // extern "C"
