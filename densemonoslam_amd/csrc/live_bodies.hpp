// Per-pixel bodies of the live half of the tracker's pyramids (the frame's own depth and colour: RGBDOdometry::initICP(depth)
// and initRGB, RGBDOdometry.cpp:118-142, :244-266, and the Sobel pass of :279-283), written against an ACCESSOR — any
// callable `T src(int y, int x)` — so that one body serves
//   * the operator layer (dms_pyrDown, dms_createVMap, dms_createNMap, dms_pyrDownGaussF, dms_pyrDownUcharGauss,
//     dms_computeDerivativeImages: one thread per output pixel over global memory, prep.hip), and
//   * the fused live-half kernels of the frame step, which compute three pyramid levels from LDS tiles (prep.hip:
//     k_live_levels) and back-project inside the depth filter's epilogue.
// What each body must compute is fixed by the reference kernel it replaces (cited per body) down to the order of the float
// additions — the parity tests demand exact bits against the oracle; how it is staged, tiled and launched is this design's.
#pragma once
#include "common.hpp"

namespace dms {
namespace live {

// ---- accessors ----------------------------------------------------------------------------------------------------
template <typename T>
struct Pitched {  // dense or pitched image in global memory
  const T* base;
  unsigned pitch;  // bytes
  __device__ __forceinline__ T operator()(int y, int x) const {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)y * pitch + (size_t)x * sizeof(T));
  }
};
template <typename T, int W>
struct Tile {  // a W-wide window of an image in LDS whose top-left element is image pixel (y0, x0)
  const T* s;
  int y0, x0;
  __device__ __forceinline__ T operator()(int y, int x) const { return s[(y - y0) * W + (x - x0)]; }
};

// ---- depth pyramid: half-resolution u16 depth through a 5x5 binomial that skips taps more than 90 mm from the centre
// (pyrDownGaussKernel, cudafuncs.cu:57-91 with sigma_color = 30, :100).  Taps: rows / columns 2Y-2 .. 2Y+2 clipped to the
// image, row-major; value = (sum of tap * wx * wy) / (sum of wx * wy), truncated.
template <class Src>
__device__ __forceinline__ unsigned short depth_half(const Src& src, int X, int Y, int scols, int srows) {
  const int cx = 2 * X, cy = 2 * Y;
  const int centre = src(cy, cx);
  const int y_lo = max(cy - 2, 0), y_hi = min(cy + 3, srows), x_lo = max(cx - 2, 0), x_hi = min(cx + 3, scols);
  float num = 0.f, den = 0.f;
  for (int y = y_lo; y < y_hi; ++y) {
    const int ay = abs(y - cy);
    const float wy = ay == 0 ? 0.375f : (ay == 1 ? 0.25f : 0.0625f);
    for (int x = x_lo; x < x_hi; ++x) {
      const int tap = src(y, x);
      const int ax = abs(x - cx);
      const float wx = ax == 0 ? 0.375f : (ax == 1 ? 0.25f : 0.0625f);
      if ((float)abs(tap - centre) < 90.f) {
        num += ((float)tap * wx) * wy;
        den += wx * wy;
      }
    }
  }
  return (unsigned short)f2i_rz(num / den);
}

// ---- the reference's other two pyramid steps share a window and a weight table: rows [max(0, 2Y-2), min(2Y+3, rows-1)) — the
// last source row / column is never read — with the 1 4 6 4 1 weights indexed from the END of the clipped window
// (pyrDownKernelGaussF, cudafuncs.cu:416-443; pyrDownKernelIntensityGauss, :544-573).  `keep(tap)` says which taps count;
// the weight count is an integer.
template <class Src, class Keep>
__device__ __forceinline__ float binom_half(const Src& src, int X, int Y, int scols, int srows, Keep keep) {
  const int y_end = min(2 * Y + 3, srows - 1), x_end = min(2 * X + 3, scols - 1);
  float sum = 0.f;
  int count = 0;
  for (int y = max(0, 2 * Y - 2); y < y_end; ++y) {
    const int ry = y_end - y - 1;  // 0 .. 4 from the window's end
    const float wy = ry == 2 ? 6.f : ((ry == 1 || ry == 3) ? 4.f : 1.f);
    for (int x = max(0, 2 * X - 2); x < x_end; ++x) {
      const int rx = x_end - x - 1;
      const float wx = rx == 2 ? 6.f : ((rx == 1 || rx == 3) ? 4.f : 1.f);
      const float tap = (float)src(y, x);
      if (keep(tap)) {
        const float g = wy * wx;
        sum += tap * g;
        count += (int)g;
      }
    }
  }
  return sum / (float)count;
}
template <class Src>
__device__ __forceinline__ float float_half(const Src& src, int X, int Y, int scols, int srows) {  // NaN taps skipped
  return binom_half(src, X, Y, scols, srows, [](float t) { return !isnan(t); });
}
template <class Src>
__device__ __forceinline__ unsigned char u8_half(const Src& src, int X, int Y, int scols, int srows) {  // zero taps skipped, truncated
  return (unsigned char)f2i_rz(binom_half(src, X, Y, scols, srows, [](float t) { return t > 0.f; }));
}

// ---- back-projection of one depth sample (computeVmapKernel, cudafuncs.cu:106-128): mm -> m, invalid / beyond the cutoff
// -> NaN in x (only x is ever tested)
struct LevelCam {
  float fx_inv, fy_inv, cx, cy;
};
__device__ __forceinline__ f3 vertex_of(unsigned short d, int u, int v, const LevelCam& k, float cutoff) {
  const float z = (float)d / 1000.f;
  if (z != 0.f && z < cutoff) return mk3((z * ((float)u - k.cx)) * k.fx_inv, (z * ((float)v - k.cy)) * k.fy_inv, z);
  return mk3(qnan(), 0.f, 0.f);
}

// ---- normal from forward differences (computeNmapKernel, cudafuncs.cu:149-182): the pixel, its right and its lower
// neighbour; NaN in x when one of them is invalid or the pixel lies on the last row / column
__device__ __forceinline__ f3 normal_of(const f3& here, const f3& right, const f3& below, bool on_border) {
  if (on_border || isnan(here.x) || isnan(right.x) || isnan(below.x)) return mk3(qnan(), 0.f, 0.f);
  return normalized3(cross3(right - here, below - here));
}

// ---- RGBA8 -> intensity (bgr2IntensityKernel, cudafuncs.cu:643-655; weight order as written there)
__device__ __forceinline__ unsigned char intensity_of(uchar4 c) {
  return (unsigned char)f2i_rz(((float)c.x * 0.114f + (float)c.y * 0.299f) + (float)c.z * 0.587f);
}

// ---- Sobel-like derivatives (applyKernel, cudafuncs.cu:674-695: the 3x3 masks are indexed from 8 downward over the CLAMPED
// window, so a border pixel sees shifted masks) + the pose-independent part of the photometric correspondence test
// (RGBResidual::getProducts, reduce.cu:775-797): inside the border, the clipped 4x4 window all non-zero, |gradient|^2 >= minScale
struct Grad {
  short dx, dy;
  unsigned char gate;
};
template <class Src>
__device__ __forceinline__ Grad gradient_gate(const Src& img, int x, int y, int cols, int rows, float minScale) {
  float gx = 0.f, gy = 0.f;
  int k = 8;
  for (int j = max(y - 1, 0); j <= min(y + 1, rows - 1); ++j)
    for (int i = max(x - 1, 0); i <= min(x + 1, cols - 1); ++i) {
      const float p = (float)img(j, i);
      const int kc = k % 3, kr = k / 3;  // position in the 3x3 mask, rows top to bottom
      const float mx = kc == 0 ? (kr == 1 ? 0.79451f : 0.52201f) : (kc == 2 ? (kr == 1 ? -0.79451f : -0.52201f) : (kr == 1 ? -0.f : 0.f));
      const float my = kr == 0 ? (kc == 1 ? 0.79451f : 0.52201f) : (kr == 2 ? (kc == 1 ? -0.79451f : -0.52201f) : 0.f);
      gx += p * mx;
      gy += p * my;
      --k;
    }
  Grad g;
  g.dx = (short)f2i_rz(gx);
  g.dy = (short)f2i_rz(gy);
  bool ok = (x < cols - 5 && y < rows - 1);
  for (int u = max(y - 2, 0); u < min(y + 2, rows); ++u)
    for (int v = max(x - 2, 0); v < min(x + 2, cols); ++v) ok = ok && (img(u, v) > 0);
  const int vx = g.dx, vy = g.dy;
  g.gate = (ok && (float)(vx * vx + vy * vy) >= minScale) ? 1 : 0;
  return g;
}

}  // namespace live
}  // namespace dms
