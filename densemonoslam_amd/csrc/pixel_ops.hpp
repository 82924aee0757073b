// Per-pixel residual / Jacobian-row functions of the tracker.  Shared by the operator-layer
// kernels (reduce.hip) and the fused device-resident Gauss-Newton loop (track.hip).
#pragma once
#include "common.hpp"

namespace dms {

// ---- optional fused multiply-adds ----------------------------------------------------------------------------------
// FMA = false: every multiply and add rounds on its own, in the order written (what the operator layer and the oracle
// evaluate: bit-identical rows).  FMA = true (resident tracker kernels): the same expressions with each multiply-add chain
// fused, the association order unchanged — what nvcc's default -fmad=true does to the reference's own kernels; a row
// element then differs from the unfused one by at most an ulp or two, and the pixel passes, which are bound by their
// instruction count, lose about a quarter of their arithmetic instructions.
template <bool FMA>
__device__ __forceinline__ float madd(float a, float b, float c) {
  return FMA ? fmaf(a, b, c) : a * b + c;
}
template <bool FMA>
__device__ __forceinline__ float dot3t(const f3& a, const f3& b) {  // == dot3 for FMA = false (sums commute)
  return madd<FMA>(a.z, b.z, madd<FMA>(a.y, b.y, a.x * b.x));
}
template <bool FMA>
__device__ __forceinline__ f3 mult(const M33& m, const f3& a) {
  return mk3(dot3t<FMA>(m.r0, a), dot3t<FMA>(m.r1, a), dot3t<FMA>(m.r2, a));
}
template <bool FMA>
__device__ __forceinline__ f3 cross3t(const f3& a, const f3& b) {
  if (FMA) return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
  return cross3(a, b);
}
template <bool FMA>
__device__ __forceinline__ float norm3t(const f3& a) {
  return sqrtf(dot3t<FMA>(a, a));
}

// ---- ICP: projective association + point-to-plane row ---------------------------------
// reference ICPReduction::search / getProducts (reduce.cu:259-344)
struct IcpParams {
  M33 Rcurr;
  f3 tcurr;
  M33 Rprev_inv;
  f3 tprev;
  float fx, fy, cx, cy;
  float distThres, angleThres;
  // the two thresholds as bounds on the SQUARED quantities (sqrt_le_bound / sqrt_lt_bound below): the correctly rounded
  // square root is monotone, so sqrtf(x) <= T and sqrtf(x) < T are each equivalent to x <= B for one float B — the loop
  // compares the sums of squares and takes no square root (two of them per pixel and iteration, ~12 instructions each)
  float dist2Le, sine2Le;
  int cols, rows;
};

// largest float x with sqrtf(x) <= T (strict = false) or sqrtf(x) < T (strict = true); -1 when no x >= 0 qualifies
inline float sqrt_bound(float T, bool strict) {
  auto ok = [&](float x) { return strict ? sqrtf(x) < T : sqrtf(x) <= T; };
  if (!(T == T) || !ok(0.f)) return -1.f;
  if (T > 1.8e19f) return strict ? 3.402823466e+38F : __builtin_inff();  // T * T overflows: every finite x qualifies (inf only without strict)
  float x = (float)((double)T * (double)T);
  while (ok(x)) x = nextafterf(x, __builtin_inff());
  while (!ok(x)) x = nextafterf(x, 0.f);
  return x;
}
inline float sqrt_le_bound(float T) { return sqrt_bound(T, false); }
inline float sqrt_lt_bound(float T) { return sqrt_bound(T, true); }

struct MapPtrs {  // stacked-plane maps: plane stride = rows * pitch
  const float* vcurr;
  unsigned vcurr_pitch;
  const float* ncurr;
  unsigned ncurr_pitch;
  const float* vprev;
  unsigned vprev_pitch;
  const float* nprev;
  unsigned nprev_pitch;
};

// gathers of the Gauss-Newton loop: one 32-bit byte offset (24-bit multiply: rows and pitches are far below 2^24) added to
// the image's base pointer, instead of a 64-bit multiply-add per load (quarter-rate instructions in a loop that is bound by
// vector-instruction issue)
__device__ __forceinline__ unsigned goff(int y, unsigned pitch, int x, unsigned elem) { return __umul24((unsigned)y, pitch) + (unsigned)x * elem; }
template <typename T>
__device__ __forceinline__ T gld(const T* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ const float* prow(const float* base, size_t pitch, int y) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)y * pitch);
}

// The row is evaluated in three stages so that a caller can put the loads of several pixels (and
// of the photometric term) in flight together: a pixel costs two dependent memory round trips
// (own vertex/normal -> projected model vertex/normal) instead of one per early-out test.
// Speculative loads use clamped addresses; the arithmetic and its order are the reference's.
struct IcpOwn {  // stage A: this pixel's vertex and normal (pose independent)
  f3 vcurr, ncurr;
};
struct IcpProj {  // stage B: projection into the model frame
  f3 vcurr_g;
  int ux, uy;
  bool ok;
};
struct IcpModel {  // stage C: model vertex / normal under the projection
  f3 vprev_g, nprev_g;
};

__device__ __forceinline__ IcpOwn icp_load_own(const MapPtrs& m, int x, int y, int rows) {
  IcpOwn o;
  o.vcurr.x = prow(m.vcurr, m.vcurr_pitch, y)[x];
  o.vcurr.y = prow(m.vcurr, m.vcurr_pitch, y + rows)[x];
  o.vcurr.z = prow(m.vcurr, m.vcurr_pitch, y + 2 * rows)[x];
  o.ncurr.x = prow(m.ncurr, m.ncurr_pitch, y)[x];
  o.ncurr.y = prow(m.ncurr, m.ncurr_pitch, y + rows)[x];
  o.ncurr.z = prow(m.ncurr, m.ncurr_pitch, y + 2 * rows)[x];
  return o;
}

template <bool FMA = false>
__device__ __forceinline__ IcpProj icp_project(const IcpParams& p, const IcpOwn& o) {
  IcpProj r;
  r.vcurr_g = mult<FMA>(p.Rcurr, o.vcurr) + p.tcurr;
  const f3 vcurr_cp = mult<FMA>(p.Rprev_inv, r.vcurr_g - p.tprev);
  r.ux = f2i_rn((vcurr_cp.x * p.fx) / vcurr_cp.z + p.cx);
  r.uy = f2i_rn((vcurr_cp.y * p.fy) / vcurr_cp.z + p.cy);
  // 0 <= ux < cols && 0 <= uy < rows (one unsigned compare each) && !(z < 0)
  r.ok = (int)((unsigned)r.ux < (unsigned)p.cols) & (int)((unsigned)r.uy < (unsigned)p.rows) & (int)!(vcurr_cp.z < 0.f);
  return r;
}

__device__ __forceinline__ IcpModel icp_load_model(const MapPtrs& m, const IcpProj& r, int rows) {
  const int ux = r.ok ? r.ux : 0, uy = r.ok ? r.uy : 0;
  IcpModel c;
  const unsigned vo = goff(uy, m.vprev_pitch, ux, 4u), vp = __umul24((unsigned)rows, m.vprev_pitch);
  const unsigned no = goff(uy, m.nprev_pitch, ux, 4u), np = __umul24((unsigned)rows, m.nprev_pitch);
  c.vprev_g.x = gld(m.vprev, vo);
  c.vprev_g.y = gld(m.vprev, vo + vp);
  c.vprev_g.z = gld(m.vprev, vo + 2u * vp);
  c.nprev_g.x = gld(m.nprev, no);
  c.nprev_g.y = gld(m.nprev, no + np);
  c.nprev_g.z = gld(m.nprev, no + 2u * np);
  return c;
}

// row[0..5] = Jacobian, row[6] = residual.  Returns found flag; row is zero when not found.
template <bool FMA = false>
__device__ __forceinline__ bool icp_finish(const IcpParams& p, const IcpOwn& o, const IcpProj& r, const IcpModel& c, float (&row)[7]) {
#pragma unroll
  for (int i = 0; i < 7; ++i) row[i] = 0.f;
  const f3 ncurr_g = mult<FMA>(p.Rcurr, o.ncurr);
  // dist = |vprev_g - vcurr_g| <= distThres, sine = |ncurr_g x nprev_g| < angleThres, on the squares (IcpParams)
  const f3 dv = c.vprev_g - r.vcurr_g, cn = cross3t<FMA>(ncurr_g, c.nprev_g);
  const float dist2 = dot3t<FMA>(dv, dv), sine2 = dot3t<FMA>(cn, cn);
  const bool found = r.ok && (sine2 <= p.sine2Le && dist2 <= p.dist2Le && !isnan(o.ncurr.x) && !isnan(c.nprev_g.x));
  if (!found) return false;
  const f3 s_cp = mult<FMA>(p.Rprev_inv, r.vcurr_g - p.tprev);
  const f3 d_cp = mult<FMA>(p.Rprev_inv, c.vprev_g - p.tprev);
  const f3 n_cp = mult<FMA>(p.Rprev_inv, c.nprev_g);
  const f3 cr = cross3t<FMA>(s_cp, n_cp);
  row[0] = n_cp.x;
  row[1] = n_cp.y;
  row[2] = n_cp.z;
  row[3] = cr.x;
  row[4] = cr.y;
  row[5] = cr.z;
  row[6] = dot3t<FMA>(n_cp, s_cp - d_cp);
  return true;
}

__device__ __forceinline__ bool icp_row(const IcpParams& p, const MapPtrs& m, int x, int y, float (&row)[7]) {
  const IcpOwn o = icp_load_own(m, x, y, p.rows);
  const IcpProj r = icp_project(p, o);
  const IcpModel c = icp_load_model(m, r, p.rows);
  return icp_finish(p, o, r, c, row);
}

// accumulate the 27 upper-triangle products + residual^2 + inlier into acc[29]
// (field order of JtJJtrSE3, types.cuh:123-136)
template <bool FMA = false>
__device__ __forceinline__ void accumulate_se3(float (&acc)[kSE3], const float (&row)[7], bool found) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 7; ++j) {
      acc[k] = madd<FMA>(row[i], row[j], acc[k]);
      ++k;
    }
  acc[27] = madd<FMA>(row[6], row[6], acc[27]);
  acc[28] += found ? 1.f : 0.f;
}

// ---- RGB correspondences (reference RGBResidual::getProducts, reduce.cu:767-843) -------
struct RgbResParams {
  float minScale;
  float maxDepthDelta;
  f3 kt;
  M33 krkinv;
  int cols, rows;
};
struct RgbResPtrs {
  const short* dIdx;
  unsigned dI_pitch;
  const short* dIdy;
  const float* lastDepth;
  unsigned lastDepth_pitch;
  const float* nextDepth;
  unsigned nextDepth_pitch;
  const unsigned char* lastImage;
  unsigned lastImage_pitch;
  const unsigned char* nextImage;
  unsigned nextImage_pitch;
  // optional precomputed pose-independent gate (k_rgb_gate); null = evaluate window and gradient here
  const unsigned char* gate;
  unsigned gate_pitch;
};

template <typename T>
__device__ __forceinline__ const T* trow(const T* base, size_t pitch, int y) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)y * pitch);
}

// Staged like the ICP row: stage A is pose independent (4x4 window test, gradient gate, own
// depth), stage B reads the model depth / intensity under the projection.
struct RgbOwn {
  bool gate;  // inside the border, 4x4 window all non-zero, gradient above minScale, depth not NaN
  float d1;
  unsigned char i1;
};
struct RgbProj {
  float transformed_d1;
  int u0, v0;
  bool ok;
};
struct RgbModel {
  float d0;
  unsigned char l0;
};

// gated variant: window / gradient test read from the per-frame gate image (3 loads instead of 19)
__device__ __forceinline__ RgbOwn rgb_load_own_gated(const RgbResPtrs& q, int j0, int i) {
  RgbOwn o;
  const unsigned char g = trow(q.gate, q.gate_pitch, i)[j0];
  o.i1 = trow(q.nextImage, q.nextImage_pitch, i)[j0];
  o.d1 = trow(q.nextDepth, q.nextDepth_pitch, i)[j0];
  o.gate = (int)(g != 0) & (int)!isnan(o.d1);
  return o;
}

__device__ __forceinline__ RgbOwn rgb_load_own(const RgbResParams& p, const RgbResPtrs& q, int j0, int i) {
  const int cols = p.cols, rows = p.rows;
  const bool inside = (j0 < cols - 5 && i < rows - 1);
  // taps outside the image are clamped onto taps that are inside the reference's clipped window,
  // so the AND over the 16 loads equals the AND over the clipped window
  unsigned acc = 1u;
  unsigned char centre = 0;
#pragma unroll
  for (int du = -2; du < 2; ++du) {
    const int u = min(max(i + du, 0), rows - 1);
    const unsigned char* r = trow(q.nextImage, q.nextImage_pitch, u);
#pragma unroll
    for (int dv = -2; dv < 2; ++dv) {
      const int v = min(max(j0 + dv, 0), cols - 1);
      const unsigned char t = r[v];
      acc &= (t > 0) ? 1u : 0u;
      if (du == 0 && dv == 0) centre = t;
    }
  }
  const short valx = trow(q.dIdx, q.dI_pitch, i)[j0];
  const short valy = trow(q.dIdy, q.dI_pitch, i)[j0];
  RgbOwn o;
  o.d1 = trow(q.nextDepth, q.nextDepth_pitch, i)[j0];
  o.i1 = centre;
  const float mTwo = (float)((int)valx * (int)valx + (int)valy * (int)valy);
  // bitwise, not short-circuit: a branch here makes the compiler sink the gradient loads behind it
  o.gate = (int)inside & (int)(acc != 0u) & (int)(mTwo >= p.minScale) & (int)!isnan(o.d1);
  return o;
}

template <bool FMA = false>
__device__ __forceinline__ RgbProj rgb_project(const RgbResParams& p, const RgbOwn& o, int x, int y) {
  const float d1 = o.d1;
  const float fx_ = (float)x, fy_ = (float)y;
  RgbProj r;
  // d1 * ((k.x * x + k.y * y) + k.z) + kt
  r.transformed_d1 = madd<FMA>(d1, madd<FMA>(p.krkinv.r2.y, fy_, p.krkinv.r2.x * fx_) + p.krkinv.r2.z, p.kt.z);
  r.u0 = f2i_rn(madd<FMA>(d1, madd<FMA>(p.krkinv.r0.y, fy_, p.krkinv.r0.x * fx_) + p.krkinv.r0.z, p.kt.x) / r.transformed_d1);
  r.v0 = f2i_rn(madd<FMA>(d1, madd<FMA>(p.krkinv.r1.y, fy_, p.krkinv.r1.x * fx_) + p.krkinv.r1.z, p.kt.y) / r.transformed_d1);
  r.ok = (int)o.gate & (int)((unsigned)r.u0 < (unsigned)p.cols) & (int)((unsigned)r.v0 < (unsigned)p.rows);
  return r;
}

__device__ __forceinline__ RgbModel rgb_load_model(const RgbResPtrs& q, const RgbProj& r) {
  const int u0 = r.ok ? r.u0 : 0, v0 = r.ok ? r.v0 : 0;
  RgbModel m;
  m.d0 = gld(q.lastDepth, goff(v0, q.lastDepth_pitch, u0, 4u));
  m.l0 = gld(q.lastImage, goff(v0, q.lastImage_pitch, u0, 1u));
  return m;
}

// Returns validity; fills the DataTerm fields.  diff2_int is int(diff*diff) (reference
// stores corres.diff * corres.diff into an int, reduce.cu:831).
__device__ __forceinline__ bool rgb_finish(const RgbResParams& p, const RgbOwn& o, const RgbProj& r, const RgbModel& m, int x, int y,
                                           dms_dataterm& out, int& diff2_int) {
  out.zero_x = out.zero_y = out.one_x = out.one_y = 0;
  out.diff = 0.f;
  out.valid = 0;
  diff2_int = 0;
  if (!(r.ok && m.d0 > 0.f && fabsf(r.transformed_d1 - m.d0) <= p.maxDepthDelta && m.l0 != 0)) return false;
  out.zero_x = (short)r.u0;
  out.zero_y = (short)r.v0;
  out.one_x = (short)x;
  out.one_y = (short)y;
  out.diff = (float)o.i1 - (float)m.l0;
  out.valid = 1;
  diff2_int = f2i_rz(out.diff * out.diff);
  return true;
}

__device__ __forceinline__ bool rgb_residual(const RgbResParams& p, const RgbResPtrs& q, int j0, int i, dms_dataterm& out,
                                             int& diff2_int) {
  const RgbOwn o = rgb_load_own(p, q, j0, i);
  const RgbProj r = rgb_project(p, o, j0, i);
  const RgbModel m = rgb_load_model(q, r);
  return rgb_finish(p, o, r, m, j0, i, out, diff2_int);
}

// ---- RGB Jacobian row (reference RGBReduction::getProducts, reduce.cu:561-620) ---------
struct RgbStepParams {
  float sigma, fx, fy, sobelScale;
};

struct RgbRowIn {  // loads of one correspondence: cloud point of the model pixel, gradient of the live pixel
  f3 pt;
  short gx, gy;
};

// safe for an invalid correspondence (its coordinates are all zero)
__device__ __forceinline__ RgbRowIn rgb_row_load(const dms_dataterm& c, const float* __restrict__ cloud, size_t cloud_pitch,
                                                 const short* __restrict__ dIdx, const short* __restrict__ dIdy, size_t dI_pitch) {
  RgbRowIn in;
  const float* cp = trow(cloud, cloud_pitch, c.zero_y) + 3 * c.zero_x;
  in.pt = mk3(cp[0], cp[1], cp[2]);
  in.gx = trow(dIdx, dI_pitch, c.one_y)[c.one_x];
  in.gy = trow(dIdy, dI_pitch, c.one_y)[c.one_x];
  return in;
}

// row is all zero for an invalid correspondence
// reference: float invz = 1.0 / cloudPoint.z  (double division rounded to float).  Rounding the correctly rounded
// double quotient of two floats to float gives the correctly rounded float quotient (double rounding is innocuous for
// division when the wider format has at least 2 * 24 + 2 bits): the IEEE float division is the same value, a third of
// the instructions.  (sigma-independent: the resident kernels take it while the correspondence count is still in flight.)
__device__ __forceinline__ float rgb_row_invz(const f3& pt) { return 1.0f / pt.z; }

template <bool FMA = false>
__device__ __forceinline__ void rgb_row_finish(const RgbStepParams& p, const dms_dataterm& c, const RgbRowIn& in, float (&row)[7], float invz_pre = 0.f,
                                               bool have_invz = false) {
#pragma unroll
  for (int i = 0; i < 7; ++i) row[i] = 0.f;
  if (!c.valid) return;
  float w = p.sigma + fabsf(c.diff);
  w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
  if (p.sigma == -1.f) w = 1.f;
  row[6] = -w * c.diff;
  const f3 pt = in.pt;
  const float invz = have_invz ? invz_pre : rgb_row_invz(pt);
  const float dI_dx_val = (w * p.sobelScale) * (float)in.gx;
  const float dI_dy_val = (w * p.sobelScale) * (float)in.gy;
  const float v0 = (dI_dx_val * p.fx) * invz;
  const float v1 = (dI_dy_val * p.fy) * invz;
  const float v2 = -madd<FMA>(v1, pt.y, v0 * pt.x) * invz;
  row[0] = v0;
  row[1] = v1;
  row[2] = v2;
  row[3] = madd<FMA>(pt.y, v2, -pt.z * v1);
  row[4] = FMA ? fmaf(-pt.x, v2, pt.z * v0) : pt.z * v0 - pt.x * v2;
  row[5] = madd<FMA>(pt.x, v1, -pt.y * v0);
}

// (the caller zero-fills row and skips invalid correspondences, as the reference does)
__device__ __forceinline__ void rgb_row(const RgbStepParams& p, const dms_dataterm& c, const float* __restrict__ cloud,
                                        size_t cloud_pitch, const short* __restrict__ dIdx, const short* __restrict__ dIdy,
                                        size_t dI_pitch, float (&row)[7]) {
  const RgbRowIn in = rgb_row_load(c, cloud, cloud_pitch, dIdx, dIdy, dI_pitch);
  rgb_row_finish(p, c, in, row);
}

// ---- SO3 (reference SO3Reduction::getProducts, reduce.cu:942-1032) ----------------------
struct So3Params {
  M33 imageBasis, kinv, krlr;
  int cols, rows;
};

__device__ __forceinline__ void so3_gradient(const unsigned char* img, size_t pitch, int x, int y, float& gx, float& gy) {
  const float actu = (float)trow(img, pitch, y)[x];
  float back = (float)trow(img, pitch, y)[x - 1];
  float fore = (float)trow(img, pitch, y)[x + 1];
  gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
  back = (float)trow(img, pitch, y - 1)[x];
  fore = (float)trow(img, pitch, y + 1)[x];
  gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

__device__ __forceinline__ bool so3_row(const So3Params& p, const unsigned char* lastImage, size_t last_pitch,
                                        const unsigned char* nextImage, size_t next_pitch, int x, int y, float (&row)[4]) {
  row[0] = row[1] = row[2] = row[3] = 0.f;
  const f3 unwarped = mk3((float)x, (float)y, 1.0f);
  const f3 warped = mul(p.imageBasis, unwarped);
  const int wx = f2i_rn(warped.x / warped.z);
  const int wy = f2i_rn(warped.y / warped.z);
  const int cols = p.cols, rows = p.rows;
  if (!(wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1)) return false;
  float gnx, gny, glx, gly;
  so3_gradient(nextImage, next_pitch, wx, wy, gnx, gny);
  so3_gradient(lastImage, last_pitch, x, y, glx, gly);
  const float gx = (gnx + glx) / 2.0f;
  const float gy = (gny + gly) / 2.0f;
  const f3 point = mul(p.kinv, unwarped);
  const float z2 = point.z * point.z;
  const float a = p.krlr.r0.x, b = p.krlr.r0.y, c = p.krlr.r0.z;
  const float d = p.krlr.r1.x, e = p.krlr.r1.y, f = p.krlr.r1.z;
  const float g = p.krlr.r2.x, h = p.krlr.r2.y, i = p.krlr.r2.z;
  const float fx = (float)x, fy = (float)y;
  const f3 left = mk3((((point.z * (d * gy + a * gx)) - ((gy * g) * fy)) - ((gx * g) * fx)) / z2,
                      (((point.z * (e * gy + b * gx)) - ((gy * h) * fy)) - ((gx * h) * fx)) / z2,
                      (((point.z * (f * gy + c * gx)) - ((gy * i) * fy)) - ((gx * i) * fx)) / z2);
  const f3 jac = cross3(left, point);
  row[0] = jac.x;
  row[1] = jac.y;
  row[2] = jac.z;
  row[3] = -((float)trow(nextImage, next_pitch, wy)[wx] - (float)trow(lastImage, last_pitch, y)[x]);
  return true;
}

__device__ __forceinline__ void accumulate_so3(float (&acc)[kSO3], const float (&row)[4], bool found) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) acc[k++] += row[i] * row[j];
  acc[9] += row[3] * row[3];
  acc[10] += found ? 1.f : 0.f;
}

}  // namespace dms
