// Model-side pyramid bodies of the frame step (initICPModel + initRGBModel, ElasticFusion.cpp:172-189), shared by the model pyramid
// kernel (prep.hip: k_model_levels012, 64 x 4 blocks) and - round 6 - by the tracker's first launch, which can run the same bodies as
// rider blocks beside the resident SO3 stage (track.hip).  Every body takes the thread's place in its block explicitly.
#pragma once
#include "common.hpp"
#include "fill.hpp"
#include "live_bodies.hpp"

namespace dms {

struct ModelSrc {
  const float4* vA;  // predicted vertex / normal / image
  const float4* nA;
  const uchar4* iA;
  const float4* vB;  // fill-in vertex / normal / image
  const float4* nB;
  const uchar4* iB;
  const int* flag;   // device flag: 1 = use the fill-in maps
  // instead of the flag: the 16 counters of the prediction's resolve pass (fill.hpp) and the sample count; the decision
  // taken from them is also stored to *flag_out (block 0), for the result block and later readers of the flag
  const unsigned* dense_cnt;
  int dense_samples;
  int* flag_out;
  int force_b_img;   // frameToFrameRGB: always the fill-in image
  const float* pose16;  // model pose for transformMaps; null = leave the maps in the camera frame (live side, initICP from maps)
};

__device__ __forceinline__ bool model_use_b(const ModelSrc& m) {
  return (m.dense_cnt ? dense_from_counters(m.dense_cnt, m.dense_samples) : *m.flag) != 0;
}

struct Pose34 {
  M33 R;
  f3 t;
};
__device__ __forceinline__ Pose34 load_pose34(const float* pose16) {
  Pose34 p;
  if (!pose16) {
    p.R.r0 = p.R.r1 = p.R.r2 = p.t = mk3(0.f, 0.f, 0.f);
    return p;
  }
  p.R.r0 = mk3(pose16[0], pose16[1], pose16[2]);
  p.R.r1 = mk3(pose16[4], pose16[5], pose16[6]);
  p.R.r2 = mk3(pose16[8], pose16[9], pose16[10]);
  p.t = mk3(pose16[3], pose16[7], pose16[11]);
  return p;
}
// copyMaps rule: z == 0 in the VERTEX makes both the vertex and the normal NaN
__device__ __forceinline__ void raw_maps(const float4 v, const float4 q, f3& rv, f3& rn) {
  const bool ok = !(v.z == 0.f);
  const float n = qnan();
  rv = ok ? mk3(v.x, v.y, v.z) : mk3(n, n, n);
  rn = ok ? mk3(q.x, q.y, q.z) : mk3(n, n, n);
}
// transformMaps rule + store into stacked planes
__device__ __forceinline__ void store_transformed(View<float> vmap, View<float> nmap, int rows, int y, int x, const f3& rv, const f3& rn,
                                                  const Pose34& P, bool xf) {
  if (!isnan(rv.x)) {
    const f3 d = xf ? mul(P.R, rv) + P.t : rv;
    vmap.at(y, x) = d.x;
    vmap.at(y + rows, x) = d.y;
    vmap.at(y + 2 * rows, x) = d.z;
  } else {
    vmap.at(y, x) = qnan();
  }
  if (!isnan(rn.x)) {
    const f3 d = xf ? mul(P.R, rn) : rn;
    nmap.at(y, x) = d.x;
    nmap.at(y + rows, x) = d.y;
    nmap.at(y + 2 * rows, x) = d.z;
  } else {
    nmap.at(y, x) = qnan();
  }
}

// level 0: transformed maps, float depth (verticesToDepth) and intensity, one thread per pixel
// (tx, ty, BYv: the thread's place in a 64 x BYv block - the built-in indices of a 2-D launch, derived from the linear id in a 1-D one)
__device__ __forceinline__ void model_level0_body(int tx, int ty, int BYv, int bx, int by, const ModelSrc& m, int rows, int cols, View<float> vmap,
                                                  View<float> nmap, View<float> depth, View<unsigned char> inten, float cutOff) {
  const int x = bx * 64 + tx;
  const int y = by * BYv + ty;
  if (x >= cols || y >= rows) return;
  const bool useB = model_use_b(m);
  const size_t i = (size_t)y * cols + x;
  const float4 v = useB ? m.vB[i] : m.vA[i];
  const float4 q = useB ? m.nB[i] : m.nA[i];
  const uchar4 c = (useB || m.force_b_img) ? m.iB[i] : m.iA[i];
  const Pose34 P = load_pose34(m.pose16);
  f3 rv, rn;
  raw_maps(v, q, rv, rn);
  store_transformed(vmap, nmap, rows, y, x, rv, rn, P, m.pose16 != nullptr);
  depth.at(y, x) = (v.z > cutOff || v.z <= 0.f) ? qnan() : v.z;
  const float f = ((float)c.x * 0.114f + (float)c.y * 0.299f) + (float)c.z * 0.587f;
  inten.at(y, x) = (unsigned char)f2i_rz(f);
}

// resizeMap rule on four raw values
template <bool NORMALIZE>
__device__ __forceinline__ f3 resize4(const f3& a, const f3& b, const f3& c, const f3& d) {
  if (isnan(a.x) || isnan(b.x) || isnan(c.x) || isnan(d.x)) return mk3(qnan(), qnan(), qnan());
  f3 n;
  n.x = (a.x + b.x + c.x + d.x) / 4;
  n.y = (a.y + b.y + c.y + d.y) / 4;
  n.z = (a.z + b.z + c.z + d.z) / 4;
  if (NORMALIZE) n = normalized3(n);
  return n;
}

// levels 1 and 2: one thread per level-1 pixel (its 2x2 level-0 pixels); the four level-1 pixels under one level-2 pixel
// sit in four neighbouring lanes (k = lane & 3: x offset k & 1, y offset k >> 1), lane 0 of the quad gathers the four
// level-1 values and writes level 2.  64 x BY thread blocks: 16 x BY level-2 pixels per block.  (One thread per level-2
// pixel — 32 strided 16-byte loads per lane on 75 blocks — took 10.8 of the launch's 17.7 us.)
__device__ __forceinline__ void model_levels12_body(int tx, int ty, int BYv, int bx, int by, const ModelSrc& m, int cols0, int rows1, int cols1, int rows2,
                                                    int cols2, View<float> v1, View<float> n1, View<float> v2, View<float> n2) {
  const int lane = tx, k = lane & 3;
  const int x2 = bx * 16 + (lane >> 2);
  const int y2 = by * BYv + ty;
  const int x1 = 2 * x2 + (k & 1), y1 = 2 * y2 + (k >> 1);
  const bool useB = model_use_b(m);
  const float4* vs = useB ? m.vB : m.vA;
  const float4* ns = useB ? m.nB : m.nA;
  const Pose34 P = load_pose34(m.pose16);
  f3 lv = mk3(0.f, 0.f, 0.f), ln = mk3(0.f, 0.f, 0.f);
  if (x1 < cols1 && y1 < rows1) {
    f3 rv[4], rn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t i = (size_t)(2 * y1 + (j >> 1)) * cols0 + (2 * x1 + (j & 1));
      raw_maps(vs[i], ns[i], rv[j], rn[j]);
    }
    lv = resize4<false>(rv[0], rv[1], rv[2], rv[3]);
    ln = resize4<true>(rn[0], rn[1], rn[2], rn[3]);
    store_transformed(v1, n1, rows1, y1, x1, lv, ln, P, m.pose16 != nullptr);
  }
  f3 qv[4], qn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int src = (lane & ~3) | j;
    qv[j] = mk3(__shfl(lv.x, src, 64), __shfl(lv.y, src, 64), __shfl(lv.z, src, 64));
    qn[j] = mk3(__shfl(ln.x, src, 64), __shfl(ln.y, src, 64), __shfl(ln.z, src, 64));
  }
  if (k == 0 && x2 < cols2 && y2 < rows2) {  // then all four level-1 pixels exist
    const f3 a = resize4<false>(qv[0], qv[1], qv[2], qv[3]);
    const f3 b = resize4<true>(qn[0], qn[1], qn[2], qn[3]);
    store_transformed(v2, n2, rows2, y2, x2, a, b, P, m.pose16 != nullptr);
  }
}

// first pyramid step of the model depth / intensity images, taken straight from the sources: every tap re-derives the
// level-0 depth (verticesToDepth rule) and intensity (imageBGRToIntensity rule) that model_level0_body stores,
// so the step does not have to wait for level 0
template <int BYV>
__device__ __forceinline__ void model_pyr_step1_body(int tx, int ty, int bx, int by, const ModelSrc& m, int rows0, int cols0, float cutOff, View<float> ddst,
                                                     View<unsigned char> idst) {
  constexpr int BX = 64, BY = BYV;
  // The block's 64 x BY outputs read a (2 * 64 + 3) x (2 * BY + 3) window of level 0: every source pixel is converted once,
  // by coalesced loads, into LDS (25 strided taps per output straight from the float4 / uchar4 sources made this group
  // the longest of the launch: 12 us alone); the taps, their order and their arithmetic are unchanged.
  constexpr int TW = 2 * BX + 3, TH = 2 * BY + 3;
  __shared__ float s_d[TH][TW + 1];
  __shared__ unsigned char s_c[TH][TW + 1];
  const bool useB = model_use_b(m);
  const float4* vs = useB ? m.vB : m.vA;
  const uchar4* is = (useB || m.force_b_img) ? m.iB : m.iA;
  const int sx0 = 2 * (bx * BX) - 2, sy0 = 2 * (by * BY) - 2;
  for (int e = ty * BX + tx; e < TH * TW; e += BX * BY) {
    const int r = e / TW, c = e - r * TW;
    const int gy = sy0 + r, gx = sx0 + c;
    float sv = qnan();
    unsigned char cv = 0;
    if (gx >= 0 && gy >= 0 && gx < cols0 && gy < rows0) {
      const size_t i = (size_t)gy * cols0 + gx;
      const float z = vs[i].z;
      sv = (z > cutOff || z <= 0.f) ? qnan() : z;
      const uchar4 cc = is[i];
      cv = (unsigned char)f2i_rz(((float)cc.x * 0.114f + (float)cc.y * 0.299f) + (float)cc.z * 0.587f);
    }
    s_d[r][c] = sv;
    s_c[r][c] = cv;
  }
  __syncthreads();
  const int x = bx * BX + tx;
  const int y = by * BY + ty;
  if (x >= ddst.cols || y >= ddst.rows) return;
  const live::Tile<float, TW + 1> dt = {&s_d[0][0], sy0, sx0};
  const live::Tile<unsigned char, TW + 1> ct = {&s_c[0][0], sy0, sx0};
  ddst.at(y, x) = live::float_half(dt, x, y, cols0, rows0);
  idst.at(y, x) = live::u8_half(ct, x, y, cols0, rows0);
}

// BOTH pyramid steps of the model depth / intensity images for a tile of LEVEL-2 pixels, straight from the sources (round 6): the
// block converts the level-0 window its outputs depend on into LDS (as model_pyr_step1_body does), takes the first step into a
// level-1 tile in LDS - the same float_half / u8_half on the same values, so the tile holds exactly what model_pyr_step1_body
// stores to the level-1 images - and the second step from that tile.  The pyramid's last step then needs no launch boundary after
// the first one (it rides on the tracker's first launch otherwise: k_so3_level's rider blocks).  16 x 8 level-2 pixels per block
// of NT threads: level-1 window 35 x 19, level-0 window 73 x 41.
template <int NT>
__device__ __forceinline__ void model_pyr_step2_body(int t, int bx, int by, const ModelSrc& m, int rows0, int cols0, float cutOff, int rows1, int cols1,
                                                     View<float> ddst2, View<unsigned char> idst2) {
  constexpr int OX = 16, OY = 8;
  constexpr int W1 = 2 * OX + 3, H1 = 2 * OY + 3, W0 = 2 * W1 + 3, H0 = 2 * H1 + 3;
  __shared__ float s_d0[H0][W0 + 1];
  __shared__ unsigned char s_c0[H0][W0 + 1];
  __shared__ float s_d1[H1][W1 + 1];
  __shared__ unsigned char s_c1[H1][W1 + 1];
  const bool useB = model_use_b(m);
  const float4* vs = useB ? m.vB : m.vA;
  const uchar4* is = (useB || m.force_b_img) ? m.iB : m.iA;
  const int x1o = 2 * (bx * OX) - 2, y1o = 2 * (by * OY) - 2;  // level-1 window origin
  const int x0o = 2 * x1o - 2, y0o = 2 * y1o - 2;              // level-0 window origin
  for (int e = t; e < H0 * W0; e += NT) {
    const int r = e / W0, c = e - r * W0;
    const int gy = y0o + r, gx = x0o + c;
    float sv = qnan();
    unsigned char cv = 0;
    if (gx >= 0 && gy >= 0 && gx < cols0 && gy < rows0) {
      const size_t i = (size_t)gy * cols0 + gx;
      const float z = vs[i].z;
      sv = (z > cutOff || z <= 0.f) ? qnan() : z;
      const uchar4 cc = is[i];
      cv = (unsigned char)f2i_rz(((float)cc.x * 0.114f + (float)cc.y * 0.299f) + (float)cc.z * 0.587f);
    }
    s_d0[r][c] = sv;
    s_c0[r][c] = cv;
  }
  __syncthreads();
  {
    const live::Tile<float, W0 + 1> dt = {&s_d0[0][0], y0o, x0o};
    const live::Tile<unsigned char, W0 + 1> ct = {&s_c0[0][0], y0o, x0o};
    for (int e = t; e < H1 * W1; e += NT) {
      const int r = e / W1, c = e - r * W1;
      const int gy = y1o + r, gx = x1o + c;
      if (gx >= 0 && gy >= 0 && gx < cols1 && gy < rows1) {  // (positions outside level 1 are never read by the second step)
        s_d1[r][c] = live::float_half(dt, gx, gy, cols0, rows0);
        s_c1[r][c] = live::u8_half(ct, gx, gy, cols0, rows0);
      }
    }
  }
  __syncthreads();
  for (int e = t; e < OX * OY; e += NT) {
    const int x = bx * OX + (e % OX), y = by * OY + (e / OX);
    if (x >= ddst2.cols || y >= ddst2.rows) continue;
    const live::Tile<float, W1 + 1> dt = {&s_d1[0][0], y1o, x1o};
    const live::Tile<unsigned char, W1 + 1> ct = {&s_c1[0][0], y1o, x1o};
    ddst2.at(y, x) = live::float_half(dt, x, y, cols1, rows1);
    idst2.at(y, x) = live::u8_half(ct, x, y, cols1, rows1);
  }
}

}  // namespace dms
