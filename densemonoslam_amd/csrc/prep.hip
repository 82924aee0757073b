// Pyramid / map preparation kernels for the tracker (operator layer (B)).
// Each kernel states which reference function it replaces.  These are HBM-streaming
// kernels: one 64-wide wave covers 64 consecutive pixels of a row (coalesced 4-byte or
// 16-byte accesses per lane); blocks are 64×4 so a 256-thread block spans 4 rows.
#include "common.hpp"
#include "pyr_body.hpp"
#include "live_bodies.hpp"
#include "fill.hpp"
#include "track_init.hpp"
#include "model_bodies.hpp"

namespace dms {

static constexpr int BX = 64, BY = 4;
static inline dim3 blk() { return dim3(BX, BY); }

// ---------------------------------------------------------------------------------------
// Operator layer of the live-side pyramid: every operator is the generic per-pixel launch below over a functor that applies
// one body of live_bodies.hpp through a global-memory accessor.  (The frame step does not use these launches: it computes the
// same bodies from LDS tiles, k_live_levels further down.)
// ---------------------------------------------------------------------------------------
template <class Op>
__global__ void k_per_pixel(int cols, int rows, Op op) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x < cols && y < rows) op(x, y);
}

struct OpDepthHalf {  // dms_pyrDown (reference pyrDown, cudafuncs.cu:93-104)
  live::Pitched<unsigned short> src;
  View<unsigned short> dst;
  int scols, srows;
  __device__ void operator()(int x, int y) const { dst.at(y, x) = live::depth_half(src, x, y, scols, srows); }
};
struct OpVertexMap {  // dms_createVMap (createVMap, cudafuncs.cu:130-147)
  live::Pitched<unsigned short> depth;
  View<float> vmap;
  live::LevelCam cam;
  float cutoff;
  int rows;
  __device__ void operator()(int u, int v) const {
    const f3 p = live::vertex_of(depth(v, u), u, v, cam, cutoff);
    vmap.at(v, u) = p.x;
    if (!isnan(p.x)) {
      vmap.at(v + rows, u) = p.y;
      vmap.at(v + 2 * rows, u) = p.z;
    }
  }
};
struct OpNormalMap {  // dms_createNMap (createNMap, cudafuncs.cu:184-198)
  View<const float> vmap;
  View<float> nmap;
  int rows, cols;
  __device__ f3 vertex(int v, int u) const {
    f3 p;
    p.x = vmap.at(v, u);
    p.y = p.z = 0.f;
    if (!isnan(p.x)) {
      p.y = vmap.at(v + rows, u);
      p.z = vmap.at(v + 2 * rows, u);
    }
    return p;
  }
  __device__ void operator()(int u, int v) const {
    const bool border = u == cols - 1 || v == rows - 1;
    const f3 nan3 = mk3(qnan(), 0.f, 0.f);
    const f3 n = live::normal_of(vertex(v, u), border ? nan3 : vertex(v, u + 1), border ? nan3 : vertex(v + 1, u), border);
    nmap.at(v, u) = n.x;
    if (!isnan(n.x)) {
      nmap.at(v + rows, u) = n.y;
      nmap.at(v + 2 * rows, u) = n.z;
    }
  }
};

// ---------------------------------------------------------------------------------------
// tranformMaps (reference tranformMapsKernel ×2, cudafuncs.cu:200-274); in-place capable
// ---------------------------------------------------------------------------------------
template <bool WITH_N>
__global__ void k_transformMaps(int rows, int cols, View<const float> vsrc, View<const float> nsrc, M33 R, f3 t,
                                View<float> vdst, View<float> ndst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  {
    f3 s;
    float outx = qnan();
    s.x = vsrc.at(y, x);
    if (!isnan(s.x)) {
      s.y = vsrc.at(y + rows, x);
      s.z = vsrc.at(y + 2 * rows, x);
      const f3 d = mul(R, s) + t;
      vdst.at(y + rows, x) = d.y;
      vdst.at(y + 2 * rows, x) = d.z;
      outx = d.x;
    }
    vdst.at(y, x) = outx;
  }
  if (WITH_N) {
    f3 s;
    float outx = qnan();
    s.x = nsrc.at(y, x);
    if (!isnan(s.x)) {
      s.y = nsrc.at(y + rows, x);
      s.z = nsrc.at(y + 2 * rows, x);
      const f3 d = mul(R, s);
      ndst.at(y + rows, x) = d.y;
      ndst.at(y + 2 * rows, x) = d.z;
      outx = d.x;
    }
    ndst.at(y, x) = outx;
  }
}

// device-pose variant used by the frame step: R | t are read from a row-major 4×4 in HBM so the
// pose never has to visit the host between tracking and map transformation
__global__ void k_transformMaps_dev(int rows, int cols, View<float> vmap, View<float> nmap, const float* __restrict__ pose16) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  M33 R;
  R.r0 = mk3(pose16[0], pose16[1], pose16[2]);
  R.r1 = mk3(pose16[4], pose16[5], pose16[6]);
  R.r2 = mk3(pose16[8], pose16[9], pose16[10]);
  const f3 t = mk3(pose16[3], pose16[7], pose16[11]);
  {
    f3 s;
    float outx = qnan();
    s.x = vmap.at(y, x);
    if (!isnan(s.x)) {
      s.y = vmap.at(y + rows, x);
      s.z = vmap.at(y + 2 * rows, x);
      const f3 d = mul(R, s) + t;
      vmap.at(y + rows, x) = d.y;
      vmap.at(y + 2 * rows, x) = d.z;
      outx = d.x;
    }
    vmap.at(y, x) = outx;
  }
  {
    f3 s;
    float outx = qnan();
    s.x = nmap.at(y, x);
    if (!isnan(s.x)) {
      s.y = nmap.at(y + rows, x);
      s.z = nmap.at(y + 2 * rows, x);
      const f3 d = mul(R, s);
      nmap.at(y + rows, x) = d.y;
      nmap.at(y + 2 * rows, x) = d.z;
      outx = d.x;
    }
    nmap.at(y, x) = outx;
  }
}

// dst = (*flag ? b : a), 16 bytes per element
__global__ void k_select_copy16(float4* __restrict__ dst, const float4* __restrict__ a, const float4* __restrict__ b,
                                const int* __restrict__ flag, int force_b, size_t n) {
  const float4* src = (force_b || (flag && *flag)) ? b : a;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)blockDim.x * gridDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------
// copyMaps: dense RGBA32F -> stacked planes, z == 0 => NaN in all three planes
// (reference copyMapsKernel ×2, cudafuncs.cu:313-378).  One float4 load per lane.
// ---------------------------------------------------------------------------------------
template <bool WITH_N>
__global__ void k_copyMaps(int rows, int cols, const float4* __restrict__ vsrc, const float4* __restrict__ nsrc,
                           View<float> vdst, View<float> ndst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  const float4 v = vsrc[(size_t)y * cols + x];
  const bool ok = !(v.z == 0.f);
  const float n = qnan();
  vdst.at(y, x) = ok ? v.x : n;
  vdst.at(y + rows, x) = ok ? v.y : n;
  vdst.at(y + 2 * rows, x) = ok ? v.z : n;
  if (WITH_N) {
    const float4 q = nsrc[(size_t)y * cols + x];
    ndst.at(y, x) = ok ? q.x : n;
    ndst.at(y + rows, x) = ok ? q.y : n;
    ndst.at(y + 2 * rows, x) = ok ? q.z : n;
  }
}

// ---------------------------------------------------------------------------------------
// resizeVMap / resizeNMap: 2×2 box, NaN-propagating on plane x (reference resizeMapKernel,
// cudafuncs.cu:445-492)
// ---------------------------------------------------------------------------------------
template <bool NORMALIZE>
__global__ void k_resizeMap(int drows, int dcols, int srows, View<const float> in, View<float> out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dcols || y >= drows) return;
  const int xs = x * 2, ys = y * 2;
  const float x00 = in.at(ys, xs), x01 = in.at(ys, xs + 1), x10 = in.at(ys + 1, xs), x11 = in.at(ys + 1, xs + 1);
  if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) {
    out.at(y, x) = qnan();
    return;
  }
  f3 n;
  n.x = (x00 + x01 + x10 + x11) / 4;
  const float y00 = in.at(ys + srows, xs), y01 = in.at(ys + srows, xs + 1), y10 = in.at(ys + srows + 1, xs),
              y11 = in.at(ys + srows + 1, xs + 1);
  n.y = (y00 + y01 + y10 + y11) / 4;
  const float z00 = in.at(ys + 2 * srows, xs), z01 = in.at(ys + 2 * srows, xs + 1), z10 = in.at(ys + 2 * srows + 1, xs),
              z11 = in.at(ys + 2 * srows + 1, xs + 1);
  n.z = (z00 + z01 + z10 + z11) / 4;
  if (NORMALIZE) n = normalized3(n);
  out.at(y, x) = n.x;
  out.at(y + drows, x) = n.y;
  out.at(y + 2 * drows, x) = n.z;
}

struct OpFloatHalf {  // dms_pyrDownGaussF (pyrDownGaussF, cudafuncs.cu:446-542)
  live::Pitched<float> src;
  View<float> dst;
  int scols, srows;
  __device__ void operator()(int x, int y) const { dst.at(y, x) = live::float_half(src, x, y, scols, srows); }
};
struct OpU8Half {  // dms_pyrDownUcharGauss (pyrDownUcharGauss, cudafuncs.cu:575-595)
  live::Pitched<unsigned char> src;
  View<unsigned char> dst;
  int scols, srows;
  __device__ void operator()(int x, int y) const { dst.at(y, x) = live::u8_half(src, x, y, scols, srows); }
};

// ---------------------------------------------------------------------------------------
// Fused model-side pyramid of the frame step (initICPModel + initRGBModel of ElasticFusion.cpp:172-189
// = select x3, copyMaps, resizeMap x4, transformMaps x3, verticesToDepth, pyrDownGaussF x2,
// bgr2Intensity, pyrDownUchar x2: 17 launches of the operator layer) in four launches.  Every value
// is computed by the same expressions as the stand-alone kernels above, and — like them — a pixel
// whose x plane is NaN gets only that NaN written (its y / z planes are never read downstream).
// ---------------------------------------------------------------------------------------
// Level 0 and levels 1 + 2 read the same sources and do not depend on each other: one launch, the first
// g0x * g0y blocks are level 0's 2-D grid, the rest levels 1 + 2's (a launch boundary and the short second
// kernel's ramp are saved; the small grid runs in the shadow of the large one).  A third group of blocks takes
// the first step of the depth / intensity pyramid directly from the sources.
__global__ void k_model_levels012(ModelSrc m, int g0x, int g0y, int g12x, int rows0, int cols0, View<float> v0, View<float> n0,
                                  View<float> depth0, View<unsigned char> inten0, float cutOff, int rows1, int cols1, int rows2, int cols2,
                                  View<float> v1, View<float> n1, View<float> v2, View<float> n2, int g12y, int gsx, View<float> depth1,
                                  View<unsigned char> inten1, TrackInitArgs ti) {
  // A fourth group, first in the grid (block 0's one-lane state set-up is the longest dependency chain of the launch): the
  // set-up of the tracker call that follows (track_init.hpp) — it depends on the prior pose only, not on this pyramid.
  if ((int)blockIdx.x < ti.blocks) {
    track_init_body((int)blockIdx.x, ti.blocks, (int)(threadIdx.y * blockDim.x + threadIdx.x), (int)(blockDim.x * blockDim.y), ti.st, ti.prior,
                    ti.prior_pose16, ti.fx, ti.fy, ti.cx, ti.cy, ti.so3, ti.first_level, ti.sync_words, ti.n_sync, ti.inject_timeout, nullptr);
    return;
  }
  // (the two small groups come first in the grid: their threads carry 32 / 50 dependent-latency loads each and would
  // otherwise start only when the 1 200 level-0 blocks have been handed out — the launch's tail)
  const int b = (int)blockIdx.x - ti.blocks;
  if (m.dense_cnt && b == 0 && threadIdx.x == 0 && threadIdx.y == 0) *m.flag_out = model_use_b(m) ? 1 : 0;
  const int nb0 = g0x * g0y, nb12 = g12x * g12y, nbs = (int)gridDim.x - ti.blocks - nb0 - nb12;
  if (b < nb12) {
    model_levels12_body((int)threadIdx.x, (int)threadIdx.y, (int)blockDim.y, b % g12x, b / g12x, m, cols0, rows1, cols1, rows2, cols2, v1, n1, v2, n2);
  } else if (b < nb12 + nbs) {
    const int c = b - nb12;
    model_pyr_step1_body<BY>((int)threadIdx.x, (int)threadIdx.y, c % gsx, c / gsx, m, rows0, cols0, cutOff, depth1, inten1);
  } else {
    const int c = b - nb12 - nbs;
    model_level0_body((int)threadIdx.x, (int)threadIdx.y, (int)blockDim.y, c % g0x, c / g0x, m, rows0, cols0, v0, n0, depth0, inten0, cutOff);
  }
}

// one pyramid step of both model images: float depth (pyrDownGaussF) and intensity (pyrDownUchar)
__global__ void k_model_pyr_step(View<const float> dsrc, View<float> ddst, View<const unsigned char> isrc, View<unsigned char> idst) {
  model_pyr_step_pixel(blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y, dsrc, ddst, isrc, idst);
}

// reference verticesToDepthKernel / verticesToDepth2DKernel (cudafuncs.cu:597-630)
__global__ void k_verticesToDepth(const float4* __restrict__ vsrc, View<float> dst, float cutOff) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dst.cols || y >= dst.rows) return;
  const float z = vsrc[(size_t)y * dst.cols + x].z;
  dst.at(y, x) = (z > cutOff || z <= 0.f) ? qnan() : z;
}
__global__ void k_verticesToDepth2D(View<const float> vsrc, View<float> dst, float cutOff) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dst.cols || y >= dst.rows) return;
  const float z = vsrc.at(y + dst.rows * 2, x);
  dst.at(y, x) = (z > cutOff || z <= 0.f) ? qnan() : z;
}

struct OpIntensity {  // dms_imageBGRToIntensity
  View<const uchar4> src;
  View<unsigned char> dst;
  __device__ void operator()(int x, int y) const { dst.at(y, x) = live::intensity_of(src.at(y, x)); }
};
template <bool GATE>
struct OpGradient {  // dms_computeDerivativeImages; GATE: + the per-frame photometric gate of the tracker (live_bodies.hpp)
  live::Pitched<unsigned char> src;
  View<short> dx, dy;
  View<unsigned char> gate;
  int cols, rows;
  float minScale;
  __device__ void operator()(int x, int y) const {
    const live::Grad g = live::gradient_gate(src, x, y, cols, rows, minScale);
    dx.at(y, x) = g.dx;
    dy.at(y, x) = g.dy;
    if (GATE) gate.at(y, x) = g.gate;
  }
};

// ---------------------------------------------------------------------------------------
// Fused live half of the frame step.  What RGBDOdometry::initICP(filteredDepth) + initRGB(rgb) + the Sobel pass do in
// fifteen launches of the operator layer (pyrDown x2, createVMap x3, createNMap x3, bgr2Intensity, pyrDownUcharGauss x2,
// derivatives x3: RGBDOdometry.cpp:118-142, :244-266, :279-283) is here
//   * the depth filter's epilogue (fusion_pre.hip): level-0 depth + level-0 vertex map of the value it has just filtered;
//   * k_live_ingest (fusion_frame.hip): level-0 intensity of the texel it has just converted;
//   * ONE launch of k_live_levels: group A — level-0 normals and level-0 gradients / gate, one thread per pixel; group B —
//     everything of levels 1 and 2 (depth, vertex, normal, intensity, gradients, gate), one block per 16 x 16 level-1 tile,
//     computed from LDS tiles of the level-0 depth and intensity: each level-0 texel is read from memory once per block and
//     every intermediate level stays in LDS.
// The bodies are live_bodies.hpp's: the same bits as the operator chain (GPU test).
// ---------------------------------------------------------------------------------------
struct LiveLevel {
  unsigned short* depth;  // dense u16
  float* vmap;            // 3 stacked dense planes
  float* nmap;
  unsigned char* image;   // dense u8
  short* dx;
  short* dy;
  unsigned char* gate;
  int cols, rows;
  live::LevelCam cam;
  float minScale;
};
struct LiveArgs {
  LiveLevel L[3];
  float cutoff;
  int gAx, gAy;  // group A: 64 x 4 tiles of level 0
  int gBx, gBy;  // group B: 16 x 16 tiles of level 1
};

__device__ __forceinline__ void store3(float* base, int rows, int cols, int y, int x, const f3& p) {
  const size_t i = (size_t)y * cols + x, plane = (size_t)rows * cols;
  base[i] = p.x;
  if (!isnan(p.x)) {
    base[plane + i] = p.y;
    base[2 * plane + i] = p.z;
  }
}

// group A: level 0, one thread per pixel.  The normal needs the vertices of the pixel, its right and its lower neighbour:
// re-derived from the level-0 depth (the expression of the filter's epilogue), so nothing waits for the vertex map.
__device__ __forceinline__ void live_level0_pixel(const LiveArgs& a, int x, int y) {
  const LiveLevel& l = a.L[0];
  if (x >= l.cols || y >= l.rows) return;
  const live::Pitched<unsigned short> d = {l.depth, (unsigned)l.cols * 2u};
  const bool border = x == l.cols - 1 || y == l.rows - 1;
  const f3 nan3 = mk3(qnan(), 0.f, 0.f);
  const f3 here = live::vertex_of(d(y, x), x, y, l.cam, a.cutoff);
  const f3 right = border ? nan3 : live::vertex_of(d(y, x + 1), x + 1, y, l.cam, a.cutoff);
  const f3 below = border ? nan3 : live::vertex_of(d(y + 1, x), x, y + 1, l.cam, a.cutoff);
  store3(l.nmap, l.rows, l.cols, y, x, live::normal_of(here, right, below, border));
  const live::Pitched<unsigned char> img = {l.image, (unsigned)l.cols};
  const live::Grad g = live::gradient_gate(img, x, y, l.cols, l.rows, l.minScale);
  const size_t i = (size_t)y * l.cols + x;
  l.dx[i] = g.dx;
  l.dy[i] = g.dy;
  l.gate[i] = g.gate;
}

// group B tile geometry (level-1 tile of T1 x T1 pixels at (X1, Y1) = (16 bx, 16 by); level-2 tile T2 x T2 at (X1 / 2, Y1 / 2)):
//   level-2 depth      [X2, X2 + T2]            (+1: forward difference of the normal)                 origin X2      width T2 + 1
//   level-2 intensity  [X2 - 2, X2 + T2 + 1]    (gate window -2 .. +1, gradient -1 .. +1)              origin X2 - 2  width T2 + 4
//   level-1 depth      [X1 - 2, X1 + T1 + 2]    (own + 1, and the 5 x 5 sources of level-2 depth)      origin X1 - 2  width T1 + 5
//   level-1 intensity  [X1 - 6, X1 + T1 + 4]    (own windows, and the sources of level-2 intensity)    origin X1 - 6  width T1 + 11
//   level-0 depth      [2 X1 - 6, 2 X1 + 2 T1 + 6]                                                      origin 2 X1 - 6   width 2 T1 + 13
//   level-0 intensity  [2 X1 - 14, 2 X1 + 2 T1 + 10]                                                    origin 2 X1 - 14  width 2 T1 + 25
constexpr int kT1 = 16, kT2 = 8;
constexpr int kD2W = kT2 + 1, kI2W = kT2 + 4, kD1W = kT1 + 5, kI1W = kT1 + 11, kD0W = 2 * kT1 + 13, kI0W = 2 * kT1 + 25;

__device__ __forceinline__ void live_levels12_tile(const LiveArgs& a, int bx, int by) {
  __shared__ unsigned short s_d0[kD0W * kD0W], s_d1[kD1W * kD1W], s_d2[kD2W * kD2W];
  __shared__ unsigned char s_i0[kI0W * kI0W], s_i1[kI1W * kI1W], s_i2[kI2W * kI2W];
  const LiveLevel &l0 = a.L[0], &l1 = a.L[1], &l2 = a.L[2];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
  const int X1 = bx * kT1, Y1 = by * kT1, X2 = X1 / 2, Y2 = Y1 / 2;
  // ---- level-0 tiles from memory (texels outside the image are never read by a body: zero) ----
  const int d0x = 2 * X1 - 6, d0y = 2 * Y1 - 6, i0x = 2 * X1 - 14, i0y = 2 * Y1 - 14;
  for (int e = tid; e < kD0W * kD0W; e += nt) {
    const int r = e / kD0W, c = e - r * kD0W, gy = d0y + r, gx = d0x + c;
    s_d0[e] = (gx >= 0 && gy >= 0 && gx < l0.cols && gy < l0.rows) ? l0.depth[(size_t)gy * l0.cols + gx] : (unsigned short)0;
  }
  for (int e = tid; e < kI0W * kI0W; e += nt) {
    const int r = e / kI0W, c = e - r * kI0W, gy = i0y + r, gx = i0x + c;
    s_i0[e] = (gx >= 0 && gy >= 0 && gx < l0.cols && gy < l0.rows) ? l0.image[(size_t)gy * l0.cols + gx] : (unsigned char)0;
  }
  __syncthreads();
  // ---- level 1 into LDS ----
  const live::Tile<unsigned short, kD0W> td0 = {s_d0, d0y, d0x};
  const live::Tile<unsigned char, kI0W> ti0 = {s_i0, i0y, i0x};
  const int d1x = X1 - 2, d1y = Y1 - 2, i1x = X1 - 6, i1y = Y1 - 6;
  for (int e = tid; e < kD1W * kD1W; e += nt) {
    const int r = e / kD1W, c = e - r * kD1W, gy = d1y + r, gx = d1x + c;
    s_d1[e] = (gx >= 0 && gy >= 0 && gx < l1.cols && gy < l1.rows) ? live::depth_half(td0, gx, gy, l0.cols, l0.rows) : (unsigned short)0;
  }
  for (int e = tid; e < kI1W * kI1W; e += nt) {
    const int r = e / kI1W, c = e - r * kI1W, gy = i1y + r, gx = i1x + c;
    s_i1[e] = (gx >= 0 && gy >= 0 && gx < l1.cols && gy < l1.rows) ? live::u8_half(ti0, gx, gy, l0.cols, l0.rows) : (unsigned char)0;
  }
  __syncthreads();
  // ---- level 2 into LDS ----
  const live::Tile<unsigned short, kD1W> td1 = {s_d1, d1y, d1x};
  const live::Tile<unsigned char, kI1W> ti1 = {s_i1, i1y, i1x};
  const int i2x = X2 - 2, i2y = Y2 - 2;
  for (int e = tid; e < kD2W * kD2W; e += nt) {
    const int r = e / kD2W, c = e - r * kD2W, gy = Y2 + r, gx = X2 + c;
    s_d2[e] = (gx < l2.cols && gy < l2.rows) ? live::depth_half(td1, gx, gy, l1.cols, l1.rows) : (unsigned short)0;
  }
  for (int e = tid; e < kI2W * kI2W; e += nt) {
    const int r = e / kI2W, c = e - r * kI2W, gy = i2y + r, gx = i2x + c;
    s_i2[e] = (gx >= 0 && gy >= 0 && gx < l2.cols && gy < l2.rows) ? live::u8_half(ti1, gx, gy, l1.cols, l1.rows) : (unsigned char)0;
  }
  __syncthreads();
  const live::Tile<unsigned short, kD2W> td2 = {s_d2, Y2, X2};
  const live::Tile<unsigned char, kI2W> ti2 = {s_i2, i2y, i2x};
  // ---- outputs: the block's own level-1 pixels (one per thread) and level-2 pixels (first 64 threads) ----
  auto emit = [&](const LiveLevel& l, auto depth_tile, auto image_tile, int x, int y) {
    if (x >= l.cols || y >= l.rows) return;
    const size_t i = (size_t)y * l.cols + x;
    const unsigned short d = depth_tile(y, x);
    l.depth[i] = d;
    l.image[i] = image_tile(y, x);
    const bool border = x == l.cols - 1 || y == l.rows - 1;
    const f3 nan3 = mk3(qnan(), 0.f, 0.f);
    const f3 here = live::vertex_of(d, x, y, l.cam, a.cutoff);
    const f3 right = border ? nan3 : live::vertex_of(depth_tile(y, x + 1), x + 1, y, l.cam, a.cutoff);
    const f3 below = border ? nan3 : live::vertex_of(depth_tile(y + 1, x), x, y + 1, l.cam, a.cutoff);
    store3(l.vmap, l.rows, l.cols, y, x, here);
    store3(l.nmap, l.rows, l.cols, y, x, live::normal_of(here, right, below, border));
    const live::Grad g = live::gradient_gate(image_tile, x, y, l.cols, l.rows, l.minScale);
    l.dx[i] = g.dx;
    l.dy[i] = g.dy;
    l.gate[i] = g.gate;
  };
  emit(l1, td1, ti1, X1 + (tid & (kT1 - 1)), Y1 + (tid >> 4));
  if (tid < kT2 * kT2) emit(l2, td2, ti2, X2 + (tid & (kT2 - 1)), Y2 + (tid >> 3));
}

__global__ __launch_bounds__(256) void k_live_levels(LiveArgs a) {
  const int b = blockIdx.x, nB = a.gBx * a.gBy;
  if (b < nB) {  // (the tile blocks first: each is a longer dependency chain than a level-0 block)
    live_levels12_tile(a, b % a.gBx, b / a.gBx);
  } else {
    const int c = b - nB;
    live_level0_pixel(a, (c % a.gAx) * 64 + threadIdx.x, (c / a.gAx) * 4 + threadIdx.y);
  }
}

// depth / vmap / nmap / image / dx / dy / gate: the three pyramid levels of the tracker's live buffers (level-0 depth, vertex map
// and intensity already written by the depth filter's epilogue and the ingest kernel)
int liveLevelsFused(const dms_image2d* depth, const dms_image2d* vmap, const dms_image2d* nmap, const dms_image2d* image, const dms_image2d* dx,
                    const dms_image2d* dy, const dms_image2d* gate, const dms_camera* cam0, float cutoff, const float* minScale, hipStream_t s) {
  LiveArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < 3; ++l) {
    const int rows = depth[l].rows, cols = depth[l].cols;
    DMS_REQUIRE(depth[l].pitch == (size_t)cols * 2 && vmap[l].pitch == (size_t)cols * 4 && nmap[l].pitch == (size_t)cols * 4 &&
                    image[l].pitch == (size_t)cols && dx[l].pitch == (size_t)cols * 2 && dy[l].pitch == (size_t)cols * 2 &&
                    gate[l].pitch == (size_t)cols,
                "the fused live half needs dense images");
    DMS_REQUIRE(l == 0 || (rows == depth[l - 1].rows / 2 && cols == depth[l - 1].cols / 2), "pyramid shape");
    LiveLevel& L = a.L[l];
    L.depth = (unsigned short*)depth[l].data;
    L.vmap = (float*)vmap[l].data;
    L.nmap = (float*)nmap[l].data;
    L.image = (unsigned char*)image[l].data;
    L.dx = (short*)dx[l].data;
    L.dy = (short*)dy[l].data;
    L.gate = (unsigned char*)gate[l].data;
    L.cols = cols;
    L.rows = rows;
    const float div = (float)(1 << l);
    L.cam.fx_inv = 1.f / (cam0->fx / div);
    L.cam.fy_inv = 1.f / (cam0->fy / div);
    L.cam.cx = cam0->cx / div;
    L.cam.cy = cam0->cy / div;
    L.minScale = minScale[l];
  }
  a.cutoff = cutoff;
  a.gAx = (a.L[0].cols + 63) / 64;
  a.gAy = (a.L[0].rows + 3) / 4;
  a.gBx = (a.L[1].cols + kT1 - 1) / kT1;
  a.gBy = (a.L[1].rows + kT1 - 1) / kT1;
  hipLaunchKernelGGL(k_live_levels, dim3(a.gBx * a.gBy + a.gAx * a.gAy), dim3(64, 4), 0, s, a);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// reference projectPointsKernel (cudafuncs.cu:727-741); cloud is packed float3
__global__ void k_projectPoints(View<const float> depth, View<float> cloud3, float invFx, float invFy, float cx, float cy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= depth.cols || y >= depth.rows) return;
  const float z = depth.at(y, x);
  float* c = cloud3.row(y) + 3 * x;
  c[0] = (((float)x - cx) * z) * invFx;
  c[1] = (((float)y - cy) * z) * invFy;
  c[2] = z;
}

// ---------------------------------------------------------------------------------------
// host launchers (internal C++ API; the C ABI wrappers live in capi.hip)
// ---------------------------------------------------------------------------------------
#define LAUNCH2D(kern, cols, rows, stream, ...)                                  \
  do {                                                                           \
    dim3 b = blk();                                                              \
    dim3 g = grid2d((cols), (rows), b);                                          \
    hipLaunchKernelGGL(kern, g, b, 0, (hipStream_t)(stream), __VA_ARGS__);       \
    DMS_CHECK_LAUNCH();                                                          \
  } while (0)

int pyrDown(const dms_image2d* src, dms_image2d* dst, hipStream_t s) {
  DMS_REQUIRE(src && dst && src->data && dst->data, "null image");
  DMS_REQUIRE(dst->rows == src->rows / 2 && dst->cols == src->cols / 2, "dst must be src/2");
  const OpDepthHalf op = {{(const unsigned short*)src->data, (unsigned)src->pitch}, view<unsigned short>(dst), src->cols, src->rows};
  LAUNCH2D(k_per_pixel<OpDepthHalf>, dst->cols, dst->rows, s, dst->cols, dst->rows, op);
  return DMS_OK;
}

int createVMap(const dms_camera* intr, const dms_image2d* depth, dms_image2d* vmap, float cutoff, hipStream_t s) {
  DMS_REQUIRE(intr && depth && vmap && depth->data && vmap->data, "null argument");
  DMS_REQUIRE(vmap->rows == depth->rows * 3 && vmap->cols == depth->cols, "vmap must be (3*rows) x cols");
  const OpVertexMap op = {{(const unsigned short*)depth->data, (unsigned)depth->pitch}, view<float>(vmap),
                          {1.f / intr->fx, 1.f / intr->fy, intr->cx, intr->cy}, cutoff, depth->rows};
  LAUNCH2D(k_per_pixel<OpVertexMap>, depth->cols, depth->rows, s, depth->cols, depth->rows, op);
  return DMS_OK;
}

int createNMap(const dms_image2d* vmap, dms_image2d* nmap, hipStream_t s) {
  DMS_REQUIRE(vmap && nmap && vmap->data && nmap->data, "null argument");
  DMS_REQUIRE(vmap->rows == nmap->rows && vmap->cols == nmap->cols && vmap->rows % 3 == 0, "shape mismatch");
  const int rows = vmap->rows / 3, cols = vmap->cols;
  const OpNormalMap op = {view<const float>(vmap), view<float>(nmap), rows, cols};
  LAUNCH2D(k_per_pixel<OpNormalMap>, cols, rows, s, cols, rows, op);
  return DMS_OK;
}

int transformMaps(const dms_image2d* vs, const dms_image2d* ns, const dms_mat33* R, const dms_float3* t, dms_image2d* vd,
                  dms_image2d* nd, hipStream_t s) {
  DMS_REQUIRE(vs && R && t && vd && vs->data && vd->data, "null argument");
  DMS_REQUIRE(vs->rows % 3 == 0 && vd->rows == vs->rows && vd->cols == vs->cols, "shape mismatch");
  const int rows = vs->rows / 3, cols = vs->cols;
  if (ns) {
    DMS_REQUIRE(nd && ns->data && nd->data && ns->rows == vs->rows && nd->rows == vs->rows, "normal map shape mismatch");
    LAUNCH2D(k_transformMaps<true>, cols, rows, s, rows, cols, view<const float>(vs), view<const float>(ns), to_m33(R), to_f3(t),
             view<float>(vd), view<float>(nd));
  } else {
    LAUNCH2D(k_transformMaps<false>, cols, rows, s, rows, cols, view<const float>(vs), view<const float>(vs), to_m33(R), to_f3(t),
             view<float>(vd), view<float>(vd));
  }
  return DMS_OK;
}

int transformMapsDev(dms_image2d* v, dms_image2d* n, const float* pose16_dev, hipStream_t s) {
  DMS_REQUIRE(v && n && pose16_dev && v->data && n->data && v->rows % 3 == 0 && n->rows == v->rows, "bad argument");
  const int rows = v->rows / 3, cols = v->cols;
  LAUNCH2D(k_transformMaps_dev, cols, rows, s, rows, cols, view<float>(v), view<float>(n), pose16_dev);
  return DMS_OK;
}

int selectCopy16(void* dst, const void* a, const void* b, const int* flag_dev, int force_b, size_t n16, hipStream_t s) {
  DMS_REQUIRE(dst && a && b, "null argument");
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_select_copy16, dim3((unsigned)blocks), dim3(256), 0, s, (float4*)dst, (const float4*)a, (const float4*)b, flag_dev,
                     force_b, n16);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int copyMaps(const float* vsrc, const float* nsrc, dms_image2d* vd, dms_image2d* nd, hipStream_t s) {
  DMS_REQUIRE(vsrc && vd && vd->data && vd->rows % 3 == 0, "null argument");
  const int rows = vd->rows / 3, cols = vd->cols;
  if (nsrc) {
    DMS_REQUIRE(nd && nd->data && nd->rows == vd->rows && nd->cols == vd->cols, "normal map shape mismatch");
    LAUNCH2D(k_copyMaps<true>, cols, rows, s, rows, cols, (const float4*)vsrc, (const float4*)nsrc, view<float>(vd), view<float>(nd));
  } else {
    LAUNCH2D(k_copyMaps<false>, cols, rows, s, rows, cols, (const float4*)vsrc, (const float4*)vsrc, view<float>(vd), view<float>(vd));
  }
  return DMS_OK;
}

int resizeMap(const dms_image2d* in, dms_image2d* out, bool normalize, hipStream_t s) {
  DMS_REQUIRE(in && out && in->data && out->data && in->rows % 3 == 0, "null argument");
  const int in_rows = in->rows / 3, out_rows = in_rows / 2, out_cols = in->cols / 2;
  DMS_REQUIRE(out->rows == out_rows * 3 && out->cols == out_cols, "output must be input/2");
  if (normalize)
    LAUNCH2D(k_resizeMap<true>, out_cols, out_rows, s, out_rows, out_cols, in_rows, view<const float>(in), view<float>(out));
  else
    LAUNCH2D(k_resizeMap<false>, out_cols, out_rows, s, out_rows, out_cols, in_rows, view<const float>(in), view<float>(out));
  return DMS_OK;
}

int pyrDownGaussF(const dms_image2d* src, dms_image2d* dst, hipStream_t s) {
  DMS_REQUIRE(src && dst && src->data && dst->data, "null image");
  DMS_REQUIRE(dst->rows == src->rows / 2 && dst->cols == src->cols / 2, "dst must be src/2");
  const OpFloatHalf op = {{(const float*)src->data, (unsigned)src->pitch}, view<float>(dst), src->cols, src->rows};
  LAUNCH2D(k_per_pixel<OpFloatHalf>, dst->cols, dst->rows, s, dst->cols, dst->rows, op);
  return DMS_OK;
}

int pyrDownUcharGauss(const dms_image2d* src, dms_image2d* dst, hipStream_t s) {
  DMS_REQUIRE(src && dst && src->data && dst->data, "null image");
  DMS_REQUIRE(dst->rows == src->rows / 2 && dst->cols == src->cols / 2, "dst must be src/2");
  const OpU8Half op = {{(const unsigned char*)src->data, (unsigned)src->pitch}, view<unsigned char>(dst), src->cols, src->rows};
  LAUNCH2D(k_per_pixel<OpU8Half>, dst->cols, dst->rows, s, dst->cols, dst->rows, op);
  return DMS_OK;
}

int modelPyramidFused(const void* vA, const void* nA, const void* iA, const void* vB, const void* nB, const void* iB, const int* flag_dev,
                      int force_b_img, const float* pose16_dev, dms_image2d* vmaps, dms_image2d* nmaps, dms_image2d* depths,
                      dms_image2d* images, float cutOff, hipStream_t s, bool skip_last_step, const unsigned* dense_cnt, int dense_samples,
                      int* flag_out, const TrackInitArgs* init) {
  DMS_REQUIRE(vA && nA && iA && vB && nB && iB && flag_dev && vmaps && nmaps && depths && images, "null argument");
  ModelSrc m;
  m.vA = (const float4*)vA;
  m.nA = (const float4*)nA;
  m.iA = (const uchar4*)iA;
  m.vB = (const float4*)vB;
  m.nB = (const float4*)nB;
  m.iB = (const uchar4*)iB;
  m.flag = flag_dev;
  m.dense_cnt = dense_cnt;
  m.dense_samples = dense_samples;
  m.flag_out = flag_out;
  DMS_REQUIRE(!dense_cnt || (flag_out && dense_samples > 0), "dense counters need a flag destination");
  m.force_b_img = force_b_img;
  m.pose16 = pose16_dev;
  const int rows0 = vmaps[0].rows / 3, cols0 = vmaps[0].cols;
  const int rows1 = vmaps[1].rows / 3, cols1 = vmaps[1].cols, rows2 = vmaps[2].rows / 3, cols2 = vmaps[2].cols;
  DMS_REQUIRE(rows1 == rows0 / 2 && cols1 == cols0 / 2 && rows2 == rows1 / 2 && cols2 == cols1 / 2, "pyramid shapes");
  {
    const dim3 b = blk();
    const dim3 g0 = grid2d(cols0, rows0, b), g12 = dim3(((cols1 + 1) / 2 + 15) / 16, ((rows1 + 1) / 2 + b.y - 1) / b.y),  // 16 x BY level-2 pixels per block
               gs = grid2d(depths[1].cols, depths[1].rows, b);
    TrackInitArgs ti;
    memset(&ti, 0, sizeof(ti));
    if (init) ti = *init;
    hipLaunchKernelGGL(k_model_levels012, dim3(ti.blocks + g0.x * g0.y + g12.x * g12.y + gs.x * gs.y), b, 0, s, m, (int)g0.x, (int)g0.y, (int)g12.x, rows0,
                       cols0, view<float>(&vmaps[0]), view<float>(&nmaps[0]), view<float>(&depths[0]), view<unsigned char>(&images[0]), cutOff,
                       rows1, cols1, rows2, cols2, view<float>(&vmaps[1]), view<float>(&nmaps[1]), view<float>(&vmaps[2]),
                       view<float>(&nmaps[2]), (int)g12.y, (int)gs.x, view<float>(&depths[1]), view<unsigned char>(&images[1]), ti);
    DMS_CHECK_LAUNCH();
  }
  for (int l = 2; l < 3 && !skip_last_step; ++l)  // (skipped: the caller runs that step inside a kernel of its own)
    LAUNCH2D(k_model_pyr_step, depths[l].cols, depths[l].rows, s, view<const float>(&depths[l - 1]), view<float>(&depths[l]),
             view<const unsigned char>(&images[l - 1]), view<unsigned char>(&images[l]));
  return DMS_OK;
}

int verticesToDepth(const float* vsrc, dms_image2d* dst, float cutOff, hipStream_t s) {
  DMS_REQUIRE(vsrc && dst && dst->data, "null argument");
  LAUNCH2D(k_verticesToDepth, dst->cols, dst->rows, s, (const float4*)vsrc, view<float>(dst), cutOff);
  return DMS_OK;
}

int verticesToDepth2D(const dms_image2d* vsrc, dms_image2d* dst, float cutOff, hipStream_t s) {
  DMS_REQUIRE(vsrc && dst && vsrc->data && dst->data && vsrc->rows == 3 * dst->rows, "shape mismatch");
  LAUNCH2D(k_verticesToDepth2D, dst->cols, dst->rows, s, view<const float>(vsrc), view<float>(dst), cutOff);
  return DMS_OK;
}

int imageToIntensity(const dms_image2d* rgba, dms_image2d* dst, hipStream_t s) {
  DMS_REQUIRE(rgba && dst && rgba->data && dst->data && rgba->rows == dst->rows && rgba->cols == dst->cols, "shape mismatch");
  const OpIntensity op = {view<const uchar4>(rgba), view<unsigned char>(dst)};
  LAUNCH2D(k_per_pixel<OpIntensity>, dst->cols, dst->rows, s, dst->cols, dst->rows, op);
  return DMS_OK;
}

int derivativeImages(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy, hipStream_t s) {
  DMS_REQUIRE(src && dx && dy && src->data && dx->data && dy->data, "null argument");
  DMS_REQUIRE(dx->rows == src->rows && dx->cols == src->cols && dy->rows == src->rows && dy->cols == src->cols, "shape mismatch");
  const OpGradient<false> op = {{(const unsigned char*)src->data, (unsigned)src->pitch}, view<short>(dx), view<short>(dy), view<unsigned char>(src), src->cols, src->rows, 0.f};
  LAUNCH2D(k_per_pixel<OpGradient<false>>, src->cols, src->rows, s, src->cols, src->rows, op);
  return DMS_OK;
}

int derivativeGate(const dms_image2d* src, dms_image2d* dx, dms_image2d* dy, dms_image2d* gate, float minScale, hipStream_t s) {
  DMS_REQUIRE(src && dx && dy && gate && src->data && dx->data && dy->data && gate->data, "null argument");
  DMS_REQUIRE(dx->rows == src->rows && dx->cols == src->cols && dy->rows == src->rows && dy->cols == src->cols && gate->rows == src->rows &&
                  gate->cols == src->cols,
              "shape mismatch");
  const OpGradient<true> op = {{(const unsigned char*)src->data, (unsigned)src->pitch}, view<short>(dx), view<short>(dy), view<unsigned char>(gate), src->cols, src->rows, minScale};
  LAUNCH2D(k_per_pixel<OpGradient<true>>, src->cols, src->rows, s, src->cols, src->rows, op);
  return DMS_OK;
}

int projectToPointCloud(const dms_image2d* depth, dms_image2d* cloud, const dms_camera* intr, int level, hipStream_t s) {
  DMS_REQUIRE(depth && cloud && intr && depth->data && cloud->data, "null argument");
  DMS_REQUIRE(cloud->rows == depth->rows && cloud->cols == depth->cols, "shape mismatch");
  const int div = 1 << level;
  const float fx = intr->fx / div, fy = intr->fy / div, cx = intr->cx / div, cy = intr->cy / div;
  LAUNCH2D(k_projectPoints, depth->cols, depth->rows, s, view<const float>(depth), view<float>(cloud), 1.0f / fx, 1.0f / fy, cx, cy);
  return DMS_OK;
}

}  // namespace dms
