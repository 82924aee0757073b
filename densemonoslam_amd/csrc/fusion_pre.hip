// Per-frame image stages of the fusion half: depth bilateral filter (G1), metric depth (G2),
// hole fill-in of the model prediction (G10) and nearest-neighbour resize (G11).
// All are one-thread-per-pixel streaming kernels over dense row-major images.
#include "surfel.hpp"
#include "fill.hpp"
#include "live_bodies.hpp"

#include <mutex>

namespace dms {

static constexpr int BX = 64, BY = 4;

// G1 — depth_bilateral.frag:30-75.  Fragment (x, y) has texcoord ((x+0.5)/cols, (y+0.5)/rows);
// taps are fetched at (float(cx)/cols, float(cy)/rows) with the NEAREST rule of surfel.hpp.
// The weight exp(-(space2 / 2 sigma_s^2 + dv^2 / 2 sigma_c^2)) of a tap depends on (|dx|, |dy|) in 0..6 — through dx^2 + dy^2, so on
// the unordered pair: 28 rows — and on the integer depth difference |dv| in millimetres, and is exactly 0 from |dv| = 396 on (the
// argument falls below det_expf's -87 cut-off whatever the distance).  It is tabulated ONCE per device with the very expression of
// the shader (k_bilateral_lut_build: 28 x 397 floats, the last column the zero every larger difference is clamped to); each block
// copies the table to LDS (44 KB) and stages its tile's source texels (halo included, through the per-column / per-row tap -> texel
// tables, one fp32 division + floor each), so a tap is two LDS reads and nine vector instructions, no branch: the thirteen taps of
// a window row are read together, then their thirteen weights — two LDS round trips per row.  The 169-tap sum keeps the shader's
// order (rows outer, columns inner) and arithmetic, so the bits are the per-tap evaluation's.
// TBY rows per tile.  The kernel walks tiles with a block stride, so the launch decides its footprint: one 64 x 10 tile per block
// (two blocks = 20 waves per compute unit; 480 tiles at 640 x 480 are resident at once) or a few fat 64 x 16 blocks that own the same
// few compute units for the whole image.  The frame step uses the second form on its prep stream: the previous frame's tracker runs
// beside it, and its resident kernels need one EMPTY compute unit per block — with small blocks sprinkled over every CU each tracker
// launch waited for CUs to drain (+40 us on the level-0 kernel and +16 us on level 1 in the rocprof trace of round 2).
constexpr int kBilDv = 396, kBilStride = kBilDv + 1, kBilRows = 28;
__host__ __device__ constexpr int bil_row(int a, int b) { return a < b ? b * (b + 1) / 2 + a : a * (a + 1) / 2 + b; }
__device__ float g_bil_lut[kBilRows * kBilStride];

__global__ void k_bilateral_lut_build() {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kBilRows * kBilStride; e += gridDim.x * blockDim.x) {
    const int row = e / kBilStride, m = e - row * kBilStride;
    int hi = 0;
    while ((hi + 1) * (hi + 2) / 2 <= row) ++hi;
    const int lo = row - hi * (hi + 1) / 2;
    const float space2 = (float)(lo * lo + hi * hi);
    const float dc = (float)m;
    const float color2 = dc * dc;
    g_bil_lut[e] = m < kBilDv ? det_expf(-(space2 * 0.024691358f + color2 * 0.000555556f)) : 0.f;
  }
}

// What else the frame step derives from a filtered depth value, per pixel, in the filter's own store (three launches and two
// more passes over the image otherwise): its metric form (metriciseDepth of the filtered image, ElasticFusion.cpp:119), the
// tracker's level-0 depth (the copy of RGBDOdometry::initICP, RGBDOdometry.cpp:122-128) and the level-0 vertex map
// (createVMap at level 0, :132-137).  All optional (null = skip).
struct BilateralEpilogue {
  float* metric_filtered;     // dense float image
  unsigned short* depth_l0;   // dense u16 image
  float* vmap_l0;             // three stacked dense planes of `rows` rows
  live::LevelCam cam;
  float vmap_cutoff;
};
__device__ __forceinline__ void bilateral_store(unsigned short* __restrict__ dst, const BilateralEpilogue& ep, int px, int py, int cols, int rows,
                                                unsigned short v, unsigned gate) {
  const size_t i = (size_t)py * cols + px;
  dst[i] = v;
  if (ep.metric_filtered) ep.metric_filtered[i] = ((unsigned)v > gate || (unsigned)v < 300U) ? 0.f : (float)v / 1000.0f;  // depth_metric.frag:28-39
  if (ep.depth_l0) ep.depth_l0[i] = v;
  if (ep.vmap_l0) {
    const f3 p = live::vertex_of(v, px, py, ep.cam, ep.vmap_cutoff);
    const size_t plane = (size_t)rows * cols;
    ep.vmap_l0[i] = p.x;
    if (!isnan(p.x)) {
      ep.vmap_l0[plane + i] = p.y;
      ep.vmap_l0[2 * plane + i] = p.z;
    }
  }
}

// the 13 x 13 window of one pixel from the staged tile, row by row: thirteen tile reads, thirteen table reads, then the two sums in
// the shader's order.  MASKED: taps outside the image take the table's zero column.
template <int TBY, bool MASKED>
__device__ __forceinline__ void bilateral_rows(const unsigned short (*tile)[BX + 2 * 8 + 2], const float* lut, unsigned value, int kx0, int ky0, int x0,
                                               int y0, int cols, int rows, float& sum1, float& sum2) {
  constexpr int R = 6, D = 2 * R + 1;
#pragma unroll 1
  for (int j = 0; j < D; ++j) {  // (rows stay a loop: 169 unrolled taps hoist more loads than the registers hold)
    const int ady = j < R ? R - j : j - R;
    const unsigned short* trow = &tile[ky0 + j][kx0];
    const bool row_in = !MASKED || (unsigned)(y0 + j) < (unsigned)rows;
    unsigned tap[D];
    float w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) tap[i] = trow[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int adx = i < R ? R - i : i - R;
      unsigned dv = min((unsigned)abs((int)value - (int)tap[i]), (unsigned)kBilDv);
      if (MASKED) dv = (row_in && (unsigned)(x0 + i) < (unsigned)cols) ? dv : (unsigned)kBilDv;
      w[i] = lut[bil_row(adx, ady) * kBilStride + dv];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      sum1 += (float)tap[i] * w[i];
      sum2 += w[i];
    }
  }
}

template <int TBY>
__global__ __launch_bounds__(BX* TBY) void k_depth_bilateral(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                             int cols, int rows, float maxD, BilateralEpilogue ep) {
  constexpr int HALO = 8, R = 6, D = 2 * R + 1;
  constexpr int TW = BX + 2 * HALO, TH = TBY + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float s_lut[kBilRows * kBilStride];
  __shared__ unsigned short s_tile[TH][TW + 2];
  __shared__ int s_sx[TW];
  __shared__ int s_sy[TH];
  const float colsf = (float)cols, rowsf = (float)rows;
  const int tiles_x = (cols + BX - 1) / BX, tiles_y = (rows + TBY - 1) / TBY;
  const int tid = threadIdx.y * BX + threadIdx.x;
  static_assert(kBilRows * kBilStride % 4 == 0, "the table is copied as float4");
  for (int e = tid; e < kBilRows * kBilStride / 4; e += BX * TBY)
    reinterpret_cast<float4*>(s_lut)[e] = reinterpret_cast<const float4*>(g_bil_lut)[e];
  const unsigned gate = (unsigned)f2i_rz(maxD * 1000.0f);
  for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int bx0 = txi * BX - HALO, by0 = tyi * TBY - HALO;
    __syncthreads();  // the tables of the previous tile are no longer read
    if (tid < TW) s_sx[tid] = texel((float)(bx0 + tid) / colsf, colsf, cols);
    if (tid >= 128 && tid < 128 + TH) s_sy[tid - 128] = texel((float)(by0 + tid - 128) / rowsf, rowsf, rows);
    __syncthreads();
    for (int e = tid; e < TH * TW; e += BX * TBY) {
      const int ky = e / TW, kx = e - ky * TW;
      s_tile[ky][kx] = src[(size_t)s_sy[ky] * cols + s_sx[kx]];
    }
    __syncthreads();
    const int px = txi * BX + threadIdx.x;
    const int py = tyi * TBY + threadIdx.y;
    if (px >= cols || py >= rows) continue;
    const unsigned value = src[(size_t)py * cols + px];
    if (value > gate || value < 300U) {
      bilateral_store(dst, ep, px, py, cols, rows, 0, gate);
      continue;
    }
    // int(texcoord * cols): texcoord of the fragment centre
    const float tcx = ((float)px + 0.5f) / colsf, tcy = ((float)py + 0.5f) / rowsf;
    const int x = (int)(tcx * colsf);
    const int y = (int)(tcy * rowsf);
    const int x0 = x - R, y0 = y - R;
    const int kx0 = x0 - bx0, ky0 = y0 - by0;
    float sum1 = 0.f, sum2 = 0.f;
    const bool fits = kx0 >= 0 && kx0 + D <= TW && ky0 >= 0 && ky0 + D <= TH;  // the window lies in the staged tile (always, in practice)
    const bool interior = x0 >= 0 && x0 + D <= cols && y0 >= 0 && y0 + D <= rows;
    // A window the image clips: the shader's loops skip the taps outside; here they get the table's zero column — a tap of weight
    // +0 adds +0 to both sums, which leaves them bit for bit — so a border pixel runs the same branch-free rows.  The choice is
    // per wave (one masked lane makes the wave take the masked rows: two compares + two selects more per tap).
    if (fits && __builtin_amdgcn_ballot_w64(!interior) == 0ull) {
      bilateral_rows<TBY, false>(s_tile, s_lut, value, kx0, ky0, x0, y0, cols, rows, sum1, sum2);
    } else if (fits) {
      bilateral_rows<TBY, true>(s_tile, s_lut, value, kx0, ky0, x0, y0, cols, rows, sum1, sum2);
    } else {
      // a fragment centre that rounds out of its tile (not seen; kept as the definition): the shader's loops, texels through the tables where they reach
      const int tx = min(x0 + D, cols), ty = min(y0 + D, rows);
      for (int cy = max(y0, 0); cy < ty; ++cy) {
        const int ky = cy - by0;
        const int sy = (ky >= 0 && ky < TH) ? s_sy[ky] : texel((float)cy / rowsf, rowsf, rows);
        const unsigned short* srow = src + (size_t)sy * cols;
        const int ady = abs(y - cy);
        for (int cx = max(x0, 0); cx < tx; ++cx) {
          const int kx = cx - bx0;
          const int sx = (kx >= 0 && kx < TW) ? s_sx[kx] : texel((float)cx / colsf, colsf, cols);
          const unsigned tap = srow[sx];
          const unsigned dv = min((unsigned)abs((int)value - (int)tap), (unsigned)kBilDv);
          const float weight = s_lut[bil_row(abs(x - cx), ady) * kBilStride + dv];
          sum1 += (float)tap * weight;
          sum2 += weight;
        }
      }
    }
    bilateral_store(dst, ep, px, py, cols, rows, (unsigned short)(unsigned)f2i_rz(roundf(sum1 / sum2)), gate);
  }
}

// G2 — depth_metric.frag:28-39
__global__ __launch_bounds__(BX* BY) void k_depth_metric(const unsigned short* __restrict__ src, float* __restrict__ dst, int n,
                                                         float maxD) {
  const unsigned gate = (unsigned)f2i_rz(maxD * 1000.0f);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const unsigned value = src[i];
    dst[i] = (value > gate || value < 300U) ? 0.f : (float)value / 1000.0f;
  }
}

// G10 — fill_vertex.frag / fill_normal.frag / fill_rgb.frag (per-pixel body: fill.hpp)
__global__ __launch_bounds__(BX* BY) void k_fill_in(FillArgs a) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  const int py = blockIdx.y * blockDim.y + threadIdx.y;
  if (a.dense_flag && (int)blockIdx.y == a.rows_blocks) {  // the extra block row: block 0 of it does the test, the others idle
    if (blockIdx.x != 0) return;
    __shared__ int s_sum[BX * BY / 64];
    fill_dense_test<BX * BY>(a, threadIdx.y * blockDim.x + threadIdx.x, s_sum);
    return;
  }
  if (a.mirror_words > 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < a.mirror_words) a.mirror_dst[t] = a.mirror_src[t];
  }
  if (px >= a.cols || py >= a.rows) return;
  const size_t i = (size_t)py * a.cols + px;
  fill_pixel(a, px, py, a.ex_vertex[i], a.ex_normal[i], a.ex_image[i]);
}

// G11 — resize.frag: dst pixel (i, j) samples src at ((i+0.5)/dw, (j+0.5)/dh), NEAREST
template <typename T>
__global__ void k_resize_nn(const T* __restrict__ src, int scols, int srows, T* __restrict__ dst, int dcols, int drows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= dcols || j >= drows) return;
  const float u = ((float)i + 0.5f) / (float)dcols, v = ((float)j + 0.5f) / (float)drows;
  const int sx = texel(u, (float)scols, scols), sy = texel(v, (float)srows, srows);
  dst[(size_t)j * dcols + i] = src[(size_t)sy * scols + sx];
}

static bool dense(const dms_image2d* im, size_t elem) { return im && im->data && im->pitch == (size_t)im->cols * elem; }

int depth_bilateral(const dms_image2d* src, dms_image2d* dst, float maxD, hipStream_t s, int narrow_blocks, const dms_image2d* metric_filtered,
                    const dms_image2d* depth_l0, const dms_image2d* vmap_l0, const dms_camera* cam_l0, float vmap_cutoff) {
  DMS_REQUIRE(dense(src, 2) && dense(dst, 2), "dense u16 images required");
  DMS_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, "shape mismatch");
  BilateralEpilogue ep;
  memset(&ep, 0, sizeof(ep));
  if (metric_filtered) {
    DMS_REQUIRE(dense(metric_filtered, 4) && metric_filtered->rows == src->rows && metric_filtered->cols == src->cols, "metric image shape");
    ep.metric_filtered = (float*)metric_filtered->data;
  }
  if (depth_l0) {
    DMS_REQUIRE(dense(depth_l0, 2) && depth_l0->rows == src->rows && depth_l0->cols == src->cols, "level-0 depth shape");
    ep.depth_l0 = (unsigned short*)depth_l0->data;
  }
  if (vmap_l0) {
    DMS_REQUIRE(cam_l0 && dense(vmap_l0, 4) && vmap_l0->rows == 3 * src->rows && vmap_l0->cols == src->cols, "level-0 vertex map shape");
    ep.vmap_l0 = (float*)vmap_l0->data;
    ep.cam.fx_inv = 1.f / cam_l0->fx;
    ep.cam.fy_inv = 1.f / cam_l0->fy;
    ep.cam.cx = cam_l0->cx;
    ep.cam.cy = cam_l0->cy;
    ep.vmap_cutoff = vmap_cutoff;
  }
  {  // the weight table, once per device (the first call waits for it, so that any stream may read it afterwards); marked built
     // only when the build succeeded: a failed first call is repeated by the next one instead of leaving a zero table behind
    static std::mutex mu;
    static bool built[64];
    int dev = 0;
    DMS_HIP(hipGetDevice(&dev));
    DMS_REQUIRE(dev >= 0 && dev < 64, "device index");
    std::lock_guard<std::mutex> lock(mu);
    if (!built[dev]) {
      hipLaunchKernelGGL(k_bilateral_lut_build, dim3(44), dim3(256), 0, s);
      DMS_HIP(hipGetLastError());
      DMS_HIP(hipStreamSynchronize(s));
      built[dev] = true;
    }
  }
  constexpr int WBY = 10;  // whole-chip form: 64 x 10 tiles, 640 threads, two blocks per compute unit (8 / 16 / 5 rows: 26 / 34 / 35 us against 24)
  if (narrow_blocks > 0) {  // a few 1 024-thread blocks that keep to their compute units (see the kernel)
    // ... and to themselves: the launch asks for the rest of the unit's 160 KB of LDS, so no other block — a resident tracker
    // block above all, which would then run at the pace of a shared unit and hold the whole grid's all-reduces back — is placed
    // beside a filter block (2100 against 2340 frames/s in the driver's form without it)
    // (the unit's LDS from the device, not from gfx950's data sheet; a part with less LDS than the static size just gets no pad)
    static const int pad = [] {
      hipFuncAttributes a;
      int dev = 0, lds_unit = 0, lds_block = 0;
      if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_depth_bilateral<16>)) != hipSuccess || hipGetDevice(&dev) != hipSuccess) return 0;
      if (hipDeviceGetAttribute(&lds_unit, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess) lds_unit = 0;
      if (hipDeviceGetAttribute(&lds_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) lds_block = 0;
      const int avail = lds_unit > lds_block ? lds_unit : lds_block;  // (what a block may ask for is at most the unit's)
      const int rest = avail - (int)a.sharedSizeBytes;
      return rest > 0 ? rest : 0;
    }();
    static bool pad_refused = false;
    hipLaunchKernelGGL((k_depth_bilateral<16>), dim3(narrow_blocks), dim3(BX, 16), pad_refused ? 0 : pad, s, (const unsigned short*)src->data,
                       (unsigned short*)dst->data, src->cols, src->rows, maxD, ep);
    if (!pad_refused && pad > 0 && hipGetLastError() != hipSuccess) {  // the pad is a placement hint, never a reason to fail
      pad_refused = true;
      hipLaunchKernelGGL((k_depth_bilateral<16>), dim3(narrow_blocks), dim3(BX, 16), 0, s, (const unsigned short*)src->data,
                         (unsigned short*)dst->data, src->cols, src->rows, maxD, ep);
    }
  } else {
    const int tiles = ((src->cols + BX - 1) / BX) * ((src->rows + WBY - 1) / WBY);
    hipLaunchKernelGGL((k_depth_bilateral<WBY>), dim3(tiles), dim3(BX, WBY), 0, s, (const unsigned short*)src->data,
                       (unsigned short*)dst->data, src->cols, src->rows, maxD, ep);
  }
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int depth_metric(const dms_image2d* src, dms_image2d* dst, float maxD, hipStream_t s) {
  DMS_REQUIRE(dense(src, 2) && dense(dst, 4), "dense images required");
  DMS_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, "shape mismatch");
  const int n = src->rows * src->cols;
  hipLaunchKernelGGL(k_depth_metric, dim3(min((n + 255) / 256, 2048)), dim3(256), 0, s, (const unsigned short*)src->data,
                     (float*)dst->data, n, maxD);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// the argument block of the fill-in for `ex` -> `out` (shared with the prediction's fused resolve pass)
int fill_args(const dms_predict_out* ex, const dms_image2d* depth, const dms_image2d* rgba, const dms_camera* cam, int pass_geom, int pass_rgb,
              dms_predict_out* out, const void* mirror_src, void* mirror_dst, int mirror_bytes, int* dense_flag, FillArgs* res) {
  DMS_REQUIRE(ex && depth && rgba && cam && out, "null argument");
  DMS_REQUIRE(dense(&ex->vertex, 16) && dense(&ex->normal, 16) && dense(&ex->image, 4) && dense(depth, 2) && dense(rgba, 4) &&
                  dense(&out->vertex, 16) && dense(&out->normal, 16) && dense(&out->image, 4),
              "dense images required");
  FillArgs a;
  a.ex_vertex = (const float4*)ex->vertex.data;
  a.ex_normal = (const float4*)ex->normal.data;
  a.ex_image = (const uchar4*)ex->image.data;
  a.depth = (const unsigned short*)depth->data;
  a.rgba = (const uchar4*)rgba->data;
  a.out_vertex = (float4*)out->vertex.data;
  a.out_normal = (float4*)out->normal.data;
  a.out_image = (uchar4*)out->image.data;
  a.cols = depth->cols;
  a.rows = depth->rows;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.ifx = 1.0f / cam->fx;
  a.ify = 1.0f / cam->fy;
  a.pass_geom = pass_geom ? 1 : 0;
  a.pass_rgb = pass_rgb ? 1 : 0;
  DMS_REQUIRE(mirror_bytes % 4 == 0 && mirror_bytes / 4 <= BX * BY, "mirror block too large");
  a.mirror_src = (const unsigned*)mirror_src;
  a.mirror_dst = (unsigned*)mirror_dst;
  a.mirror_words = (mirror_src && mirror_dst) ? mirror_bytes / 4 : 0;
  a.dense_flag = dense_flag;
  a.rows_blocks = 0;
  a.dense_cnt = nullptr;
  a.sample_mask = nullptr;
  a.thumb_block = nullptr;
  a.thumb_mask = nullptr;
  a.thumb_w = a.thumb_h = 0;
  a.thumb_pose_src = nullptr;
  a.thumb_pose_dst = nullptr;
  a.thumb_tick_dst = nullptr;
  a.thumb_tick = 0;
  *res = a;
  return DMS_OK;
}

int fill_in(const dms_predict_out* ex, const dms_image2d* depth, const dms_image2d* rgba, const dms_camera* cam, int pass_geom,
            int pass_rgb, dms_predict_out* out, hipStream_t s, const void* mirror_src, void* mirror_dst, int mirror_bytes, int* dense_flag) {
  FillArgs a;
  const int rc = fill_args(ex, depth, rgba, cam, pass_geom, pass_rgb, out, mirror_src, mirror_dst, mirror_bytes, dense_flag, &a);
  if (rc) return rc;
  dim3 b(BX, BY), g = grid2d(a.cols, a.rows, b);
  a.rows_blocks = (int)g.y;
  if (dense_flag) g.y += 1;
  hipLaunchKernelGGL(k_fill_in, g, b, 0, s, a);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int resize_nn(const dms_image2d* src, dms_image2d* dst, int elem, hipStream_t s) {
  DMS_REQUIRE(src && dst && src->data && dst->data, "null argument");
  DMS_REQUIRE(dense(src, elem) && dense(dst, elem), "dense images required");
  dim3 b(32, 8), g = grid2d(dst->cols, dst->rows, b);
  if (elem == 4)
    hipLaunchKernelGGL(k_resize_nn<unsigned>, g, b, 0, s, (const unsigned*)src->data, src->cols, src->rows, (unsigned*)dst->data,
                       dst->cols, dst->rows);
  else if (elem == 16)
    hipLaunchKernelGGL(k_resize_nn<float4>, g, b, 0, s, (const float4*)src->data, src->cols, src->rows, (float4*)dst->data, dst->cols,
                       dst->rows);
  else if (elem == 2)
    hipLaunchKernelGGL(k_resize_nn<unsigned short>, g, b, 0, s, (const unsigned short*)src->data, src->cols, src->rows,
                       (unsigned short*)dst->data, dst->cols, dst->rows);
  else
    DMS_REQUIRE(false, "elem_bytes must be 2, 4 or 16");
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

}  // namespace dms
