// Shared device/host helpers for the gfx950 kernels.  All per-element arithmetic is fp32
// with contraction OFF (the library is built with -ffp-contract=off) so that every
// multiply/add rounds exactly once, in the order written here; fmaf() is used only where
// named.  Division and sqrt are IEEE-correct (hipcc default).  That makes the per-pixel
// and per-surfel results a pure function of the inputs, which is what lets the parity
// tests demand integer-exact association.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dmslam.h"

namespace dms {

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves: one per SIMD of a CU
constexpr int kMaxPartialBlocks = DMS_MAX_PARTIAL_BLOCKS;
constexpr int kAutoPartialBlocks = 512;  // 2 blocks per CU: enough waves to cover HBM latency, half the records to fold
constexpr int kSE3 = 29;                 // 27 products + residual + inliers (types.cuh:123-171)
constexpr int kSO3 = 11;                 // 9 products + residual + inliers (types.cuh:173-197)

// ---- error plumbing --------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define DMS_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::dms::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define DMS_CHECK_LAUNCH() DMS_HIP(hipGetLastError())

#define DMS_REQUIRE(cond, msg)                    \
  do {                                            \
    if (!(cond)) {                                \
      ::dms::set_error("%s: %s", __func__, msg);  \
      return DMS_ERR_INVALID_ARG;                 \
    }                                             \
  } while (0)

// ---- pitched views ---------------------------------------------------------------------
template <typename T>
struct View {
  T* data;
  size_t pitch;  // bytes
  int rows, cols;
  __host__ __device__ __forceinline__ T* row(int y) const {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(data)) + (size_t)y * pitch);
  }
  __host__ __device__ __forceinline__ T& at(int y, int x) const { return row(y)[x]; }
};

template <typename T>
inline View<T> view(const dms_image2d* im) {
  View<T> v;
  v.data = reinterpret_cast<T*>(im->data);
  v.pitch = im->pitch;
  v.rows = im->rows;
  v.cols = im->cols;
  return v;
}

// ---- tiny vector math (evaluation order is part of the contract) ------------------------
struct f3 {
  float x, y, z;
};
struct M33 {
  f3 r0, r1, r2;
};

__host__ __device__ __forceinline__ f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
__host__ __device__ __forceinline__ f3 operator-(const f3& a, const f3& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ f3 operator+(const f3& a, const f3& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ float dot3(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ __forceinline__ f3 cross3(const f3& a, const f3& b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ __forceinline__ float norm3(const f3& a) { return sqrtf(dot3(a, a)); }
// normalise = v * (1 / sqrt(v.v)) — the reciprocal-square-root form of the reference
// (operators.cuh:79-83) with an exactly rounded reciprocal and square root.
__host__ __device__ __forceinline__ f3 normalized3(const f3& a) {
  const float rn = 1.0f / sqrtf(dot3(a, a));
  return mk3(a.x * rn, a.y * rn, a.z * rn);
}
__host__ __device__ __forceinline__ f3 mul(const M33& m, const f3& a) { return mk3(dot3(m.r0, a), dot3(m.r1, a), dot3(m.r2, a)); }

inline M33 to_m33(const dms_mat33* m) {
  M33 r;
  r.r0 = mk3(m->m[0], m->m[1], m->m[2]);
  r.r1 = mk3(m->m[3], m->m[4], m->m[5]);
  r.r2 = mk3(m->m[6], m->m[7], m->m[8]);
  return r;
}
inline f3 to_f3(const dms_float3* v) { return mk3(v->x, v->y, v->z); }

__host__ __device__ __forceinline__ float qnan() {
#if defined(__HIP_DEVICE_COMPILE__)
  return __int_as_float(0x7fffffff);
#else
  uint32_t u = 0x7fffffffu;
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// round-to-nearest-even float -> int; NaN -> 0, saturating (CUDA __float2int_rn semantics,
// which the reference relies on for NaN vertices).  v_cvt_i32_f32 has exactly these semantics (NaN -> 0,
// out-of-range values saturate, truncation): the instruction is named explicitly because the C conversion is undefined for
// those inputs and the guarded C form costs ~8 instructions, four times per pixel and iteration in the tracker.
__device__ __forceinline__ int f2i_rz(float v) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int f2i_rn(float v) { return f2i_rz(rintf(v)); }

// ---- wave64 reductions -----------------------------------------------------------------
// Butterfly inside each row of 16 lanes with DPP (no LDS traffic), then the two cross-row
// steps with row broadcasts.  After the call lane 63 holds the sum of all 64 lanes.
// The order is fixed, so results are run-to-run deterministic.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror     -> every lane holds its row's sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast15: rows 1,3 += lane 15 of rows 0,2
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast31: rows 2,3 += lane 31
  return v;
}

// The same for NV values at once, step-major: between two dependent DPP ops of one value sit the
// ops of the other NV-1 values, so the DPP read-after-write wait states (s_nop) disappear and the
// whole batch needs a single exec-masked store region afterwards.  Value by value (above) the
// compiler emits ~20 instructions per value; this form ~7.
template <int NV>
__device__ __forceinline__ void wave_sum_all_to_lane63(float (&v)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0xB1>(v[k]);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0x4E>(v[k]);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0x141>(v[k]);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0x140>(v[k]);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0x142, 0xa>(v[k]);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] += dpp_mov<0x143, 0xc>(v[k]);
}

// Sum of NV <= 32 per-lane values over the wave with the values spread over the lanes as the tree narrows: at every step
// the two partner lanes split their values — each keeps the sum of one half and hands the other half over — so a step
// costs one exchange per OUTPUT value (16 + 8 + 4 + 2 + 1 + 1 = 32 exchanges for 32 values) instead of one per value
// (6 x NV for the butterfly above: 261 instructions for the 29 sums of a Gauss-Newton pass, a third of the pass).
// Partners: lane ^ 1, ^ 2 (DPP quad permutes), ^ 4, ^ 8, ^ 16 (ds_swizzle, no memory), ^ 32 (one permute).  Returns, in
// lanes k and k + 32, the sum of value k over all 64 lanes (k < 32; zero for k >= NV).  Fixed order: deterministic.
template <int PATTERN>
__device__ __forceinline__ float swizzle_f(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN));
}
template <int NV>
__device__ __forceinline__ float wave_sum_transpose(const float (&v)[NV]) {
  static_assert(NV <= 32, "at most 32 values");
  const int lane = threadIdx.x & (kWave - 1);
  float a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) a[k] = k < NV ? v[k] : 0.f;
  float b[16], c[8], d[4], e[2];
  {
    const bool hi = (lane & 1) != 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = (hi ? a[2 * k + 1] : a[2 * k]) + dpp_mov<0xB1>(hi ? a[2 * k] : a[2 * k + 1]);
  }
  {
    const bool hi = (lane & 2) != 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = (hi ? b[2 * k + 1] : b[2 * k]) + dpp_mov<0x4E>(hi ? b[2 * k] : b[2 * k + 1]);
  }
  {
    const bool hi = (lane & 4) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = (hi ? c[2 * k + 1] : c[2 * k]) + swizzle_f<0x101F>(hi ? c[2 * k] : c[2 * k + 1]);
  }
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) e[k] = (hi ? d[2 * k + 1] : d[2 * k]) + swizzle_f<0x201F>(hi ? d[2 * k] : d[2 * k + 1]);
  }
  const bool hi = (lane & 16) != 0;
  float r = (hi ? e[1] : e[0]) + swizzle_f<0x401F>(hi ? e[0] : e[1]);
  r += __shfl_xor(r, 32, 64);
  return r;
}

// fp64 variant: the two halves of the double travel through the same DPP controls
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_mov_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_to_lane63_d(double v) {
  v += dpp_mov_d<0xB1>(v);
  v += dpp_mov_d<0x4E>(v);
  v += dpp_mov_d<0x141>(v);
  v += dpp_mov_d<0x140>(v);
  v += dpp_mov_d<0x142, 0xa>(v);
  v += dpp_mov_d<0x143, 0xc>(v);
  return v;
}

// sum over each row of 16 lanes, result in every lane of the row (fixed butterfly order)
__device__ __forceinline__ double row16_sum_d(double v) {
  v += dpp_mov_d<0xB1>(v);
  v += dpp_mov_d<0x4E>(v);
  v += dpp_mov_d<0x141>(v);
  v += dpp_mov_d<0x140>(v);
  return v;
}
// the same over each group of 8 lanes
__device__ __forceinline__ double row8_sum_d(double v) {
  v += dpp_mov_d<0xB1>(v);
  v += dpp_mov_d<0x4E>(v);
  v += dpp_mov_d<0x141>(v);
  return v;
}
__device__ __forceinline__ int row8_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  return v;
}
__device__ __forceinline__ int row16_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  return v;
}

__device__ __forceinline__ int wave_sum_to_lane63_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
  return v;
}

// Partial sums are stored one 128-byte record per block (kPartStride floats), so the fold reads
// them with 16-byte loads: 8 lanes cover one record, 32 record-groups are in flight per pass.
constexpr int kPartStride = 32;

// Block reduction of NV running sums held per thread.  256-thread blocks = 4 waves.
// fp32 DPP butterfly inside the wave, fp64 across the 4 waves, one fp32 record per block at
// out[out_col * kPartStride + k].  The fold of the records (fold_records256) is fp64: rounding
// noise of a tree sum is dominated by its top levels (the few, large partial sums), so keeping
// those in fp64 makes the total independent of launch shape to ~1e-8 relative and directly
// comparable with the oracle's fp64 accumulation, while the per-pixel / per-wave work stays fp32.
template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&v)[NV], float* out, int /*out_stride*/, int out_col) {
  __shared__ float lds[kBlock / kWave][NV];
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x >> 6;
  const float tot = wave_sum_transpose<NV>(v);
  if (lane < NV) lds[wid][lane] = tot;
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    const int nw = blockDim.x >> 6;
    double s = (double)lds[0][k];
    for (int w = 1; w < nw; ++w) s += (double)lds[w][k];
    out[(size_t)out_col * kPartStride + k] = (float)s;
  }
}

// Fold of the per-block records by one 256-thread block: thread (g = tid / 8, k4 = tid % 8) sums
// the float4 column k4 of records g, g + 32, ... .  Eight independent 16-byte loads per set are
// issued before the first add so their L2 latencies overlap (a load-add-load-add loop costs one
// round trip per record group: measured 4-6 us per fold); two record sets can be folded in one
// sweep.  Finally 32 lanes add the 32 group sums in a fixed order.  sums*[] is LDS or global.
template <bool TWO>
__device__ __forceinline__ void fold_records256_t(const float* partialsA, const float* partialsB, int nblocks, int nv, float* sumsA,
                                                  float* sumsB) {
  __shared__ double s_grp[TWO ? 2 : 1][32][32];
  constexpr int U = 8;
  const int k4 = threadIdx.x & 7, g = threadIdx.x >> 3;
  const float4* a4 = reinterpret_cast<const float4*>(partialsA);
  const float4* b4 = reinterpret_cast<const float4*>(partialsB);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  double aA[4] = {0., 0., 0., 0.}, aB[4] = {0., 0., 0., 0.};
  for (int b0 = g; b0 < nblocks; b0 += 32 * U) {
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + 32 * u;
      // clamped index + select AFTER the load: a conditional load makes the compiler serialise
      // the batch (one branch and one s_waitcnt per element)
      const int bc = b < nblocks ? b : 0;
      va[u] = a4[(size_t)bc * 8 + k4];
      if (TWO) vb[u] = b4[(size_t)bc * 8 + k4];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (b0 + 32 * u >= nblocks) {
        va[u] = z;
        if (TWO) vb[u] = z;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      aA[0] += (double)va[u].x; aA[1] += (double)va[u].y; aA[2] += (double)va[u].z; aA[3] += (double)va[u].w;
      if (TWO) { aB[0] += (double)vb[u].x; aB[1] += (double)vb[u].y; aB[2] += (double)vb[u].z; aB[3] += (double)vb[u].w; }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    s_grp[0][g][k4 * 4 + c] = aA[c];
    if (TWO) s_grp[TWO ? 1 : 0][g][k4 * 4 + c] = aB[c];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int set = threadIdx.x >> 5, k = threadIdx.x & 31;
    if (set == 0 || TWO) {
      double acc = 0.;
#pragma unroll
      for (int gg = 0; gg < 32; ++gg) acc += s_grp[TWO ? set : 0][gg][k];
      if (k < nv) (set == 0 ? sumsA : sumsB)[k] = (float)acc;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void fold_records256(const float* partials, int nblocks, int nv, float* sums) {
  fold_records256_t<false>(partials, partials, nblocks, nv, sums, sums);
}

// launch-shape helper: blocks of 256 threads, one pixel per thread up to the partial cap,
// grid-stride beyond.  ≥ 2 blocks per CU at 640×480 so every XCD's L2 sees its share.
inline int reduce_blocks_for(int n) {
  int b = (n + kBlock - 1) / kBlock;
  static const int cap = [] {  // tuning knob for A/B runs: DMS_PARTIAL_BLOCKS=64..1024
    const char* e = getenv("DMS_PARTIAL_BLOCKS");
    int v = e ? atoi(e) : kAutoPartialBlocks;
    return v < 64 ? 64 : (v > kMaxPartialBlocks ? kMaxPartialBlocks : v);
  }();
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return b;
}

// XCD-aware block order for per-pixel kernels: consecutive workgroups go round-robin to the 8 XCDs (each with its own L2), so
// with the identity order neighbouring tiles — which gather the same surfels and share window halos — sit in eight different
// L2s.  With `on`, the blocks of one XCD (b % 8) take a contiguous eighth of the tile sequence instead (DMS_XCD_REMAP=0: off).
// Used by the per-PIXEL passes (index resolve, splat resolve, associate: 2590 -> 2620 frames/s).  Measured and not used: the
// per-SURFEL passes (splat project 42 -> 49 us, clean flags 20 -> 24 us: contiguous ranges of the map have correlated cost —
// culled or not, in the window or not — so one XCD ends up with the expensive eighth) and the resident tracker levels
// (no change: their model-map gathers already hit in L2).
// (A grid that is not a multiple of 8 keeps its last nb % 8 blocks in place; the map is a bijection either way.)
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb, int on) {
  const unsigned nb8 = nb & ~7u;
  return (on && b < nb8) ? (b & 7u) * (nb8 >> 3) + (b >> 3) : b;
}
inline int xcd_remap_enabled() {
  static const int v = [] {
    const char* e = getenv("DMS_XCD_REMAP");
    return e ? (atoi(e) != 0 ? 1 : 0) : 1;
  }();
  return v;
}

inline dim3 grid2d(int cols, int rows, dim3 block) { return dim3((cols + block.x - 1) / block.x, (rows + block.y - 1) / block.y); }

// The thumbnail block a camera publishes / a key frame stores (include/dmslam_fusion.h, dms_thumb_block_bytes): n = tw * th pixels,
// [RGBA8 image, padded to a multiple of 16 bytes | RGBA32F vertex | RGBA32F normal] - the float4 sections stay 16-byte aligned for any
// n (1241 x 376: 155 x 47 thumbnails).  For n a multiple of 4 this is the packed layout of rounds 3 - 4.
__host__ __device__ inline size_t thumb_vertex_off(size_t n) { return (n * 4 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t thumb_normal_off(size_t n) { return thumb_vertex_off(n) + n * 16; }
__host__ __device__ inline size_t thumb_block_size(size_t n) { return thumb_vertex_off(n) + n * 32; }

}  // namespace dms
