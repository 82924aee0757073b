// Surfel map storage and the shared device helpers of the fusion kernels.
//
// HBM layout (MI355X-first; the reference keeps 60-byte interleaved records in a GL VBO,
// Shaders/Vertex.cpp:21-50): structure-of-arrays of float4 planes, so every pass streams only
// the attributes it needs with 16-byte coalesced accesses per lane:
//   pos[i] = (x, y, z, confidence)           col[i] = (colour, 0, initTime, stamp)
//   nrm[i] = (nx, ny, nz, radius)            times[s * capacity + i] = last-seen time of sensor s
// Two such sets ping-pong through the order-preserving compaction of `clean`.
#pragma once
#include "common.hpp"
#include "detmath.hpp"
#include "../../include/dmslam_fusion.h"

namespace dms {

struct SurfelPlanes {
  float4* pos;
  float4* col;
  float4* nrm;
  float* times;  // [DMS_MAX_SENSORS][cap]
};

constexpr unsigned kEmptyWinner = 0xFFFFFFFFu;
constexpr unsigned long long kZClear = (0xFFFFFFull << 32) | 0xFFFFFFFFull;  // depth 1.0 (24-bit), no surfel
constexpr int kScanChunk = 256;   // elements per scan block: one per thread, so no thread walks a serial chain of elements

// --- GL-rule helpers (DESIGN.md "Rasteriser rules") ---------------------------------------
// NEAREST texel of a normalised coordinate: floor(u * n) evaluated in fp32, CLAMP_TO_EDGE.
__host__ __device__ __forceinline__ int texel(float u, float n_f, int n) {
  int i = (int)floorf(u * n_f);
  if (i < 0) i = 0;
  if (i > n - 1) i = n - 1;
  return i;
}
// 24-bit fixed-point window depth of zw in [0,1]: round(zw * (2^24 - 1)), exact in fp64
__host__ __device__ __forceinline__ unsigned depth24(float zw) {
  if (!(zw >= 0.f)) return 0xFFFFFFFFu;  // clipped (also NaN)
  if (zw > 1.f) return 0xFFFFFFFFu;
  return (unsigned)(int)rint((double)zw * 16777215.0);  // (at most 2^24 - 1: the 32-bit conversion is the 64-bit one's value)
}

// mat4 * (p, 1): row-major, accumulated left to right, translation last
__host__ __device__ __forceinline__ f3 xform_point(const float* M, const f3& p) {
  return mk3(((M[0] * p.x + M[1] * p.y) + M[2] * p.z) + M[3], ((M[4] * p.x + M[5] * p.y) + M[6] * p.z) + M[7],
             ((M[8] * p.x + M[9] * p.y) + M[10] * p.z) + M[11]);
}
// mat3(M) * v
__host__ __device__ __forceinline__ f3 xform_dir(const float* M, const f3& v) {
  return mk3((M[0] * v.x + M[1] * v.y) + M[2] * v.z, (M[4] * v.x + M[5] * v.y) + M[6] * v.z,
             (M[8] * v.x + M[9] * v.y) + M[10] * v.z);
}
__host__ __device__ __forceinline__ float length3(const f3& a) { return sqrtf(dot3(a, a)); }

// surfels.glsl:19-34; cam_z = 1/fx, cam_w = 1/fy as the caller's `cam` uniform holds them
__host__ __device__ __forceinline__ float surfel_radius(float depth, float norm_z, float cam_z, float cam_w) {
  const float meanFocal = ((1.0f / fabsf(cam_z)) + (1.0f / fabsf(cam_w))) / 2.0f;
  const float sqrt2 = 1.41421356237f;
  const float radius = (depth / meanFocal) * sqrt2;
  float radius_n = radius;
  radius_n = radius_n / fabsf(norm_z);
  radius_n = fminf(2.0f * radius, radius_n);
  return radius_n;
}
// surfels.glsl:36-46
__host__ __device__ __forceinline__ float surfel_confidence(float x, float y, float cx, float cy, float weighting) {
  const float maxRadDist = 400.f;
  const float twoSigmaSquared = 0.72f;
  const float px = x - cx, py = y - cy;
  const float radialDist = sqrtf(px * px + py * py) / maxRadDist;
  return det_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}
// color.glsl:19-34
__host__ __device__ __forceinline__ float encode_color_bytes(unsigned r, unsigned g, unsigned b) {
  return (float)(int)((((r << 8) + g) << 8) + b);
}
__host__ __device__ __forceinline__ float encode_color(float cx, float cy, float cz) {
  int rgb = (int)roundf(cx * 255.0f);
  rgb = (rgb << 8) + (int)roundf(cy * 255.0f);
  rgb = (rgb << 8) + (int)roundf(cz * 255.0f);
  return (float)rgb;
}
__host__ __device__ __forceinline__ f3 decode_color(float c) {
  const int ci = (int)c;
  return mk3((float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}

// The 4×4-tap association / clean windows of data.vert:118-134 and copy_unstable.vert:84-122 walk
// a float loop `for (w = c - 2*step; w < c + 2*step; w += step)` with step = half a texel and
// fetch NEAREST, so consecutive taps repeat texels (at most 3 distinct per axis).  The loop is
// replayed here exactly (fp32 accumulation of `w`), but collapsed into distinct texels with
// multiplicities held in named slots — no dynamically indexed private arrays (those live in
// scratch memory), and each distinct texel is fetched once.
struct AxisTaps {
  int t0, t1, t2, t3;
  int m0, m1, m2, m3;
};
__device__ __forceinline__ AxisTaps axis_taps(float lo, float hi, float step, float nf, int n) {
  AxisTaps a = {-1, -1, -1, -1, 0, 0, 0, 0};
  int k = -1, last = -0x7fffffff;
  for (float w = lo; w < hi; w += step) {
    const int u = texel(w, nf, n);
    if (u != last) {
      ++k;
      last = u;
      if (k == 0) a.t0 = u;
      else if (k == 1) a.t1 = u;
      else if (k == 2) a.t2 = u;
      else a.t3 = u;
    }
    if (k == 0) a.m0++;
    else if (k == 1) a.m1++;
    else if (k == 2) a.m2++;
    else a.m3++;
  }
  return a;
}
template <class F>
__device__ __forceinline__ void for_taps(const AxisTaps& a, F f) {
  if (a.m0) f(a.t0, a.m0);
  if (a.m1) f(a.t1, a.m1);
  if (a.m2) f(a.t2, a.m2);
  if (a.m3) f(a.t3, a.m3);
}

// uv buffer entry of the reference (GlobalModel.cpp:100-108, FeedbackBuffer.cpp:38-46):
// ((float)i / (float)n) + 1.0 / (2 * (float)n) evaluated in double, stored as float
__host__ __device__ __forceinline__ float uv_coord(int i, int n) {
  return (float)((double)((float)i / (float)n) + 1.0 / (double)(2.0f * (float)n));
}

// G8: update (update.vert:42-104) of surfel `id` by its winning measurement `slot`, in place
__device__ __forceinline__ void fuse_update_apply(unsigned id, unsigned slot, const float4* __restrict__ slot_pos, const float4* __restrict__ slot_col,
                                                  const float4* __restrict__ slot_nrm, const SurfelPlanes& sp, size_t cap, int time, int timeIdx) {
  const float4 newPos = slot_pos[slot], newColor = slot_col[slot], newNorm = slot_nrm[slot];
  const float4 vPosition = sp.pos[id], vColor = sp.col[id], vNormRad = sp.nrm[id];
  const float c_k = vPosition.w;
  const float av = newPos.w;
  if (newNorm.w < (1.0f + 0.5f) * vNormRad.w) {
    const float wsum = c_k + av;
    sp.pos[id] = make_float4(((c_k * vPosition.x) + (av * newPos.x)) / wsum, ((c_k * vPosition.y) + (av * newPos.y)) / wsum,
                             ((c_k * vPosition.z) + (av * newPos.z)) / wsum, wsum);
    const f3 oldCol = decode_color(vColor.x), newCol = decode_color(newColor.x);
    const float ar = ((c_k * oldCol.x) + (av * newCol.x)) / wsum, ag = ((c_k * oldCol.y) + (av * newCol.y)) / wsum,
                ab = ((c_k * oldCol.z) + (av * newCol.z)) / wsum;
    sp.col[id] = make_float4(encode_color(ar, ag, ab), vColor.y, vColor.z, vColor.w);
    f3 n = mk3(((c_k * vNormRad.x) + (av * newNorm.x)) / wsum, ((c_k * vNormRad.y) + (av * newNorm.y)) / wsum,
               ((c_k * vNormRad.z) + (av * newNorm.z)) / wsum);
    const float r = ((c_k * vNormRad.w) + (av * newNorm.w)) / wsum;
    n = normalized3(n);
    sp.nrm[id] = make_float4(n.x, n.y, n.z, r);
  } else {
    sp.pos[id] = make_float4(vPosition.x, vPosition.y, vPosition.z, c_k + av);
  }
  sp.times[(size_t)timeIdx * cap + id] = (float)time;
}

}  // namespace dms

struct dms_model {
  size_t cap = 0;
  int width = 0, height = 0;
  int slots = 0, slot_h = 0;  // fuse candidate grid ((W+1)/2 × (H+1)/2), column-major slot = i*slot_h + j
  dms::SurfelPlanes buf[2];
  int cur = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  unsigned* d_count = nullptr;      // [0] model count.  Two cells exist; clean writes the new count into the other
  unsigned* d_count_alt = nullptr;  // one and the handles swap (no device copy, readers got the pointer at enqueue time)
  unsigned* h_count = nullptr;      // pinned mirror
  size_t count_upper = 0;           // host-side upper bound of the model count
  int count_hold = 0;               // frames for which the frame pipeline must not tighten count_upper from its (older) result blocks
  // fuse scratch (per candidate slot)
  float4 *slot_pos = nullptr, *slot_col = nullptr, *slot_nrm = nullptr;
  unsigned* slot_best = nullptr;
  unsigned char* slot_flag = nullptr;
  unsigned* winner = nullptr;       // [cap] column-major slot of the winning measurement per surfel
  // compaction scratch
  unsigned char* keep = nullptr;    // [cap + slots]
  unsigned* block_count = nullptr;  // [(cap + slots) / kScanChunk + 2]
  unsigned* block_offset = nullptr;
  unsigned* clean_first = nullptr;  // suffix-mode clean: index of the first block that is not left in place
  float* nodes = nullptr;           // deformation node table, 16 floats / node
  int max_nodes = 2048;
  // model_fuse(..., defer_update): the update pass (G8) has not run yet; the next index_map applies it surfel by surfel while
  // it projects (fusion_map.hip k_index_project<true>), with these arguments.  Only the frame step uses it: fuse is
  // always followed by the index map of the clean.
  bool pending_update = false;
  int pending_time = 0, pending_timeIdx = 0;
  unsigned long version = 0;        // bumped by every operation that changes the map (cached projections are tagged with it)
  int num_sensors = 3;              // per-surfel time slots that take part in the clean's health test (reference NUM_CAMERAS = 3)
  // Time planes that may hold anything but the "never seen by this sensor" marker -3: plane s carries real times only once a camera
  // with timeIdx == s has written into this map (or a map / records / an upload that had them was taken in).  Every buffer's planes
  // are filled with -3 at creation and every writer puts -3 into the planes of the other sensors, so the clean reads and moves only
  // the first live_planes planes of a surfel (1 for a single camera: 52 instead of 80 bytes per moved surfel).  Never decreases.
  int live_planes = 0;
  size_t clean_suffix_min = (size_t)1 << 20;  // map size from which the clean runs in suffix mode (DMS_CLEAN_SUFFIX_MIN at create)
  // After a map merge several cameras (frame-step contexts) fuse into this one map (dms_fusion_join_map): how many do, and which of
  // them enqueued the last frame - only that one's result block says anything about the current count.
  int sharers = 1;
  const void* last_writer = nullptr;
};
