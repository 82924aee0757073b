// Fusion of a frame into the surfel map (G7 data.*, G8 update.vert) and map maintenance
// (G9 copy_unstable.*).
//
// MI355X design.  The reference runs data.vert over all W·H pixels, scatters the measurements
// into three 5700² RGBA32F textures addressed by surfel id, then streams the *whole* map through
// update.vert (120 B/surfel) to pick them up.  Here:
//   k_fuse_associate  one thread per candidate pixel (¼ of the image: the parity gate of
//                     data.vert:112): measurement + 4×4-tap association; the measurement is
//                     parked in its column-major slot and competes for its surfel with
//                     atomicMin(winner[id], slot) — "first in draw order wins" of the
//                     depth-tested scatter;
//   k_fuse_update     one thread per candidate slot: the winning measurement is averaged into
//                     its surfel in place (update.vert:59-96).  O(pixels), not O(map).
//   k_fuse_emit       stable compaction of the slots into the new-unstable list (data.geom
//                     emits both merged and new measurements, in draw order).
// clean = flags → block scan → scatter over [map surfels ‖ new-unstable list], preserving
// order (transform-feedback semantics), with the window tests of copy_unstable.vert.
#include "scan.hpp"
#include "smallmath.hpp"
#include <utility>

#include "surfel.hpp"

namespace dms {

// ---------------------------------------------------------------------------------------
// G7: association (data.vert:76-192)
// ---------------------------------------------------------------------------------------
struct FuseArgs {
  const dms_pose_block* pose;
  const uchar4* rgba;
  const float* dr;   // DEPTH_METRIC
  const float* drf;  // DEPTH_METRIC_FILTERED
  const unsigned* index;
  const float4* vertConf;
  const float4* normRad;
  int cols, rows, slot_h;
  float cx, cy, icx, icy;  // cam = (cx, cy, 1/fx, 1/fy) with the reciprocals taken in double (GlobalModel.cpp:546-550)
  float maxDepth, timef, weighting;
  const float* weighting_dev;
  int time, timeIdx;
  int transposed;  // index-map images stored column-major (fusion_map.hip ProjArgs::transposed)
  int xcd;         // XCD-aware block order (common.hpp xcd_block)
};

__device__ __forceinline__ f3 dv_vertex(const float* depth, int cols, int sx, int sy, float x, float y, float cx, float cy, float icx,
                                        float icy) {  // geometry.glsl:21-25
  const float z = depth[(size_t)sy * cols + sx];
  return mk3(((x - cx) * z) * icx, ((y - cy) * z) * icy, z);
}

__device__ __forceinline__ f3 dv_normal(const float* depth, int cols, int rows, const f3& vPosition, float tx, float ty, float x, float y,
                                        float cx, float cy, float icx, float icy) {  // geometry.glsl:27-39
  const float colsf = (float)cols, rowsf = (float)rows;
  const int sx = texel(tx, colsf, cols), sy = texel(ty, rowsf, rows);
  const int sxf = texel(tx + (1.0f / colsf), colsf, cols), sxb = texel(tx - (1.0f / colsf), colsf, cols);
  const int syf = texel(ty + (1.0f / rowsf), rowsf, rows), syb = texel(ty - (1.0f / rowsf), rowsf, rows);
  const f3 xf = dv_vertex(depth, cols, sxf, sy, x + 1.f, y, cx, cy, icx, icy);
  const f3 xb = dv_vertex(depth, cols, sxb, sy, x - 1.f, y, cx, cy, icx, icy);
  const f3 yf = dv_vertex(depth, cols, sx, syf, x, y + 1.f, cx, cy, icx, icy);
  const f3 yb = dv_vertex(depth, cols, sx, syb, x, y - 1.f, cx, cy, icx, icy);
  const f3 del_x = mk3(((xb.x + vPosition.x) / 2.f) - ((xf.x + vPosition.x) / 2.f), ((xb.y + vPosition.y) / 2.f) - ((xf.y + vPosition.y) / 2.f),
                       ((xb.z + vPosition.z) / 2.f) - ((xf.z + vPosition.z) / 2.f));
  const f3 del_y = mk3(((yb.x + vPosition.x) / 2.f) - ((yf.x + vPosition.x) / 2.f), ((yb.y + vPosition.y) / 2.f) - ((yf.y + vPosition.y) / 2.f),
                       ((yb.z + vPosition.z) / 2.f) - ((yf.z + vPosition.z) / 2.f));
  return normalized3(cross3(del_x, del_y));
}

__device__ __forceinline__ float angle_between(const f3& a, const f3& b) {  // data.vert:66-69
  return det_acosf(dot3(a, b) / (length3(a) * length3(b)));
}

__global__ __launch_bounds__(256) void k_fuse_associate(FuseArgs a, float4* __restrict__ slot_pos, float4* __restrict__ slot_col,
                                                        float4* __restrict__ slot_nrm, unsigned* __restrict__ slot_best,
                                                        unsigned char* __restrict__ slot_flag, unsigned* __restrict__ winner, int slot_w) {
  // candidate (i, j) -> pixel (2i + p, 2j + p), p = time % 2; slot = i * slot_h + j (column-major)
  // Four lanes per candidate: lane `sub` of a quad evaluates the (up to four) taps of x slot `sub` of the association
  // window, the quad then agrees on the winner by shuffles and its lane 0 writes the slot.  (One lane per candidate —
  // 76 800 threads, 48 dependent-latency gathers each — left the kernel latency-bound at about one wave per SIMD: 15.8 us,
  // of which the evaluation itself was 2.)  One 2 x 8 tile of candidates per wave, quads running down the column first:
  // the column-major index maps and the row-major live images are both read in runs of 8 neighbouring candidates.
  const int tiles_j = (a.slot_h + 7) >> 3;
  const int t = xcd_block(blockIdx.x, gridDim.x, a.xcd) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int ti = t / tiles_j, tj = t - ti * tiles_j;
  const int lane = threadIdx.x & 63, quad = lane >> 2, sub = lane & 3;
  const int i = ti * 2 + (quad >> 3);
  const int j = tj * 8 + (quad & 7);
  if (i >= slot_w || j >= a.slot_h) return;
  const int slot = i * a.slot_h + j;
  const int par = ((a.time % 2) + 2) % 2;
  const int px = 2 * i + par, py = 2 * j + par;
  unsigned char flag = 0;
  if (px < a.cols && py < a.rows) {
    const float colsf = (float)a.cols, rowsf = (float)a.rows;
    const float tx = uv_coord(px, a.cols), ty = uv_coord(py, a.rows);
    const float x = tx * colsf, y = ty * rowsf;
    const int sx = texel(tx, colsf, a.cols), sy = texel(ty, rowsf, a.rows);
    const f3 vPosLocal = dv_vertex(a.dr, a.cols, sx, sy, x, y, a.cx, a.cy, a.icx, a.icy);
    // data.vert:112-114 gate (int(x)%2 == int(time)%2 etc. holds by construction of (px, py))
    bool ok = ((int)x % 2 == (int)a.timef % 2) && ((int)y % 2 == (int)a.timef % 2);
    if (ok) {  // checkNeighbours on the raw metric depth (data.vert:45-64)
      const int sxb = texel(tx - (1.0f / colsf), colsf, a.cols), sxf = texel(tx + (1.0f / colsf), colsf, a.cols);
      const int syb = texel(ty - (1.0f / rowsf), rowsf, a.rows), syf = texel(ty + (1.0f / rowsf), rowsf, a.rows);
      ok = !(a.dr[(size_t)sy * a.cols + sxb] == 0.f) && !(a.dr[(size_t)syb * a.cols + sx] == 0.f) &&
           !(a.dr[(size_t)sy * a.cols + sxf] == 0.f) && !(a.dr[(size_t)syf * a.cols + sx] == 0.f);
    }
    ok = ok && vPosLocal.z > 0.f && vPosLocal.z <= a.maxDepth;
    // a negative device-side weight marks a frame whose tracker result is invalid (grid-barrier timeout,
    // frame_after_track_body): nothing is measured, so nothing merges and nothing is appended
    const float weighting = a.weighting_dev ? *a.weighting_dev : a.weighting;
    ok = ok && !(weighting < 0.f);
    if (ok) {
      const float* P = a.pose->pose;
      const f3 vPos = xform_point(P, vPosLocal);
      const f3 vPos_f = dv_vertex(a.drf, a.cols, sx, sy, x, y, a.cx, a.cy, a.icx, a.icy);
      const uchar4 c = a.rgba[(size_t)sy * a.cols + sx];
      const f3 vNormLocal = dv_normal(a.drf, a.cols, a.rows, vPos_f, tx, ty, x, y, a.cx, a.cy, a.icx, a.icy);
      const f3 nG = xform_dir(P, vNormLocal);
      const float rad = surfel_radius(vPos_f.z, vNormLocal.z, a.icx, a.icy);
      const float conf = surfel_confidence(x, y, a.cx, a.cy, weighting);

      // 4×4-tap association window in the index map (data.vert:116-160)
      int counter = 0;
      unsigned best = 0u;
      const float scale = 1.0f;  // IndexMap::FACTOR
      const float indexXStep = (1.0f / (colsf * scale)) * 0.5f;
      const float indexYStep = (1.0f / (rowsf * scale)) * 0.5f;
      float bestDist = 1000.f;
      const float windowMultiplier = 2.f;
      const float xl = (x - a.cx) * a.icx;
      const float yl = (y - a.cy) * a.icy;
      const float lambda = sqrtf((xl * xl + yl * yl) + 1.f);
      const f3 ray = mk3(xl, yl, 1.f);
      const float ray_len = length3(ray);
      const float x_lo = tx - ((scale * indexXStep) * windowMultiplier), x_hi = tx + ((scale * indexXStep) * windowMultiplier);
      const float y_lo = ty - ((scale * indexYStep) * windowMultiplier), y_hi = ty + ((scale * indexYStep) * windowMultiplier);
      // repeated taps of one texel cannot change `best` (the distance test is strict) and
      // `counter` only matters as > 0, so each distinct texel is visited once, in tap order
      const AxisTaps tx_ = axis_taps(x_lo, x_hi, indexXStep, colsf, a.cols);
      const AxisTaps ty_ = axis_taps(y_lo, y_hi, indexYStep, rowsf, a.rows);
      // Sequentially (data.vert:118-160) a tap is accepted when it passes the depth and normal tests and lies strictly closer
      // to the ray than every tap accepted before it: the winner is the closest tap of those that pass the two tests, the
      // earliest in tap order (x outer, y inner) among equals — which can be evaluated per tap and reduced.
      const int txs_s = sub == 0 ? tx_.t0 : sub == 1 ? tx_.t1 : sub == 2 ? tx_.t2 : tx_.t3;
      const int mxs_s = sub == 0 ? tx_.m0 : sub == 1 ? tx_.m1 : sub == 2 ? tx_.m2 : tx_.m3;
      const int tys[4] = {ty_.t0, ty_.t1, ty_.t2, ty_.t3}, mys[4] = {ty_.m0, ty_.m1, ty_.m2, ty_.m3};
      size_t q[4];
      bool used[4];
      unsigned cur[4];
      float4 vcs[4], nrs[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const bool u = mxs_s != 0 && mys[jj] != 0;
        const int ux = u ? txs_s : 0, uy = u ? tys[jj] : 0;
        used[jj] = u;
        q[jj] = a.transposed ? (size_t)ux * a.rows + uy : (size_t)uy * a.cols + ux;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        cur[jj] = a.index[q[jj]];
        vcs[jj] = a.vertConf[q[jj]];
        nrs[jj] = a.normRad[q[jj]];
      }
      int order = 64;  // tap order of this lane's winner (sub * 4 + jj); 64 = none
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const unsigned current = cur[jj];
        if (used[jj] && current > 0u) {
          const float4 vc = vcs[jj];
          if (fabsf((vc.z * lambda) - (vPosLocal.z * lambda)) < 0.05f) {
            const float dist = length3(cross3(ray, mk3(vc.x, vc.y, vc.z))) / ray_len;
            const float4 nr = nrs[jj];
            if (dist < bestDist && (fabsf(nr.z) < 0.75f || fabsf(angle_between(mk3(nr.x, nr.y, nr.z), vNormLocal)) < 0.5f)) {
              counter++;
              bestDist = dist;
              best = current;
              order = sub * 4 + jj;
            }
          }
        }
      }
      // the quad's winner: smaller distance, then earlier tap (a lane without a winner holds distance 1000, order 64)
#pragma unroll
      for (int m = 1; m < 4; m <<= 1) {
        const float od = __shfl_xor(bestDist, m, 64);
        const int oo = __shfl_xor(order, m, 64);
        const unsigned ob = __shfl_xor(best, m, 64);
        counter += __shfl_xor(counter, m, 64);
        if (od < bestDist || (od == bestDist && oo < order)) {
          bestDist = od;
          order = oo;
          best = ob;
        }
      }
      flag = counter > 0 ? 1 : 2;
      if (sub == 0) {
        slot_pos[slot] = make_float4(vPos.x, vPos.y, vPos.z, conf);
        slot_col[slot] = make_float4(encode_color_bytes(c.x, c.y, c.z), 0.f, a.timef, flag == 1 ? -1.f : -2.f);
        slot_nrm[slot] = make_float4(nG.x, nG.y, nG.z, rad);
        slot_best[slot] = best;
        if (flag == 1) atomicMin(winner + best, (unsigned)slot);
      }
    }
  }
  if (sub == 0) slot_flag[slot] = flag;
}

// ---------------------------------------------------------------------------------------
// G8: update (update.vert:42-104), in place, by the winning measurement of each surfel
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fuse_update(int nslots, const float4* __restrict__ slot_pos, const float4* __restrict__ slot_col,
                                                     const float4* __restrict__ slot_nrm, const unsigned* __restrict__ slot_best,
                                                     const unsigned char* __restrict__ slot_flag, unsigned* __restrict__ winner,
                                                     SurfelPlanes sp, size_t cap, int time, int timeIdx) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots || slot_flag[slot] != 1) return;
  const unsigned id = slot_best[slot];
  if (winner[id] != (unsigned)slot) return;
  winner[id] = kEmptyWinner;  // re-arm for the next frame
  fuse_update_apply(id, (unsigned)slot, slot_pos, slot_col, slot_nrm, sp, cap, time, timeIdx);
}

// ---------------------------------------------------------------------------------------
// G9: clean (copy_unstable.vert:53-352) over elements [0, M) = map, [M, M + slots) = slots
// ---------------------------------------------------------------------------------------
struct CleanArgs {
  const dms_pose_block* pose;
  const unsigned* index;
  const float4* vertConf;
  const float4* colorTime;
  const float* depth_synth;
  int cols, rows;
  float cx, cy, fx, fy;
  float confThreshold, maxDepth;
  int time, timeIdx, timeDelta, isFern;
  int nodes;
  const float* node_table;
  int nslots;
  int transposed;
  int suffix;  // large maps: surfels in the leading run of untouched blocks stay where they are (see model_clean)
  int num_sensors;  // vTimes.length() of copy_unstable.vert: the reference's NUM_CAMERAS = 3 (size.glsl:2), <= DMS_MAX_SENSORS
  int planes;       // time planes that can hold anything but -3 (surfel.hpp, live_planes): the others are neither read nor moved
};

// plain aggregates (HIP's float4 is a class with a union inside: as a member of the element it keeps
// the whole 96-byte element in scratch memory instead of registers)
struct F4 {
  float x, y, z, w;
};
__device__ __forceinline__ F4 ld4(const float4* p) {
  const float4 t = *p;
  F4 r;
  r.x = t.x;
  r.y = t.y;
  r.z = t.z;
  r.w = t.w;
  return r;
}
__device__ __forceinline__ float4 to4(const F4& v) { return make_float4(v.x, v.y, v.z, v.w); }
struct CleanElem {
  F4 pos, col, nrm;
  float times[DMS_MAX_SENSORS];
  float vt;  // this sensor's time (times[timeIdx]), kept separately: selecting it out of the array by a runtime
             // index makes the compiler index the array dynamically, which moves the whole element to scratch memory
};

// load element e (map surfel or parked measurement); false when the slot holds nothing
__device__ __forceinline__ bool clean_load(unsigned e, unsigned M, const SurfelPlanes& sp, size_t cap, const float4* slot_pos,
                                           const float4* slot_col, const float4* slot_nrm, const unsigned char* slot_flag, int nslots,
                                           int timeIdx, int planes, CleanElem& o) {
  if (e < M) {
    o.pos = ld4(sp.pos + e);
    o.col = ld4(sp.col + e);
    o.nrm = ld4(sp.nrm + e);
#pragma unroll
    for (int s = 0; s < DMS_MAX_SENSORS; ++s) o.times[s] = s < planes ? sp.times[(size_t)s * cap + e] : -3.f;  // (what the plane holds)
    o.vt = sp.times[(size_t)timeIdx * cap + e];
    return true;
  }
  const unsigned slot = e - M;
  if ((int)slot >= nslots || slot_flag[slot] == 0) return false;
  o.pos = ld4(slot_pos + slot);
  o.col = ld4(slot_col + slot);
  o.nrm = ld4(slot_nrm + slot);
  // data.geom:50-54: -3 for every sensor but this one, which carries the -1 / -2 marker
#pragma unroll
  for (int s = 0; s < DMS_MAX_SENSORS; ++s) o.times[s] = (s == timeIdx) ? o.col.w : -3.f;
  o.vt = o.col.w;
  return true;
}

__device__ __forceinline__ int clean_test(const CleanArgs& a, const CleanElem& v) {
  int test = 1;
  const float* Tinv = a.pose->t_inv;
  const f3 localPos = xform_point(Tinv, mk3(v.pos.x, v.pos.y, v.pos.z));
  const float x = ((a.fx * localPos.x) / localPos.z) + a.cx;
  const float y = ((a.fy * localPos.y) / localPos.z) + a.cy;
  const f3 localNorm = normalized3(xform_dir(Tinv, mk3(v.nrm.x, v.nrm.y, v.nrm.z)));
  const float colsf = (float)a.cols, rowsf = (float)a.rows;
  const float scale = 1.0f;
  const float indexXStep = (1.0f / (colsf * scale)) * 0.5f;
  const float indexYStep = (1.0f / (rowsf * scale)) * 0.5f;
  const float windowMultiplier = 2.f;
  int count = 0, zCount = 0;
  const float vt = v.vt;
  if ((float)a.time - vt < (float)a.timeDelta && localPos.z > 0.f && x > 0.f && y > 0.f && x < colsf && y < rowsf) {
    const float xc = x / colsf, yc = y / rowsf;
    const float x_lo = xc - ((scale * indexXStep) * windowMultiplier), x_hi = xc + ((scale * indexXStep) * windowMultiplier);
    const float y_lo = yc - ((scale * indexYStep) * windowMultiplier), y_hi = yc + ((scale * indexYStep) * windowMultiplier);
    const AxisTaps tx_ = axis_taps(x_lo, x_hi, indexXStep, colsf, a.cols);
    const AxisTaps ty_ = axis_taps(y_lo, y_hi, indexYStep, rowsf, a.rows);
    // The (at most 4 x 4) distinct texels are visited one x slot at a time: the 4 vertex / confidence loads of a slot's
    // four texels are in flight together, a slot no lane of the wave uses is skipped, and the register footprint stays
    // small enough that the 80-byte element is not spilled.  Unused y slots carry multiplicity 0 and read texel 0.
    // Both counts need a stable texel that lies behind the surfel (vc.w > confThreshold && vc.z > localPos.z): only for
    // those is the texel's id and colour / time pair fetched — the window loads are what this kernel spends its time
    // on (27 of 35 us), and about half of the taps end at the first test.
    const int txs[4] = {tx_.t0, tx_.t1, tx_.t2, tx_.t3}, mxs[4] = {tx_.m0, tx_.m1, tx_.m2, tx_.m3};
    const int tys[4] = {ty_.t0, ty_.t1, ty_.t2, ty_.t3}, mys[4] = {ty_.m0, ty_.m1, ty_.m2, ty_.m3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (mxs[i] != 0) {  // (divergent lanes wait here; a slot unused by the whole wave costs nothing)
        float4 vcs[4];
        unsigned qs[4];
        int mult[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mult[j] = mxs[i] * mys[j];
          const int uy = mult[j] ? tys[j] : 0;
          qs[j] = a.transposed ? (unsigned)txs[i] * (unsigned)a.rows + (unsigned)uy : (unsigned)uy * (unsigned)a.cols + (unsigned)txs[i];
          vcs[j] = a.vertConf[qs[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mult[j];
          const float4 vc = vcs[j];
          if (m != 0 && vc.w > a.confThreshold && vc.z > localPos.z) {
            const unsigned cur = a.index[qs[j]];
            const float2 ct = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(&a.colorTime[qs[j]]) + 2);  // .z .w
            if (cur > 0u) {
              const float ctz = ct.x, ctw = ct.y;
              const float dx = vc.x - localPos.x, dy = vc.y - localPos.y;
              if (ctz < v.col.z && vc.z - localPos.z < 0.01f && sqrtf(dx * dx + dy * dy) < v.nrm.w * 1.4f) count += m;  // every repeated tap of this texel counts
              if (ctw == (float)a.time && vc.z - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f) zCount += m;
            }
          }
        }
      }
    }
  }
  if (count > 8 || zCount > 4) test = 0;
  // new unstable point: times become `time` before the health test (copy_unstable.vert:124-129)
  const float vt2 = (vt == -2.f) ? (float)a.time : vt;
  // "unhealthy for every sensor" (copy_unstable.vert:137-150): the loop runs over vTimes.length() = NUM_CAMERAS
  // slots (3 in the reference, size.glsl:2).  The map stores DMS_MAX_SENSORS slots; only the first
  // a.num_sensors take part — an unused slot holds -3 and would otherwise count as healthy during the
  // first 17 ticks, keeping surfels the reference removes.
  int unHealthy = 0;
#pragma unroll
  for (int s = 0; s < DMS_MAX_SENSORS; ++s) {
    const float ts = (s == a.timeIdx) ? vt2 : v.times[s];
    if (s < a.num_sensors && (ts == -1.f || (((float)a.time - ts) > 20.f && v.pos.w < a.confThreshold))) unHealthy++;
  }
  if (unHealthy == a.num_sensors) test = 0;
  if (vt2 > 0.f && (float)a.time - vt2 > (float)a.timeDelta) test = 1;
  return test;
}

__global__ __launch_bounds__(256) void k_clean_flags(CleanArgs a, SurfelPlanes sp, size_t cap, const unsigned* __restrict__ d_count,
                                                     const float4* slot_pos, const float4* slot_col, const float4* slot_nrm,
                                                     const unsigned char* slot_flag, unsigned char* __restrict__ keep,
                                                     unsigned* __restrict__ block_count, unsigned* __restrict__ first_touched) {
  const unsigned M = d_count[0];
  const unsigned total = M + (unsigned)a.nslots;
  const unsigned base = blockIdx.x * kScanChunk;
  if (first_touched && blockIdx.x == 0 && threadIdx.x == 0) *first_touched = 0xFFFFFFFFu;  // the scan takes the minimum
  unsigned cnt = 0;
  if (base < total) {
    for (int k = 0; k < kScanChunk / 256; ++k) {
      const unsigned e = base + k * 256 + threadIdx.x;
      unsigned char f = 0;
      if (e < total) {
        CleanElem v;
        if (a.suffix && e < M) {
          // large maps are bandwidth-bound here: colour and normal (32 of the 80 bytes) are only used by the
          // window test of surfels that are in view and inside the time window — fetch them for those only
          v.pos = ld4(sp.pos + e);
#pragma unroll
          for (int s = 0; s < DMS_MAX_SENSORS; ++s) v.times[s] = s < a.planes ? sp.times[(size_t)s * cap + e] : -3.f;
          v.vt = sp.times[(size_t)a.timeIdx * cap + e];
          const f3 lp = xform_point(a.pose->t_inv, mk3(v.pos.x, v.pos.y, v.pos.z));
          const float x = ((a.fx * lp.x) / lp.z) + a.cx, y = ((a.fy * lp.y) / lp.z) + a.cy;
          const bool windowed = (float)a.time - v.vt < (float)a.timeDelta && lp.z > 0.f && x > 0.f && y > 0.f && x < (float)a.cols &&
                                y < (float)a.rows;  // the condition of clean_test, evaluated the same way
          const F4 zero = {0.f, 0.f, 0.f, 0.f};
          v.col = windowed ? ld4(sp.col + e) : zero;
          v.nrm = windowed ? ld4(sp.nrm + e) : zero;
          f = (unsigned char)clean_test(a, v);
        } else if (clean_load(e, M, sp, cap, slot_pos, slot_col, slot_nrm, slot_flag, a.nslots, a.timeIdx, a.planes, v)) {
          f = (unsigned char)clean_test(a, v);
        }
        keep[e] = f;
      }
      cnt += f;
    }
  }
  const unsigned tot = block_sum_u32(cnt);
  if (threadIdx.x == 0) block_count[blockIdx.x] = tot;
}

// deformation-graph application (copy_unstable.vert:161-351)
__device__ __forceinline__ f3 node_pos(const float* nt, int j) { return mk3(nt[j * 16 + 0], nt[j * 16 + 1], nt[j * 16 + 2]); }

__device__ void clean_deform(const CleanArgs& a, CleanElem& v) {
  const float* nt = a.node_table;
  const int nodes = a.nodes;
  const int k = 4;
  const int lookBack = 20;
  int nearNodes[20];
  float nearDists[20];
  for (int i = 0; i < lookBack; i++) {
    nearNodes[i] = -1;
    nearDists[i] = 16777216.0f;
  }
  const int poseTime = (int)v.col.z;
  int foundIndex = 0;
  int imin = 0, imax = nodes - 1, imid = (imin + imax) / 2;
  while (imax >= imin) {
    imid = (imin + imax) / 2;
    const int nodeTime = (int)nt[imid * 16 + 15];
    if (nodeTime < poseTime)
      imin = imid + 1;
    else if (nodeTime > poseTime)
      imax = imid - 1;
    else
      break;
  }
  imin = min(imin, nodes - 1);
  // the node table is a 1-D NEAREST / CLAMP_TO_EDGE texture: texel (k*16+15) for k >= 0; imax = -1
  // (pose older than every node) samples a negative coordinate, which clamps to texel 0
  const int nodeMin = (int)nt[imin * 16 + 15];
  const int nodeMid = (int)nt[imid * 16 + 15];
  const int nodeMax = imax < 0 ? (int)nt[0] : (int)nt[imax * 16 + 15];
  if (abs(nodeMin - poseTime) <= abs(nodeMid - poseTime) && abs(nodeMin - poseTime) <= abs(nodeMax - poseTime))
    foundIndex = imin;
  else if (abs(nodeMid - poseTime) <= abs(nodeMin - poseTime) && abs(nodeMid - poseTime) <= abs(nodeMax - poseTime))
    foundIndex = imid;
  else
    foundIndex = imax;
  if (foundIndex == nodes) foundIndex = nodes - 1;
  if (foundIndex < 0) foundIndex = 0;
  const f3 P = mk3(v.pos.x, v.pos.y, v.pos.z);
  int nearNodeIndex = 0, distanceBack = 0;
  for (int j = foundIndex; j >= 0; j--) {
    const f3 d = P - node_pos(nt, j);
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = sqrtf(dot3(d, d));
    nearNodeIndex++;
    if (++distanceBack == lookBack / 2) break;
  }
  for (int j = foundIndex + 1; j < nodes; j++) {
    const f3 d = P - node_pos(nt, j);
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = sqrtf(dot3(d, d));
    nearNodeIndex++;
    if (++distanceBack == lookBack) break;
  }
  for (int i = 0; i < lookBack - 1; ++i)
    for (int j = i + 1; j < lookBack; ++j)
      if (nearDists[j] < nearDists[i]) {
        const float t = nearDists[i];
        nearDists[i] = nearDists[j];
        nearDists[j] = t;
        const int t2 = nearNodes[i];
        nearNodes[i] = nearNodes[j];
        nearNodes[j] = t2;
      }
  const float dMax = nearDists[k];
  float nodeWeights[4];
  float weightSum = 0.f;
  for (int j = 0; j < k; j++) {
    const int nj = max(nearNodes[j], 0);
    const f3 d = P - node_pos(nt, nj);
    const float w1 = 1.0f - (sqrtf(dot3(d, d)) / dMax);
    nodeWeights[j] = w1 * w1;  // pow(., 2)
    weightSum += nodeWeights[j];
  }
  for (int j = 0; j < k; j++) nodeWeights[j] /= weightSum;
  f3 newPos = mk3(0.f, 0.f, 0.f), newNorm = mk3(0.f, 0.f, 0.f);
  const f3 N = mk3(v.nrm.x, v.nrm.y, v.nrm.z);
  for (int i = 0; i < k; i++) {
    const int ni = max(nearNodes[i], 0);
    const float* q = nt + ni * 16;
    const f3 g = mk3(q[0], q[1], q[2]);
    // rotation stored column-major (Eigen, Deformation.cpp:192-201): columns (q3..5), (q6..8), (q9..11)
    const float R[9] = {q[3], q[6], q[9], q[4], q[7], q[10], q[5], q[8], q[11]};  // row-major
    const f3 tr = mk3(q[12], q[13], q[14]);
    const f3 d = P - g;
    const f3 Rd = mk3((R[0] * d.x + R[1] * d.y) + R[2] * d.z, (R[3] * d.x + R[4] * d.y) + R[5] * d.z, (R[6] * d.x + R[7] * d.y) + R[8] * d.z);
    const f3 cand = (Rd + g) + tr;
    newPos = newPos + mk3(nodeWeights[i] * cand.x, nodeWeights[i] * cand.y, nodeWeights[i] * cand.z);
    float Ri[9], RiT[9];
    sm::inv3<float>(R, Ri);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) RiT[r * 3 + c] = Ri[c * 3 + r];
    const f3 nn = mk3((RiT[0] * N.x + RiT[1] * N.y) + RiT[2] * N.z, (RiT[3] * N.x + RiT[4] * N.y) + RiT[5] * N.z,
                      (RiT[6] * N.x + RiT[7] * N.y) + RiT[8] * N.z);
    newNorm = newNorm + mk3(nodeWeights[i] * nn.x, nodeWeights[i] * nn.y, nodeWeights[i] * nn.z);
  }
  v.pos.x = newPos.x;
  v.pos.y = newPos.y;
  v.pos.z = newPos.z;
  const f3 nn = normalized3(newNorm);
  v.nrm.x = nn.x;
  v.nrm.y = nn.y;
  v.nrm.z = nn.z;
  if (v.pos.w > a.confThreshold && a.isFern == 0) {
    const float* Tinv = a.pose->t_inv;
    const f3 lp = xform_point(Tinv, mk3(v.pos.x, v.pos.y, v.pos.z));
    const float x = ((a.fx * lp.x) / lp.z) + a.cx;
    const float y = ((a.fy * lp.y) / lp.z) + a.cy;
    const float colsf = (float)a.cols, rowsf = (float)a.rows;
    if (lp.z > 0.f && lp.z < a.maxDepth && x > 0.f && y > 0.f && x < colsf && y < rowsf) {
      const int ux = texel(x / colsf, colsf, a.cols), uy = texel(y / rowsf, rowsf, a.rows);
      const float currentDepth = a.depth_synth ? a.depth_synth[(size_t)uy * a.cols + ux] : 0.f;
      if (currentDepth > 0.0f && lp.z < currentDepth + 0.1f) {
        v.col.w = (float)a.time;
        v.vt = (float)a.time;
      }
    }
  }
}

// Suffix mode (maps of a million surfels and more): in steady state removals and appends happen among
// the most recent surfels, i.e. at the end of the buffer; a block of kScanChunk map surfels that is
// complete and preceded only by complete blocks keeps its place and its contents (the graph-free clean
// changes nothing in a surviving map surfel), so it is neither read nor written again.  Such blocks
// form a prefix; the scatter stages everything behind it in the other buffer and k_clean_copy_back
// returns it, so the map stays in one buffer and the traffic is 80 B per surfel for the flags plus
// 320 B per surfel of the suffix instead of 240 B per surfel of the whole map.
__global__ __launch_bounds__(256) void k_clean_copy_back(SurfelPlanes staged, SurfelPlanes map, size_t cap, const unsigned* __restrict__ count_new,
                                                         const unsigned* __restrict__ first_touched, int planes) {
  // a fixed grid walks the blocks behind the prefix that stayed in place: the work follows the suffix, not the map
  const unsigned n = count_new[0];
  for (unsigned b = first_touched[0] + blockIdx.x; (size_t)b * kScanChunk < n; b += gridDim.x) {
    const unsigned i = b * kScanChunk + threadIdx.x;
    if (i >= n) continue;
    map.pos[i] = staged.pos[i];
    map.col[i] = staged.col[i];
    map.nrm[i] = staged.nrm[i];
#pragma unroll
    for (int s = 0; s < DMS_MAX_SENSORS; ++s)
      if (s < planes) map.times[(size_t)s * cap + i] = staged.times[(size_t)s * cap + i];
  }
}

__global__ __launch_bounds__(256) void k_clean_scatter(CleanArgs a, SurfelPlanes sp, size_t cap, const unsigned* __restrict__ d_count,
                                                       const float4* slot_pos, const float4* slot_col, const float4* slot_nrm,
                                                       unsigned char* slot_flag, const unsigned char* __restrict__ keep,
                                                       const unsigned* __restrict__ block_offset, SurfelPlanes out,
                                                       const unsigned* __restrict__ block_count, unsigned* __restrict__ count_new,
                                                       unsigned* __restrict__ count_new2) {
  const unsigned M = d_count[0];
  const unsigned total = M + (unsigned)a.nslots;
  const unsigned base = blockIdx.x * kScanChunk;
  unsigned running;
  if (block_count) {
    // small grids: every block sums the flag counts of the blocks before it itself (and block 0 the
    // grand total = new map size), which saves the scan launch between the two clean kernels
    unsigned below = 0, all = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
      const unsigned c = block_count[b];
      all += c;
      below += b < blockIdx.x ? c : 0u;
    }
    running = block_sum_u32(below);
    if (blockIdx.x == 0) {
      const unsigned tot = block_sum_u32(all);
      if (threadIdx.x == 0) {
        const unsigned c = tot < (unsigned)cap ? tot : (unsigned)cap;
        count_new[0] = c;
        if (count_new2) count_new2[0] = c;
      }
    }
  } else {
    running = block_offset[blockIdx.x];
  }
  if (base >= total) return;
  for (int k = 0; k < kScanChunk / 256; ++k) {
    const unsigned e = base + k * 256 + threadIdx.x;
    const bool f = (e < total) && keep[e];
    unsigned tot;
    const unsigned rank = block_exclusive_rank(f, tot);
    if (f) {
      const size_t dst = (size_t)running + rank;
      if (dst < cap) {
        CleanElem v;
        clean_load(e, M, sp, cap, slot_pos, slot_col, slot_nrm, slot_flag, a.nslots, a.timeIdx, a.planes, v);
        if (v.vt == -2.f) {  // copy_unstable.vert:124-129
          v.col.w = (float)a.time;
          v.vt = (float)a.time;
        }
        if (a.nodes > 0 && v.col.z != (float)a.time) clean_deform(a, v);  // :161
        out.pos[dst] = to4(v.pos);
        out.col[dst] = to4(v.col);
        out.nrm[dst] = to4(v.nrm);
#pragma unroll
        for (int s = 0; s < DMS_MAX_SENSORS; ++s)
          if (s < a.planes) out.times[(size_t)s * cap + dst] = (s == a.timeIdx) ? v.vt : v.times[s];  // (the other planes hold -3 already)
      }
    }
    // the parked measurement of this element is consumed (a later clean without a fuse must not
    // re-append it): only this thread ever reads the flag in this kernel
    if (e >= M && e < total) slot_flag[e - M] = 0;
    running += tot;
  }
}

// Suffix-mode scatter (large maps, no deformation graph): the blocks before first_touched[0] hold map surfels
// none of which (nor any before them) is removed — they map onto themselves, unchanged, and are not visited;
// a fixed grid walks the rest and stages the survivors in `out` (k_clean_copy_back returns them).
__global__ __launch_bounds__(256) void k_clean_scatter_suffix(CleanArgs a, SurfelPlanes sp, size_t cap, const unsigned* __restrict__ d_count,
                                                              const float4* slot_pos, const float4* slot_col, const float4* slot_nrm,
                                                              unsigned char* slot_flag, const unsigned char* __restrict__ keep,
                                                              const unsigned* __restrict__ block_offset, SurfelPlanes out,
                                                              const unsigned* __restrict__ first_touched) {
  const unsigned M = d_count[0];
  const unsigned total = M + (unsigned)a.nslots;
  for (unsigned b = first_touched[0] + blockIdx.x; (size_t)b * kScanChunk < total; b += gridDim.x) {
    const unsigned e = b * kScanChunk + threadIdx.x;
    const bool f = (e < total) && keep[e];
    unsigned tot;
    const unsigned rank = block_exclusive_rank(f, tot);
    if (f) {
      const size_t dst = (size_t)block_offset[b] + rank;
      if (dst < cap) {
        CleanElem v;
        clean_load(e, M, sp, cap, slot_pos, slot_col, slot_nrm, slot_flag, a.nslots, a.timeIdx, a.planes, v);
        if (v.vt == -2.f) {  // copy_unstable.vert:124-129
          v.col.w = (float)a.time;
          v.vt = (float)a.time;
        }
        out.pos[dst] = to4(v.pos);
        out.col[dst] = to4(v.col);
        out.nrm[dst] = to4(v.nrm);
#pragma unroll
        for (int s = 0; s < DMS_MAX_SENSORS; ++s)
          if (s < a.planes) out.times[(size_t)s * cap + dst] = (s == a.timeIdx) ? v.vt : v.times[s];  // (the other planes hold -3 already)
      }
    }
    if (e >= M && e < total) slot_flag[e - M] = 0;  // the parked measurement is consumed
  }
}

// ---------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------
static bool dense_img(const dms_image2d& im, size_t elem, int w, int h) {
  return im.data && im.cols == w && im.rows == h && im.pitch == (size_t)w * elem;
}

int model_fuse(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_image2d* rgba, const dms_image2d* dr,
               const dms_image2d* drf, const dms_indexmap_out* im, const dms_camera* cam, float depthCutoff, float weighting,
               const float* weighting_dev, int transposed, hipStream_t s, int defer_update) {
  DMS_REQUIRE(m && pose && rgba && dr && drf && im && cam, "null argument");
  DMS_REQUIRE(!m->pending_update, "a deferred update pass is still pending (index_map applies it)");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  if (m->live_planes < timeIdx + 1) m->live_planes = timeIdx + 1;  // (the update pass stamps times[timeIdx])
  const int W = m->width, H = m->height;
  DMS_REQUIRE(dense_img(*rgba, 4, W, H) && dense_img(*dr, 4, W, H) && dense_img(*drf, 4, W, H) && dense_img(im->index, 4, W, H) &&
                  dense_img(im->vertConf, 16, W, H) && dense_img(im->normRad, 16, W, H),
              "dense W×H images required");
  FuseArgs a;
  a.xcd = xcd_remap_enabled();
  a.pose = pose;
  a.rgba = (const uchar4*)rgba->data;
  a.dr = (const float*)dr->data;
  a.drf = (const float*)drf->data;
  a.index = (const unsigned*)im->index.data;
  a.vertConf = (const float4*)im->vertConf.data;
  a.normRad = (const float4*)im->normRad.data;
  a.cols = W;
  a.rows = H;
  a.slot_h = m->slot_h;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.icx = (float)(1.0 / (double)cam->fx);
  a.icy = (float)(1.0 / (double)cam->fy);
  a.maxDepth = depthCutoff;
  a.timef = (float)time;
  a.weighting = weighting;
  a.weighting_dev = weighting_dev;
  a.time = time;
  a.timeIdx = timeIdx;
  a.transposed = transposed ? 1 : 0;
  const int slot_w = (W + 1) / 2;
  const int tiles = ((m->slot_h + 7) / 8) * ((slot_w + 1) / 2);  // 2 x 8 candidates per wave, four lanes each
  dim3 b(256), g((tiles + 3) / 4);
  hipLaunchKernelGGL(k_fuse_associate, g, b, 0, s, a, m->slot_pos, m->slot_col, m->slot_nrm, m->slot_best, m->slot_flag, m->winner, slot_w);
  DMS_CHECK_LAUNCH();
  if (defer_update) {  // the caller's next index_map applies the winners while it projects: one pass and one launch less
    m->pending_update = true;
    m->pending_time = time;
    m->pending_timeIdx = timeIdx;
  } else {
    hipLaunchKernelGGL(k_fuse_update, dim3((m->slots + 255) / 256), dim3(256), 0, s, m->slots, m->slot_pos, m->slot_col, m->slot_nrm,
                       m->slot_best, m->slot_flag, m->winner, m->buf[m->cur], m->cap, time, timeIdx);
    DMS_CHECK_LAUNCH();
  }
  m->version += 1;
  return DMS_OK;
}

// A fuse with defer_update = 1 leaves the winners' update to the next index_map.  Every other reader of the surfel planes
// (download, export, consume, graph sampling) and a frame step that starts after an error between the two must not see the
// map half-updated: they apply the pending pass themselves first.
int model_flush_pending(dms_model* m, hipStream_t s) {
  if (!m || !m->pending_update) return DMS_OK;
  m->pending_update = false;
  hipLaunchKernelGGL(k_fuse_update, dim3((m->slots + 255) / 256), dim3(256), 0, s, m->slots, m->slot_pos, m->slot_col, m->slot_nrm, m->slot_best,
                     m->slot_flag, m->winner, m->buf[m->cur], m->cap, m->pending_time, m->pending_timeIdx);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int model_clean(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_indexmap_out* im, const dms_image2d* depth_synth,
                const dms_camera* cam, float confThreshold, const float* graph_host, int graph_nodes, int timeDelta, float maxDepth,
                int isFern, int transposed, unsigned* count_out2, hipStream_t s) {
  DMS_REQUIRE(m && !m->pending_update, "a deferred update pass is still pending (index_map applies it)");
  DMS_REQUIRE(m && pose && im && cam, "null argument");
  DMS_REQUIRE(timeIdx >= 0 && timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  const int W = m->width, H = m->height;
  DMS_REQUIRE(dense_img(im->index, 4, W, H) && dense_img(im->vertConf, 16, W, H) && dense_img(im->colorTime, 16, W, H),
              "dense W×H index-map images required");
  if (graph_nodes >= m->max_nodes) {  // assert(graph.size() / 16 < MAX_NODES), GlobalModel.cpp:703
    set_error("dms_model_clean: %d deformation nodes exceed the limit %d", graph_nodes, m->max_nodes);
    return DMS_ERR_CAPACITY;
  }
  if (graph_nodes > 0) {
    DMS_REQUIRE(graph_host, "null graph");
    DMS_HIP(hipMemcpyAsync(m->nodes, graph_host, (size_t)graph_nodes * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  }
  CleanArgs a;
  a.pose = pose;
  a.index = (const unsigned*)im->index.data;
  a.vertConf = (const float4*)im->vertConf.data;
  a.colorTime = (const float4*)im->colorTime.data;
  a.depth_synth = (depth_synth && depth_synth->data) ? (const float*)depth_synth->data : nullptr;
  a.cols = W;
  a.rows = H;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.fx = cam->fx;
  a.fy = cam->fy;
  a.confThreshold = confThreshold;
  a.maxDepth = maxDepth;
  a.time = time;
  a.timeIdx = timeIdx;
  a.timeDelta = timeDelta;
  a.isFern = isFern;
  a.nodes = graph_nodes;
  a.node_table = m->nodes;
  a.nslots = m->slots;
  a.transposed = transposed ? 1 : 0;
  const size_t upper = m->count_upper + (size_t)m->slots;
  const int nb = (int)((upper + kScanChunk - 1) / kScanChunk);
  // suffix mode pays one more launch: worth it once the map is large
  const bool suffix = graph_nodes == 0 && upper >= m->clean_suffix_min;  // (switch: dms_model_set_clean_suffix_min)
  a.suffix = suffix ? 1 : 0;
  a.num_sensors = m->num_sensors;
  if (m->live_planes < timeIdx + 1) m->live_planes = timeIdx + 1;
  static const bool all_planes = getenv("DMS_CLEAN_ALL_PLANES") != nullptr;  // A/B switch: read and move every plane, as before round 5
  a.planes = all_planes ? DMS_MAX_SENSORS : m->live_planes;
  const SurfelPlanes src = m->buf[m->cur], dst = m->buf[m->cur ^ 1];
  hipLaunchKernelGGL(k_clean_flags, dim3(nb), dim3(256), 0, s, a, src, m->cap, m->d_count, m->slot_pos, m->slot_col, m->slot_nrm,
                     m->slot_flag, m->keep, m->block_count, suffix ? m->clean_first : (unsigned*)nullptr);
  DMS_CHECK_LAUNCH();
  // (every scatter block sums the counts before it itself: nb^2 / 2 count reads from L2 against one launch; DMS_CLEAN_INLINE_SCAN_MAX)
  static const int inline_max = getenv("DMS_CLEAN_INLINE_SCAN_MAX") ? atoi(getenv("DMS_CLEAN_INLINE_SCAN_MAX")) : 2048;
  const bool inline_scan = nb <= inline_max && !suffix;  // suffix mode needs the block offsets in memory
  if (!inline_scan) {
    // (limit = the old count: only complete blocks of map surfels can stay in place)
    hipLaunchKernelGGL(k_scan_blocks_par, dim3((nb + 1023) / 1024), dim3(1024), 0, s, m->block_count, m->block_offset, nb, m->d_count_alt,
                       (unsigned)m->cap, count_out2, suffix ? m->clean_first : (unsigned*)nullptr, m->d_count, (unsigned)kScanChunk);
    DMS_CHECK_LAUNCH();
  }
  const int suffix_grid = nb < 2048 ? nb : 2048;  // suffix mode: blocks loop from the first touched block on
  if (suffix)
    hipLaunchKernelGGL(k_clean_scatter_suffix, dim3(suffix_grid), dim3(256), 0, s, a, src, m->cap, m->d_count, m->slot_pos, m->slot_col,
                       m->slot_nrm, m->slot_flag, m->keep, m->block_offset, dst, m->clean_first);
  else
    hipLaunchKernelGGL(k_clean_scatter, dim3(nb), dim3(256), 0, s, a, src, m->cap, m->d_count, m->slot_pos, m->slot_col, m->slot_nrm,
                       m->slot_flag, m->keep, m->block_offset, dst, inline_scan ? m->block_count : (const unsigned*)nullptr, m->d_count_alt,
                       count_out2);
  DMS_CHECK_LAUNCH();
  if (suffix) {
    hipLaunchKernelGGL(k_clean_copy_back, dim3(suffix_grid), dim3(256), 0, s, dst, src, m->cap, m->d_count_alt, m->clean_first, a.planes);
    DMS_CHECK_LAUNCH();
  }
  // the scatter still reads the old count cell; later launches get the new one
  std::swap(m->d_count, m->d_count_alt);
  if (!suffix) m->cur ^= 1;
  m->count_upper = upper < m->cap ? upper : m->cap;
  m->version += 1;
  return DMS_OK;
}

}  // namespace dms
