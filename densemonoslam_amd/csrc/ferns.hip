// Randomised-fern keyframe database (reference class Ferns, Core/src/Ferns.{h,cpp}) — see include/dmslam_ferns.h.
//
// MI355X design.  The reference keeps the database in host vectors and walks, per query, 500 inverted lists
// (`conservatory[i].ids[code]`) to count co-occurrences (Ferns.cpp:208-229) — a pointer chase that is the CPU form
// of "for every stored frame, how many of its codes equal mine".  Here the database is dense in HBM:
//   codes  [capacity][512] bytes (500 used, padded with the bad code), one 512-byte line per frame
//   blocks [capacity][dms_thumb_block_bytes] bytes: RGBA8 image (padded to 16 bytes) | RGBA32F vertex | RGBA32F normal thumbnails (what verification needs)
// and a query is three tiny launches on one stream: resize (skipped for a thumbnail block), encode (one block, one lane
// per fern), search (one wavefront per stored frame: 64 lanes x 8 codes, byte compares, DPP sum, one 64-bit atomicMin of
// {dissimilarity bits, frame id} — "first strictly smaller" of the reference's loop is "smallest key").  8 000 stored
// frames are 4 MB of codes: the search streams them once, ~1 us of HBM time.  The accept / reject decision and the
// 500-sample photometric check stay on the host as in the reference (they follow a tracker call that synchronises).
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/dmslam_ferns.h"
#include "internal.hpp"
#include "smallmath.hpp"
#include "surfel.hpp"

namespace dms {

constexpr int kFernPad = DMS_FERN_MAX;  // codes per stored frame (padded)

struct FernTable {  // device copy of the conservatory
  short x[kFernPad], y[kFernPad];
  int r[kFernPad], g[kFernPad], b[kFernPad], d[kFernPad];
};

struct Pose16f {
  float v[16];
};

struct FernHost {  // pinned result block of one query
  unsigned long long best;  // dissimilarity bits << 32 | frame id; all ones = none
  int good;                 // goodCodes of the query frame
  int hd_count, hd_equal;   // blockHDAware against the best frame
  int slot;                 // addFrame: database slot the staged frame went to, -1 = rejected, -2 = database full
  int n;                    // frames in the database after this operation
  int pad[3];
  float4 vert[kFernPad];    // vertSmall at the fern positions
  uchar4 rgb[kFernPad];     // imgSmall at the fern positions
};

// Resize::image / Resize::vertex (Shaders/Resize.cpp:67-143, resize.frag): NEAREST samples of the full-resolution
// textures at the thumbnail's pixel centres, packed as dms_fusion_thumbnails packs them
__global__ __launch_bounds__(256) void k_fern_thumbs(const uchar4* __restrict__ image, const float4* __restrict__ vertex,
                                                     const float4* __restrict__ normal, int cols, int rows, int tw, int th,
                                                     unsigned char* __restrict__ block) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= tw * th) return;
  const int j = k / tw, i = k - j * tw;
  const float u = ((float)i + 0.5f) / (float)tw, v = ((float)j + 0.5f) / (float)th;
  const int sx = texel(u, (float)cols, cols), sy = texel(v, (float)rows, rows);
  const size_t q = (size_t)sy * cols + sx;
  const size_t n = (size_t)tw * th;
  reinterpret_cast<uchar4*>(block)[k] = image[q];
  reinterpret_cast<float4*>(block + thumb_vertex_off(n))[k] = vertex[q];
  reinterpret_cast<float4*>(block + thumb_normal_off(n))[k] = normal[q];
}

// code of every fern (Ferns.cpp:208-233): one lane per fern; also the samples the host-side checks read
__global__ __launch_bounds__(kFernPad) void k_fern_encode(const unsigned char* __restrict__ block, int tw, int th, const FernTable* __restrict__ tab,
                                                          int num, unsigned char* __restrict__ codes, int* __restrict__ good,
                                                          FernHost* __restrict__ res, unsigned char* __restrict__ codes2 = nullptr) {
  __shared__ int s_good;
  const int i = threadIdx.x;
  if (i == 0) s_good = 0;
  __syncthreads();
  unsigned char code = DMS_FERN_BAD_CODE;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  uchar4 pix = make_uchar4(0, 0, 0, 0);
  if (i < num) {
    const size_t n = (size_t)tw * th;
    const int q = (int)tab->y[i] * tw + (int)tab->x[i];
    v = reinterpret_cast<const float4*>(block + thumb_vertex_off(n))[q];
    pix = reinterpret_cast<const uchar4*>(block)[q];
    if (v.z > 0.f) {
      code = (unsigned char)((((int)pix.x > tab->r[i]) << 3) | (((int)pix.y > tab->g[i]) << 2) | (((int)pix.z > tab->b[i]) << 1) |
                             (f2i_rz(v.z * 1000.0f) > tab->d[i] ? 1 : 0));
      atomicAdd(&s_good, 1);
    }
  }
  __syncthreads();
  if (codes) codes[i] = code;
  if (codes2) codes2[i] = code;  // (a second copy: the caller's descriptor beside the handle's staging area)
  if (good && i == 0) *good = s_good;
  if (res) {
    res->vert[i] = v;
    res->rgb[i] = pix;
    if (i == 0) {
      res->good = s_good;
      res->best = ~0ull;
      res->hd_count = 0;
      res->hd_equal = 0;
      res->slot = -1;
    }
  }
}

__device__ __forceinline__ int wave_sum_i(int v) {
  v = wave_sum_to_lane63_i(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// minimum dissimilarity over the stored frames (Ferns.cpp:235-248 / 327-339): one wavefront per frame
__global__ __launch_bounds__(256) void k_fern_search(const unsigned char* __restrict__ db_codes, const int* __restrict__ db_good,
                                                     const int* __restrict__ db_time, const int* __restrict__ n_dev,
                                                     const unsigned char* __restrict__ cur_codes,
                                                     const int* __restrict__ cur_good, int time, int all_frames,
                                                     unsigned long long* __restrict__ best) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= *n_dev) return;  // (the grid is sized from the host's upper bound of the count)
  const unsigned long long mine = reinterpret_cast<const unsigned long long*>(cur_codes)[lane];
  const unsigned long long theirs = reinterpret_cast<const unsigned long long*>(db_codes + (size_t)j * kFernPad)[lane];
  int co = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned m = (unsigned)(mine >> (8 * b)) & 0xFFu, t = (unsigned)(theirs >> (8 * b)) & 0xFFu;
    co += (m != DMS_FERN_BAD_CODE && m == t) ? 1 : 0;
  }
  co = wave_sum_i(co);
  if (lane != 0) return;
  if (!(all_frames || time - db_time[j] > 300)) return;
  const int g = *cur_good, gj = db_good[j];
  const float maxCo = (float)(g < gj ? g : gj);
  const float dissim = (maxCo - (float)co) / maxCo;
  if (dissim != dissim) return;  // 0 / 0: never "less than" in the reference's loop
  atomicMin(best, ((unsigned long long)__float_as_uint(dissim) << 32) | (unsigned)j);
}

// the same search for a batch of descriptors (collaborative mode: every other camera's frame block in the gathered
// buffer): blockIdx.y = query, its codes / good-code count at base + q * stride + codes_off / good_off; query `skip`
// (this rank's own block) is left alone.  best[q] must hold ~0 on entry.
__global__ __launch_bounds__(256) void k_fern_search_batch(const unsigned char* __restrict__ db_codes, const int* __restrict__ db_good,
                                                           const int* __restrict__ db_time, const int* __restrict__ n_dev,
                                                           const unsigned char* __restrict__ base, size_t stride, size_t codes_off,
                                                           size_t good_off, int skip, int time, int all_frames,
                                                           unsigned long long* __restrict__ best, unsigned long long* __restrict__ prev = nullptr,
                                                           unsigned long long* __restrict__ prev_out = nullptr) {
  const int q = blockIdx.y;
  // pipelined callers (prev != null): the previous call's results live in the OTHER word set — one block per query hands them
  // to mapped host memory and re-arms them for the next call, which searches into that set (no re-arm launch between two searches)
  if (prev && blockIdx.x == 0 && threadIdx.x == 0) {
    prev_out[q] = prev[q];
    prev[q] = ~0ull;
  }
  if (q == skip) return;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= *n_dev) return;
  const unsigned char* blk = base + (size_t)q * stride;
  const unsigned long long mine = reinterpret_cast<const unsigned long long*>(blk + codes_off)[lane];
  const unsigned long long theirs = reinterpret_cast<const unsigned long long*>(db_codes + (size_t)j * kFernPad)[lane];
  int co = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned m = (unsigned)(mine >> (8 * b)) & 0xFFu, t = (unsigned)(theirs >> (8 * b)) & 0xFFu;
    co += (m != DMS_FERN_BAD_CODE && m == t) ? 1 : 0;
  }
  co = wave_sum_i(co);
  if (lane != 0) return;
  if (!(all_frames || time - db_time[j] > 300)) return;
  const int g = *reinterpret_cast<const int*>(blk + good_off), gj = db_good[j];
  const float maxCo = (float)(g < gj ? g : gj);
  const float dissim = (maxCo - (float)co) / maxCo;
  if (dissim != dissim) return;
  atomicMin(best + q, ((unsigned long long)__float_as_uint(dissim) << 32) | (unsigned)j);
}

// hands the results of the previous batch search to `out` (mapped host memory: the caller reads them a frame later
// without a copy engine on the frame's stream) and re-arms the result words
__global__ void k_fern_best_rearm(unsigned long long* __restrict__ best, int count, unsigned long long* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  out[q] = best[q];
  best[q] = ~0ull;
}

// blockHDAware(query, best frame) (Ferns.cpp:684-704) for the frame the search chose, without a host round trip
__global__ __launch_bounds__(kFernPad) void k_fern_hd(const unsigned char* __restrict__ db_codes, const unsigned char* __restrict__ cur_codes,
                                                      int num, FernHost* __restrict__ res) {
  __shared__ int s_c, s_e;
  if (threadIdx.x == 0) s_c = s_e = 0;
  __syncthreads();
  const unsigned long long best = res->best;
  if (best == ~0ull) return;
  const int id = (int)(best & 0xFFFFFFFFull);
  const int i = threadIdx.x;
  if (i < num) {
    const unsigned char a = cur_codes[i], b = db_codes[(size_t)id * kFernPad + i];
    if (a != DMS_FERN_BAD_CODE && b != DMS_FERN_BAD_CODE) {
      atomicAdd(&s_c, 1);
      if (a == b) atomicAdd(&s_e, 1);
    }
  }
  __syncthreads();
  if (i == 0) {
    res->hd_count = s_c;
    res->hd_equal = s_e;
  }
}

// The first half of Ferns::findFrame for a batch of queries, up to the test that decides whether the tracker verifies at all
// (Ferns.cpp:327-342: minimum dissimilarity, then blockHDAware > 0.3 against the frame it chose): one block per query reads the
// search's result word and writes {candidate id or -1, dissimilarity bits, codes valid in both, of those equal} - the host (or another
// rank, the words travel inside the frame block) forms (double)(hd_equal / hd_count) > 0.3 exactly as find_common does.
__global__ __launch_bounds__(kFernPad) void k_fern_hd_batch(const unsigned char* __restrict__ db_codes, const unsigned char* __restrict__ base,
                                                            size_t stride, size_t codes_off, int num, unsigned long long* __restrict__ best,
                                                            int4* __restrict__ out, unsigned* __restrict__ mirror = nullptr, size_t mirror_off = 0,
                                                            int mirror_words = 0) {
  __shared__ int s_c, s_e;
  const int q = blockIdx.x;
  // (optional) this block's query block hands `mirror_words` words from byte offset mirror_off on to host-visible memory: the
  // pipelined session's per-tick metadata (tick, camera, pose, the PREVIOUS search's hit rows) rides this launch instead of one of its own
  if (mirror)
    for (int i = threadIdx.x; i < mirror_words; i += blockDim.x)
      mirror[(size_t)q * mirror_words + i] = reinterpret_cast<const unsigned*>(base + (size_t)q * stride + mirror_off)[i];
  const unsigned long long b = best[q];
  __syncthreads();
  if (threadIdx.x == 0) best[q] = ~0ull;  // (re-armed for the next search of this handle: no memset between two calls)
  if (b == ~0ull) {
    if (threadIdx.x == 0) out[q] = make_int4(-1, 0, 0, 0);
    return;
  }
  if (threadIdx.x == 0) s_c = s_e = 0;
  __syncthreads();
  const int id = (int)(b & 0xFFFFFFFFull);
  const int i = threadIdx.x;
  if (i < num) {
    const unsigned char a = base[(size_t)q * stride + codes_off + i], c = db_codes[(size_t)id * kFernPad + i];
    if (a != DMS_FERN_BAD_CODE && c != DMS_FERN_BAD_CODE) {
      atomicAdd(&s_c, 1);
      if (a == c) atomicAdd(&s_e, 1);
    }
  }
  __syncthreads();
  if (i == 0) out[q] = make_int4(id, (int)(unsigned)(b >> 32), s_c, s_e);
}

// dms_ferns_search_blocks_hd in ONE launch for databases of up to kPublishFusedMax key frames: one block per query - its 8 waves take
// the stored frames in turn (the minimum through one LDS word, as k_fern_publish searches), then the block forms the operands of
// blockHDAware against the frame it chose, and (optionally) mirrors its query block's tail to host-visible memory.  Same arithmetic
// as k_fern_search_batch + k_fern_hd_batch: the minimum of (dissimilarity bits << 32 | id) does not depend on the order of the visits.
__global__ __launch_bounds__(kFernPad) void k_fern_search_hd_small(const unsigned char* __restrict__ db_codes, const int* __restrict__ db_good,
                                                                   const int* __restrict__ db_time, const int* __restrict__ n_dev,
                                                                   const unsigned char* __restrict__ base, size_t stride, size_t codes_off,
                                                                   size_t good_off, int time, int all_frames, int num, int4* __restrict__ out,
                                                                   unsigned* __restrict__ mirror, size_t mirror_off, int mirror_words) {
  __shared__ unsigned long long s_best;
  __shared__ int s_c, s_e;
  const int q = blockIdx.x, i = threadIdx.x;
  const unsigned char* blk = base + (size_t)q * stride;
  if (mirror)
    for (int w = i; w < mirror_words; w += blockDim.x) mirror[(size_t)q * mirror_words + w] = reinterpret_cast<const unsigned*>(blk + mirror_off)[w];
  if (i == 0) {
    s_best = ~0ull;
    s_c = s_e = 0;
  }
  __syncthreads();
  const int n0 = *n_dev;
  {
    const int lane = i & 63, wave = i >> 6;
    const unsigned long long mine = reinterpret_cast<const unsigned long long*>(blk + codes_off)[lane];
    const int g = *reinterpret_cast<const int*>(blk + good_off);
    for (int j = wave; j < n0; j += kFernPad / 64) {
      const unsigned long long theirs = reinterpret_cast<const unsigned long long*>(db_codes + (size_t)j * kFernPad)[lane];
      int co = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned m = (unsigned)(mine >> (8 * b)) & 0xFFu, t = (unsigned)(theirs >> (8 * b)) & 0xFFu;
        co += (m != DMS_FERN_BAD_CODE && m == t) ? 1 : 0;
      }
      co = wave_sum_i(co);
      if (lane == 0 && (all_frames || time - db_time[j] > 300)) {
        const int gj = db_good[j];
        const float maxCo = (float)(g < gj ? g : gj);
        const float dissim = (maxCo - (float)co) / maxCo;
        if (dissim == dissim) atomicMin(&s_best, ((unsigned long long)__float_as_uint(dissim) << 32) | (unsigned)j);
      }
    }
  }
  __syncthreads();
  const unsigned long long b = s_best;
  if (b == ~0ull) {
    if (i == 0) out[q] = make_int4(-1, 0, 0, 0);
    return;
  }
  const int id = (int)(b & 0xFFFFFFFFull);
  if (i < num) {
    const unsigned char a = blk[codes_off + i], c = db_codes[(size_t)id * kFernPad + i];
    if (a != DMS_FERN_BAD_CODE && c != DMS_FERN_BAD_CODE) {
      atomicAdd(&s_c, 1);
      if (a == c) atomicAdd(&s_e, 1);
    }
  }
  __syncthreads();
  if (i == 0) out[q] = make_int4(id, (int)(unsigned)(b >> 32), s_c, s_e);
}

// addFrame's decision (Ferns.cpp:235-275) on the device: (minimum > threshold || empty) && goodCodes > 0 -> the
// staged frame takes slot n; its metadata is written here, its payload by k_fern_commit
// `status` (mapped host memory, may be null): {sequence number of this add, frames stored, frames dropped because the database
// was full} — the host polls it without a synchronisation (dms_ferns_status) and tightens its launch bound from it
__global__ void k_fern_decide(FernHost* __restrict__ res, int* __restrict__ n_dev, int capacity, float threshold, int srcTime,
                              const float* __restrict__ pose_dev, Pose16f pose_host, int* __restrict__ db_good, int* __restrict__ db_time,
                              float* __restrict__ db_pose, volatile int* status = nullptr, int seq = 0) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = *n_dev;
  float minimum = 3.402823466e+38F;
  if (res->good > 0 && res->best != ~0ull) minimum = __uint_as_float((unsigned)(res->best >> 32));
  int slot = -1;
  if ((minimum > threshold || n == 0) && res->good > 0) {
    if (n >= capacity) {
      slot = -2;
      n_dev[1] += 1;  // sticky: key frames lost to a full database (the reference's database grows without bound)
    } else {
      slot = n;
      db_good[n] = res->good;
      db_time[n] = srcTime;
      for (int i = 0; i < 16; ++i) db_pose[(size_t)n * 16 + i] = pose_dev ? pose_dev[i] : pose_host.v[i];
      *n_dev = n + 1;
    }
  }
  res->slot = slot;
  res->n = *n_dev;
  if (status) {
    status[1] = *n_dev;
    status[2] = n_dev[1];
    __threadfence_system();
    status[0] = seq;
  }
}

// payload of an accepted frame -> its slot
__global__ __launch_bounds__(256) void k_fern_commit(const unsigned char* __restrict__ cur_block, const unsigned char* __restrict__ cur_codes,
                                                     size_t block_bytes, unsigned char* __restrict__ db_blocks, unsigned char* __restrict__ db_codes_all,
                                                     const FernHost* __restrict__ res) {
  const int slot = res->slot;
  if (slot < 0) return;
  unsigned char* db_block = db_blocks + (size_t)slot * block_bytes;
  unsigned char* db_codes = db_codes_all + (size_t)slot * kFernPad;
  const size_t n16 = block_bytes / 16;
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n16; k += (size_t)blockDim.x * gridDim.x)
    reinterpret_cast<uint4*>(db_block)[k] = reinterpret_cast<const uint4*>(cur_block)[k];
  if (blockIdx.x == 0 && threadIdx.x < kFernPad / 16) reinterpret_cast<uint4*>(db_codes)[threadIdx.x] = reinterpret_cast<const uint4*>(cur_codes)[threadIdx.x];
}

// dms_ferns_publish_block in ONE launch for databases of up to kPublishFusedMax key frames: encode (k_fern_encode), search
// (k_fern_search: the block's 8 waves take the stored frames in turn, minimum through one LDS word), decision
// (k_fern_decide) and commit (k_fern_commit) are four dependent steps of one block's worth of work each; as four launches they
// cost four launch latencies on the frame's stream in collaborative mode.  Same arithmetic, same results.
constexpr int kPublishFusedMax = 512;
__global__ __launch_bounds__(kFernPad) void k_fern_publish(const unsigned char* __restrict__ block, int tw, int th, const FernTable* __restrict__ tab,
                                                           int num, unsigned char* __restrict__ cur_codes, unsigned char* __restrict__ codes2,
                                                           int* __restrict__ good_out, FernHost* __restrict__ res, const unsigned char* __restrict__ db_codes,
                                                           int* __restrict__ db_good, int* __restrict__ db_time, float* __restrict__ db_pose,
                                                           unsigned char* __restrict__ db_blocks, size_t block_bytes, int* __restrict__ n_dev, int capacity,
                                                           float threshold, int srcTime, const float* __restrict__ pose_dev, volatile int* status, int seq,
                                                           unsigned* __restrict__ mirror, size_t mirror_offset, int mirror_words) {
  __shared__ int s_good, s_slot;
  __shared__ unsigned long long s_best;
  __shared__ __attribute__((aligned(8))) unsigned char s_codes[kFernPad];
  const int i = threadIdx.x;
  if (i == 0) {
    s_good = 0;
    s_best = ~0ull;
  }
  __syncthreads();
  // ---- encode (Ferns.cpp:208-233)
  unsigned char code = DMS_FERN_BAD_CODE;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  uchar4 pix = make_uchar4(0, 0, 0, 0);
  if (i < num) {
    const size_t n = (size_t)tw * th;
    const int q = (int)tab->y[i] * tw + (int)tab->x[i];
    v = reinterpret_cast<const float4*>(block + thumb_vertex_off(n))[q];
    pix = reinterpret_cast<const uchar4*>(block)[q];
    if (v.z > 0.f) {
      code = (unsigned char)((((int)pix.x > tab->r[i]) << 3) | (((int)pix.y > tab->g[i]) << 2) | (((int)pix.z > tab->b[i]) << 1) |
                             (f2i_rz(v.z * 1000.0f) > tab->d[i] ? 1 : 0));
      atomicAdd(&s_good, 1);
    }
  }
  s_codes[i] = code;
  cur_codes[i] = code;
  if (codes2) codes2[i] = code;
  res->vert[i] = v;
  res->rgb[i] = pix;
  __syncthreads();
  const int good = s_good;
  // ---- search (Ferns.cpp:235-248), every stored frame
  const int n0 = *n_dev;
  {
    const int lane = i & 63, wave = i >> 6;
    const unsigned long long mine = reinterpret_cast<const unsigned long long*>(s_codes)[lane];
    for (int j = wave; j < n0; j += kFernPad / 64) {
      const unsigned long long theirs = reinterpret_cast<const unsigned long long*>(db_codes + (size_t)j * kFernPad)[lane];
      int co = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned m = (unsigned)(mine >> (8 * b)) & 0xFFu, t = (unsigned)(theirs >> (8 * b)) & 0xFFu;
        co += (m != DMS_FERN_BAD_CODE && m == t) ? 1 : 0;
      }
      co = wave_sum_i(co);
      if (lane == 0) {
        const int gj = db_good[j];
        const float maxCo = (float)(good < gj ? good : gj);
        const float dissim = (maxCo - (float)co) / maxCo;
        if (dissim == dissim) atomicMin(&s_best, ((unsigned long long)__float_as_uint(dissim) << 32) | (unsigned)j);
      }
    }
  }
  __syncthreads();
  // ---- decision (Ferns.cpp:235-275)
  if (i == 0) {
    const unsigned long long best = s_best;
    float minimum = 3.402823466e+38F;
    if (good > 0 && best != ~0ull) minimum = __uint_as_float((unsigned)(best >> 32));
    int slot = -1;
    if ((minimum > threshold || n0 == 0) && good > 0) {
      if (n0 >= capacity) {
        slot = -2;
        n_dev[1] += 1;
      } else {
        slot = n0;
        db_good[n0] = good;
        db_time[n0] = srcTime;
        for (int k = 0; k < 16; ++k) db_pose[(size_t)n0 * 16 + k] = pose_dev[k];
        *n_dev = n0 + 1;
      }
    }
    if (good_out) *good_out = good;
    res->good = good;
    res->best = best;
    res->hd_count = 0;
    res->hd_equal = 0;
    res->slot = slot;
    res->n = *n_dev;
    s_slot = slot;
    if (status) {
      status[1] = *n_dev;
      status[2] = n_dev[1];
      __threadfence_system();
      status[0] = seq;
    }
  }
  __syncthreads();
  // ---- optional: mirror_words dwords of the block from byte mirror_offset on, as they are now, to memory the host reads (the
  // session's per-tick metadata: good-code count - written above by thread 0 and taken from the register here -, tick, camera, pose, hit rows)
  if (mirror) {
    const unsigned* src = reinterpret_cast<const unsigned*>(block + mirror_offset);
    for (int w = i; w < mirror_words; w += kFernPad) mirror[w] = (good_out && (const void*)(src + w) == (const void*)good_out) ? (unsigned)good : src[w];
  }
  // ---- commit: payload of an accepted frame -> its slot
  const int slot = s_slot;
  if (slot < 0) return;
  unsigned char* db_block = db_blocks + (size_t)slot * block_bytes;
  const size_t n16 = block_bytes / 16;
  for (size_t k = i; k < n16; k += kFernPad) reinterpret_cast<uint4*>(db_block)[k] = reinterpret_cast<const uint4*>(block)[k];
  if (i < kFernPad / 16)
    reinterpret_cast<uint4*>(const_cast<unsigned char*>(db_codes) + (size_t)slot * kFernPad)[i] = reinterpret_cast<const uint4*>(s_codes)[i];
}

// ---- the table: mt19937 + uniform_int_distribution (Ferns.cpp:27-30,56,68-83) ------------------------------------
struct Mt19937 {
  uint32_t mt[624];
  int idx;
  explicit Mt19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // uniform integer in [a, b]: Lemire's nearly-divisionless mapping of one 32-bit draw (libstdc++ 11, uniform_int_dist.h)
  int uniform(int a, int b) {
    const uint32_t range = (uint32_t)(b - a) + 1u;
    uint64_t product = (uint64_t)next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (uint64_t)next() * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    return a + (int)(product >> 32);
  }
};

}  // namespace dms

using namespace dms;

struct dms_ferns {
  int num = 0, W = 0, H = 0, tw = 0, th = 0, maxDepth = 0, capacity = 0;
  int n_upper = 0;  // host-side upper bound of the number of stored frames (the count itself lives on the device)
  // asynchronous adds: {sequence number, stored, dropped} written by the decide kernel into mapped host memory; adds_issued counts
  // the adds enqueued so far, so that n_upper = stored(at sequence q) + (adds_issued - q) whenever a newer q has landed
  volatile int* h_status = nullptr;
  int* d_status = nullptr;  // device view of h_status
  int adds_issued = 0;
  // dms_ferns_search_blocks with a result mirror: the handle's alternate word set and the sequence it belongs to
  unsigned long long* d_best_alt = nullptr;
  unsigned long long* d_hd_best = nullptr;  // dms_ferns_search_blocks_hd's result words (armed once, re-armed by its second kernel)
  int hd_count = 0;
  int* pipe_best = nullptr;
  int pipe_count = 0, pipe_calls = 0;
  bool publish_fused = true;  // dms_ferns_publish_block as one launch while the database is small (DMS_FERNS_PUBLISH_FUSED=0: four)
  hipEvent_t ev_last_add = nullptr;  // recorded after the last asynchronous add: what mirror() has to wait for
  bool ev_valid = false;
  float photoThresh = 0.f;
  float cx = 0, cy = 0, fx = 0, fy = 0;  // full resolution
  std::vector<int> pos, rgbd;            // host table: [num][2], [num][4]
  std::vector<float> poses;              // [capacity][16]   host mirrors of the per-frame metadata,
  std::vector<int> times, goods;         //                  refreshed by mirror() (synchronises)
  int n_host = 0;                        // frames the mirrors hold
  int* d_n = nullptr;                    // number of stored frames
  float* d_pose = nullptr;               // [capacity][16]
  size_t block_bytes = 0;
  char* arena = nullptr;
  FernTable* d_tab = nullptr;
  unsigned char* d_codes = nullptr;   // [capacity][kFernPad]
  int* d_good = nullptr;              // [capacity]
  int* d_time = nullptr;              // [capacity]
  unsigned char* d_blocks = nullptr;  // [capacity][block_bytes]
  unsigned char* d_cur_block = nullptr;
  unsigned char* d_cur_codes = nullptr;  // [kFernPad]
  FernHost* d_res = nullptr;
  FernHost* h_res = nullptr;       // pinned
  unsigned char* h_rgb = nullptr;  // pinned: image thumbnail of the candidate (photometric check)
  dms_odometry* rgbd_odom = nullptr;
};

namespace {

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// After an asynchronous add has been enqueued on `s`: remember what mirror() must wait for, and tighten the host's bound of
// the frame count from the newest {sequence, stored} pair the decide kernels have written into mapped memory (a rejected
// frame no longer widens every later search by one frame's worth of blocks until the next synchronisation).
void after_async_add(dms_ferns* f, hipStream_t s) {
  f->ev_valid = hipEventRecord(f->ev_last_add, s) == hipSuccess;
  if (!f->ev_valid) (void)hipGetLastError();
  const int q = f->h_status[0];  // (sequence last: the pair below belongs to a sequence >= q)
  const int stored = f->h_status[1];
  int bound = q > 0 ? stored + (f->adds_issued - q) : f->n_upper + 1;
  if (bound > f->capacity) bound = f->capacity;
  f->n_upper = bound;
}

void mul44(const float* a, const float* b, float* o) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = a[i * 4 + 0] * b[0 * 4 + j];
      s += a[i * 4 + 1] * b[1 * 4 + j];
      s += a[i * 4 + 2] * b[2 * 4 + j];
      s += a[i * 4 + 3] * b[3 * 4 + j];
      o[i * 4 + j] = s;
    }
}
void mul4v(const float* m, float x, float y, float z, float* o) {
  for (int i = 0; i < 4; ++i) {
    float s = m[i * 4 + 0] * x;
    s += m[i * 4 + 1] * y;
    s += m[i * 4 + 2] * z;
    s += m[i * 4 + 3];
    o[i] = s;
  }
}

// resize (unless the frame arrives as a thumbnail block) + encode + reset of the result block
int stage(dms_ferns* f, const dms_image2d* image, const dms_image2d* vertex, const dms_image2d* normal, const void* block_dev, hipStream_t s) {
  const int n = f->tw * f->th;
  if (block_dev) {
    DMS_HIP(hipMemcpyAsync(f->d_cur_block, block_dev, f->block_bytes, hipMemcpyDeviceToDevice, s));
  } else {
    DMS_REQUIRE(image && vertex && normal && image->data && vertex->data && normal->data, "null texture");
    DMS_REQUIRE(image->rows == f->H && image->cols == f->W && image->pitch == (size_t)f->W * 4 && vertex->rows == f->H && vertex->cols == f->W &&
                    vertex->pitch == (size_t)f->W * 16 && normal->rows == f->H && normal->cols == f->W && normal->pitch == (size_t)f->W * 16,
                "dense full-resolution RGBA8 / RGBA32F textures required");
    hipLaunchKernelGGL(k_fern_thumbs, dim3((n + 255) / 256), dim3(256), 0, s, (const uchar4*)image->data, (const float4*)vertex->data,
                       (const float4*)normal->data, f->W, f->H, f->tw, f->th, f->d_cur_block);
    DMS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_fern_encode, dim3(1), dim3(kFernPad), 0, s, f->d_cur_block, f->tw, f->th, f->d_tab, f->num, f->d_cur_codes,
                     (int*)nullptr, f->d_res);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// dissimilarity search of the staged codes (grid from the host's bound of the count; the kernel reads the count itself)
int search_enqueue(dms_ferns* f, int time, int all_frames, bool with_hd, hipStream_t s) {
  if (f->n_upper > 0) {
    hipLaunchKernelGGL(k_fern_search, dim3((f->n_upper + 3) / 4), dim3(256), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n, f->d_cur_codes,
                       &f->d_res->good, time, all_frames, &f->d_res->best);
    DMS_CHECK_LAUNCH();
    if (with_hd) {
      hipLaunchKernelGGL(k_fern_hd, dim3(1), dim3(kFernPad), 0, s, f->d_codes, f->d_cur_codes, f->num, f->d_res);
      DMS_CHECK_LAUNCH();
    }
  }
  return DMS_OK;
}

int fetch_result(dms_ferns* f, hipStream_t s) {
  DMS_HIP(hipMemcpyAsync(f->h_res, f->d_res, sizeof(FernHost), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  return DMS_OK;
}

// addFrame on the staged frame, without a host synchronisation: search, decision and commit are stream ordered
int add_enqueue(dms_ferns* f, const float* pose16_host, const float* pose16_dev, int srcTime, float threshold, hipStream_t s) {
  int rc = search_enqueue(f, 0, 1, false, s);
  if (rc) return rc;
  Pose16f ph;
  memset(&ph, 0, sizeof(ph));
  if (pose16_host) memcpy(ph.v, pose16_host, sizeof(ph.v));
  f->adds_issued += 1;
  hipLaunchKernelGGL(k_fern_decide, dim3(1), dim3(64), 0, s, f->d_res, f->d_n, f->capacity, threshold, srcTime, pose16_dev, ph, f->d_good,
                     f->d_time, f->d_pose, (volatile int*)f->d_status, f->adds_issued);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fern_commit, dim3(64), dim3(256), 0, s, f->d_cur_block, f->d_cur_codes, f->block_bytes, f->d_blocks, f->d_codes,
                     f->d_res);
  DMS_CHECK_LAUNCH();
  after_async_add(f, s);
  return DMS_OK;
}

// host mirrors of count and per-frame metadata (synchronises)
int mirror(dms_ferns* f, hipStream_t s) {
  // frames may have been added on another stream (dms_ferns_add_frame_async): wait for the last such add — not for the
  // whole device, which would stall the other cameras' resident tracker kernels and the collectives
  if (f->ev_valid) DMS_HIP(hipEventSynchronize(f->ev_last_add));
  DMS_HIP(hipMemcpyAsync(&f->n_host, f->d_n, sizeof(int), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  const int n = f->n_host;
  f->n_upper = n;
  if (n > 0) {
    DMS_HIP(hipMemcpyAsync(f->goods.data(), f->d_good, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    DMS_HIP(hipMemcpyAsync(f->times.data(), f->d_time, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    DMS_HIP(hipMemcpyAsync(f->poses.data(), f->d_pose, (size_t)n * 16 * sizeof(float), hipMemcpyDeviceToHost, s));
    DMS_HIP(hipStreamSynchronize(s));
  }
  return DMS_OK;
}

int add_result(dms_ferns* f, int* added, hipStream_t s) {
  int rc = fetch_result(f, s);
  if (rc) return rc;
  f->n_upper = f->h_res->n;  // exact again
  *added = f->h_res->slot >= 0 ? 1 : 0;
  if (f->h_res->slot == -2) {
    set_error("dms_ferns: the database is full (%d frames)", f->capacity);
    return DMS_ERR_CAPACITY;
  }
  return DMS_OK;
}

// Ferns::photometricCheck (Ferns.cpp:604-668) on the 500 fern samples
float photometric_check(const dms_ferns* f, const float* estPose, const float* fernPose, const unsigned char* fernRgba) {
  const FernHost* r = f->h_res;
  const int factor = 8;
  const float cx = f->cx / factor, cy = f->cy / factor;
  const float invfx = 1.0f / (float)(f->fx / factor), invfy = 1.0f / (float)(f->fy / factor);
  float inv[16], diff[16];
  sm::inv4t<float>(fernPose, inv);
  mul44(inv, estPose, diff);
  float photoSum = 0.f;
  int photoCount = 0;
  for (int i = 0; i < f->num; ++i) {
    const float4 v = r->vert[i];
    if (v.z > 0.f && (int)(v.z * 1000.0f) < f->maxDepth) {
      float w[4];
      mul4v(diff, v.x, v.y, v.z, w);
      const int c0 = (int)(w[0] * (1 / invfx) / w[2] + cx), c1 = (int)(w[1] * (1 / invfy) / w[2] + cy);
      if (c0 >= 0 && c1 >= 0 && c0 < f->tw && c1 < f->th) {
        const unsigned char* p = fernRgba + ((size_t)c1 * f->tw + c0) * 4;
        if (p[0] > 0 || p[1] > 0 || p[2] > 0) {
          const uchar4 q = r->rgb[i];
          photoSum += (float)abs((int)p[0] - (int)q.x);
          photoSum += (float)abs((int)p[1] - (int)q.y);
          photoSum += (float)abs((int)p[2] - (int)q.z);
          photoCount++;
        }
      }
    }
  }
  return photoSum / (float)photoCount;
}

int find_common(dms_ferns* f, const float* currPose16, int time, int lost, int interMap, dms_fern_match* m, float* constraints, hipStream_t s) {
  (void)lost;
  int rc;
  memset(m, 0, sizeof(*m));
  m->closest = -1;
  m->candidate = -1;
  for (int i = 0; i < 16; ++i) m->estPose[i] = (i % 5 == 0) ? 1.f : 0.f;
  if ((rc = search_enqueue(f, time, interMap ? 1 : 0, true, s))) return rc;
  if ((rc = fetch_result(f, s))) return rc;
  const FernHost* r = f->h_res;
  if (r->best == ~0ull) return DMS_OK;
  const int minId = (int)(r->best & 0xFFFFFFFFull);
  const unsigned bits = (unsigned)(r->best >> 32);
  memcpy(&m->dissimilarity, &bits, 4);
  m->candidate = minId;
  m->blockHDAware = (float)r->hd_equal / (float)r->hd_count;
  if (!((double)m->blockHDAware > 0.3)) return DMS_OK;  // Ferns.cpp:346: float against the double literal (0.3f = 150 / 500 passes)

  // geometric verification with the thumbnail-sized tracker (Ferns.cpp:344-381)
  float fernPose[16];
  DMS_HIP(hipMemcpyAsync(fernPose, f->d_pose + (size_t)minId * 16, sizeof(fernPose), hipMemcpyDeviceToHost, s));
  DMS_HIP(hipStreamSynchronize(s));
  const size_t n = (size_t)f->tw * f->th;
  const unsigned char* fb = f->d_blocks + (size_t)minId * f->block_bytes;
  const float cutoff = (float)f->maxDepth / 1000.0f;
  if ((rc = dms_odometry_initICPModel(f->rgbd_odom, (const float*)(fb + thumb_vertex_off(n)), (const float*)(fb + thumb_normal_off(n)), cutoff, fernPose, s))) return rc;
  if ((rc = dms_odometry_initICP_maps(f->rgbd_odom, (const float*)(f->d_cur_block + thumb_vertex_off(n)), (const float*)(f->d_cur_block + thumb_normal_off(n)), cutoff, s)))
    return rc;
  DMS_HIP(hipMemcpyAsync(f->h_rgb, fb, n * 4, hipMemcpyDeviceToHost, s));
  float trans[3] = {fernPose[3], fernPose[7], fernPose[11]}, rot[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) rot[i * 3 + j] = fernPose[i * 4 + j];
  dms_track_result tr;
  // interMap == 1: the reference's inter-map settings (pyramid, SO3, 50 iterations per level); interMap == 2: every frame eligible as
  // for 1, verified with the intra-map settings (one level, 10 iterations) - see dmslam_ferns.h
  const int deep = interMap == 1 ? 1 : 0;
  if ((rc = dms_odometry_getIncrementalTransformation(f->rgbd_odom, trans, rot, 0, 100.0f, deep, 0, deep, deep, &tr, s)))
    return rc;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) m->estPose[i * 4 + j] = rot[i * 3 + j];
    m->estPose[i * 4 + 3] = trans[i];
  }
  m->icp_error = tr.lastICPError;
  m->icp_count = tr.lastICPCount;
  m->photo_error = photometric_check(f, m->estPose, fernPose, f->h_rgb);
  if (tr.lastICPError < 0.0003f && tr.lastICPCount > 400.f && m->photo_error < f->photoThresh) {
    m->closest = minId;
    const int step = f->num / 50 > 0 ? f->num / 50 : 1;
    for (int i = 0; i < f->num && m->n_constraints < 64; i += step) {
      const float4 v = r->vert[i];
      if (v.z > 0.f && (int)(v.z * 1000.0f) < f->maxDepth) {
        if (constraints) {
          float* c = constraints + (size_t)m->n_constraints * 8;
          mul4v(currPose16, v.x, v.y, v.z, c);
          mul4v(m->estPose, v.x, v.y, v.z, c + 4);
        }
        m->n_constraints += 1;
      }
    }
  }
  return DMS_OK;
}

}  // namespace

extern "C" {

int dms_ferns_create(dms_ferns** out, int num, int maxDepth_mm, float photoThresh, int width, int height, float cx, float cy, float fx,
                     float fy, unsigned int seed, int capacity) {
  DMS_REQUIRE(out, "null out");
  DMS_REQUIRE(num >= 1 && num <= DMS_FERN_MAX, "1 <= num <= DMS_FERN_MAX");
  DMS_REQUIRE(width >= 128 && height >= 128, "resolution at least 128 x 128 (thumbnails are width / 8 x height / 8, Ferns.cpp:23-24)");
  DMS_REQUIRE(maxDepth_mm >= 400, "maxDepth below the depth-threshold range [400, maxDepth] (Ferns.cpp:30)");
  DMS_REQUIRE(capacity >= 1, "capacity");
  dms_ferns* f = new dms_ferns();
  f->num = num;
  f->W = width;
  f->H = height;
  f->tw = width / 8;
  f->th = height / 8;
  f->maxDepth = maxDepth_mm;
  f->photoThresh = photoThresh;
  f->capacity = capacity;
  f->cx = cx;
  f->cy = cy;
  f->fx = fx;
  f->fy = fy;
  f->block_bytes = dms_thumb_block_bytes(width, height);
  // generateFerns (Ferns.cpp:66-84)
  Mt19937 rng(seed);
  f->pos.resize((size_t)num * 2);
  f->rgbd.resize((size_t)num * 4);
  FernTable tab;
  memset(&tab, 0, sizeof(tab));
  for (int i = 0; i < num; ++i) {
    f->pos[i * 2 + 0] = rng.uniform(0, f->tw - 1);
    f->pos[i * 2 + 1] = rng.uniform(0, f->th - 1);
    f->rgbd[i * 4 + 0] = rng.uniform(0, 255);
    f->rgbd[i * 4 + 1] = rng.uniform(0, 255);
    f->rgbd[i * 4 + 2] = rng.uniform(0, 255);
    f->rgbd[i * 4 + 3] = rng.uniform(400, maxDepth_mm);
    tab.x[i] = (short)f->pos[i * 2 + 0];
    tab.y[i] = (short)f->pos[i * 2 + 1];
    tab.r[i] = f->rgbd[i * 4 + 0];
    tab.g[i] = f->rgbd[i * 4 + 1];
    tab.b[i] = f->rgbd[i * 4 + 2];
    tab.d[i] = f->rgbd[i * 4 + 3];
  }
  f->poses.assign((size_t)capacity * 16, 0.f);
  f->times.assign(capacity, 0);
  f->goods.assign(capacity, 0);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = up256(off);
    const size_t at = off;
    off += bytes;
    return at;
  };
  const size_t o_tab = take(sizeof(FernTable)), o_codes = take((size_t)capacity * kFernPad), o_good = take((size_t)capacity * 4),
               o_time = take((size_t)capacity * 4), o_blocks = take((size_t)capacity * f->block_bytes), o_cur = take(f->block_bytes),
               o_cc = take(kFernPad), o_res = take(sizeof(FernHost)), o_n = take(64), o_pose = take((size_t)capacity * 64);
  hipError_t e = hipMalloc((void**)&f->arena, up256(off));
  if (e == hipSuccess) e = hipMemset(f->arena, 0, up256(off));
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->h_res, sizeof(FernHost), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->h_rgb, (size_t)f->tw * f->th * 4, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->h_status, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    memset((void*)f->h_status, 0, 64);
    e = hipHostGetDevicePointer((void**)&f->d_status, (void*)f->h_status, 0);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_last_add, hipEventDisableTiming);
  if (e != hipSuccess) {
    if (f->arena) (void)hipFree(f->arena);
    if (f->h_res) (void)hipHostFree(f->h_res);
    delete f;
    return hip_fail(e, "dms_ferns_create allocation", __FILE__, __LINE__);
  }
  f->d_tab = (FernTable*)(f->arena + o_tab);
  f->d_codes = (unsigned char*)(f->arena + o_codes);
  f->d_good = (int*)(f->arena + o_good);
  f->d_time = (int*)(f->arena + o_time);
  f->d_blocks = (unsigned char*)(f->arena + o_blocks);
  f->d_cur_block = (unsigned char*)(f->arena + o_cur);
  f->d_cur_codes = (unsigned char*)(f->arena + o_cc);
  f->d_res = (FernHost*)(f->arena + o_res);
  f->d_n = (int*)(f->arena + o_n);
  f->d_pose = (float*)(f->arena + o_pose);
  e = hipMemcpy(f->d_tab, &tab, sizeof(tab), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(f->d_codes, DMS_FERN_BAD_CODE, (size_t)capacity * kFernPad);
  int rc = e == hipSuccess ? DMS_OK : hip_fail(e, "dms_ferns_create upload", __FILE__, __LINE__);
  // the verification tracker at thumbnail size (Ferns.cpp:36-41: intrinsics / factor)
  if (!rc) rc = dms_odometry_create(&f->rgbd_odom, f->tw, f->th, cx / 8, cy / 8, fx / 8, fy / 8, 0.f, 0.f);
  if (rc) {
    dms_ferns_destroy(f);
    return rc;
  }
  if (const char* pf = getenv("DMS_FERNS_PUBLISH_FUSED")) f->publish_fused = atoi(pf) != 0;
  *out = f;
  return DMS_OK;
}

int dms_ferns_destroy(dms_ferns* f) {
  if (!f) return DMS_OK;
  if (f->rgbd_odom) dms_odometry_destroy(f->rgbd_odom);
  if (f->arena) (void)hipFree(f->arena);
  if (f->d_best_alt) (void)hipFree(f->d_best_alt);
  if (f->d_hd_best) (void)hipFree(f->d_hd_best);
  if (f->h_status) (void)hipHostFree((void*)f->h_status);
  if (f->ev_last_add) (void)hipEventDestroy(f->ev_last_add);
  if (f->h_res) (void)hipHostFree(f->h_res);
  if (f->h_rgb) (void)hipHostFree(f->h_rgb);
  delete f;
  return DMS_OK;
}

int dms_ferns_get_table(dms_ferns* f, int* pos2, int* rgbd4) {
  DMS_REQUIRE(f && pos2 && rgbd4, "null argument");
  memcpy(pos2, f->pos.data(), f->pos.size() * sizeof(int));
  memcpy(rgbd4, f->rgbd.data(), f->rgbd.size() * sizeof(int));
  return DMS_OK;
}

int dms_ferns_status(dms_ferns* f, int* stored, int* dropped) {
  DMS_REQUIRE(f, "null argument");
  // no synchronisation: what the newest completed asynchronous add has reported (mapped host memory)
  if (stored) *stored = f->h_status[1];
  if (dropped) *dropped = f->h_status[2];
  return DMS_OK;
}

int dms_ferns_num_frames(dms_ferns* f) {
  if (!f) return 0;
  if (mirror(f, nullptr)) return -1;
  return f->n_host;
}

int dms_ferns_get_frame(dms_ferns* f, int id, float* pose16, int* srcTime, int* goodCodes, unsigned char* codes) {
  DMS_REQUIRE(f, "null argument");
  int rc = mirror(f, nullptr);
  if (rc) return rc;
  DMS_REQUIRE(id >= 0 && id < f->n_host, "bad frame id");
  if (pose16) memcpy(pose16, &f->poses[(size_t)id * 16], 16 * sizeof(float));
  if (srcTime) *srcTime = f->times[id];
  if (goodCodes) *goodCodes = f->goods[id];
  if (codes) DMS_HIP(hipMemcpy(codes, f->d_codes + (size_t)id * kFernPad, f->num, hipMemcpyDeviceToHost));
  return DMS_OK;
}

int dms_ferns_encode(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal, unsigned char* codes_dev,
                     int* good_dev, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  hipStream_t s = (hipStream_t)st;
  int rc = stage(f, image_rgba, vertex, normal, nullptr, s);
  if (rc) return rc;
  if (codes_dev) DMS_HIP(hipMemcpyAsync(codes_dev, f->d_cur_codes, kFernPad, hipMemcpyDeviceToDevice, s));
  if (good_dev) DMS_HIP(hipMemcpyAsync(good_dev, &f->d_res->good, sizeof(int), hipMemcpyDeviceToDevice, s));
  return DMS_OK;
}

int dms_ferns_encode_thumbs(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, dms_stream st) {
  DMS_REQUIRE(f && thumb_block_dev && codes_dev && good_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)thumb_block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  // straight from the caller's block into the caller's descriptor: nothing of the handle's staging area is touched
  hipLaunchKernelGGL(k_fern_encode, dim3(1), dim3(kFernPad), 0, (hipStream_t)st, (const unsigned char*)thumb_block_dev, f->tw, f->th, f->d_tab,
                     f->num, codes_dev, good_dev, (FernHost*)nullptr);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int dms_ferns_add_frame(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal, const float* pose16,
                        int srcTime, float threshold, int* added, dms_stream st) {
  DMS_REQUIRE(f && pose16 && added, "null argument");
  hipStream_t s = (hipStream_t)st;
  int rc = stage(f, image_rgba, vertex, normal, nullptr, s);
  if (!rc) rc = add_enqueue(f, pose16, nullptr, srcTime, threshold, s);
  if (rc) return rc;
  return add_result(f, added, s);
}

int dms_ferns_add_frame_async(dms_ferns* f, const dms_image2d* image_rgba, const dms_image2d* vertex, const dms_image2d* normal,
                              const void* thumb_block_dev, const float* pose16_host, const float* pose16_dev, int srcTime, float threshold,
                              dms_stream st) {
  DMS_REQUIRE(f && (pose16_host || pose16_dev), "null argument");
  hipStream_t s = (hipStream_t)st;
  int rc = stage(f, image_rgba, vertex, normal, thumb_block_dev, s);
  if (rc) return rc;
  return add_enqueue(f, pose16_host, pose16_dev, srcTime, threshold, s);
}

int dms_ferns_publish_block(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, const float* pose16_dev,
                            int srcTime, float threshold, dms_stream st) {
  return dms_ferns_publish_block_mirror(f, thumb_block_dev, codes_dev, good_dev, pose16_dev, srcTime, threshold, nullptr, 0, 0, st);
}

int dms_ferns_publish_block_mirror(dms_ferns* f, const void* thumb_block_dev, unsigned char* codes_dev, int* good_dev, const float* pose16_dev,
                                   int srcTime, float threshold, void* mirror, size_t mirror_offset, size_t mirror_bytes, dms_stream st) {
  DMS_REQUIRE(f && thumb_block_dev && codes_dev && good_dev && pose16_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)thumb_block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  DMS_REQUIRE(!mirror || (mirror_offset % 4 == 0 && mirror_bytes % 4 == 0 && ((uintptr_t)mirror & 3) == 0), "the mirrored range must be made of dwords");
  hipStream_t s = (hipStream_t)st;
  if (f->n_upper <= kPublishFusedMax && f->publish_fused) {  // the four steps in one launch (k_fern_publish)
    f->adds_issued += 1;
    hipLaunchKernelGGL(k_fern_publish, dim3(1), dim3(kFernPad), 0, s, (const unsigned char*)thumb_block_dev, f->tw, f->th, f->d_tab, f->num,
                       f->d_cur_codes, codes_dev, good_dev, f->d_res, (const unsigned char*)f->d_codes, f->d_good, f->d_time, f->d_pose, f->d_blocks,
                       f->block_bytes, f->d_n, f->capacity, threshold, srcTime, pose16_dev, (volatile int*)f->d_status, f->adds_issued,
                       (unsigned*)mirror, mirror_offset, (int)(mirror_bytes / 4));
    DMS_CHECK_LAUNCH();
    after_async_add(f, s);
    return DMS_OK;
  }
  // one encoding pass feeds the caller's descriptor and the handle's staged codes; the block is read where it lies
  hipLaunchKernelGGL(k_fern_encode, dim3(1), dim3(kFernPad), 0, s, (const unsigned char*)thumb_block_dev, f->tw, f->th, f->d_tab, f->num,
                     f->d_cur_codes, good_dev, f->d_res, codes_dev);
  DMS_CHECK_LAUNCH();
  int rc = search_enqueue(f, 0, 1, false, s);
  if (rc) return rc;
  Pose16f ph;
  memset(&ph, 0, sizeof(ph));
  f->adds_issued += 1;
  hipLaunchKernelGGL(k_fern_decide, dim3(1), dim3(64), 0, s, f->d_res, f->d_n, f->capacity, threshold, srcTime, pose16_dev, ph, f->d_good,
                     f->d_time, f->d_pose, (volatile int*)f->d_status, f->adds_issued);
  DMS_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_fern_commit, dim3(64), dim3(256), 0, s, (const unsigned char*)thumb_block_dev, f->d_cur_codes, f->block_bytes,
                     f->d_blocks, f->d_codes, f->d_res);
  DMS_CHECK_LAUNCH();
  after_async_add(f, s);
  if (mirror && mirror_bytes)  // (the database has outgrown the one-launch form: the mirror is a launch of its own)
    return dms_copy_rows_async(mirror, mirror_bytes, (const unsigned char*)thumb_block_dev + mirror_offset, mirror_bytes, mirror_bytes, 1, st);
  return DMS_OK;
}

int dms_ferns_find_frame(dms_ferns* f, const dms_image2d* vertex, const dms_image2d* normal, const dms_image2d* image_rgba,
                         const float* currPose16, int time, int lost, int interMap, dms_fern_match* match, float* constraints, dms_stream st) {
  DMS_REQUIRE(f && currPose16 && match, "null argument");
  hipStream_t s = (hipStream_t)st;
  int rc = stage(f, image_rgba, vertex, normal, nullptr, s);
  if (rc) return rc;
  return find_common(f, currPose16, time, lost, interMap, match, constraints, s);
}

int dms_ferns_find_frame_thumbs(dms_ferns* f, const void* thumb_block_dev, const float* currPose16, int time, int lost, int interMap,
                                dms_fern_match* match, float* constraints, dms_stream st) {
  DMS_REQUIRE(f && thumb_block_dev && currPose16 && match, "null argument");
  DMS_REQUIRE(((uintptr_t)thumb_block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  hipStream_t s = (hipStream_t)st;
  int rc = stage(f, nullptr, nullptr, nullptr, thumb_block_dev, s);
  if (rc) return rc;
  return find_common(f, currPose16, time, lost, interMap, match, constraints, s);
}

int dms_ferns_search_codes(dms_ferns* f, const unsigned char* codes_dev, const int* good_dev, int time, int interMap, int* best2_dev,
                           dms_stream st) {
  DMS_REQUIRE(f && codes_dev && good_dev && best2_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)codes_dev & 7) == 0 && ((uintptr_t)best2_dev & 7) == 0, "codes and result must be 8-byte aligned");
  hipStream_t s = (hipStream_t)st;
  // (the codes are read as 64 x 8 bytes: the caller's buffer holds DMS_FERN_MAX bytes, entries past num = the bad code)
  DMS_HIP(hipMemsetAsync(best2_dev, 0xFF, 8, s));
  if (f->n_upper > 0) {
    hipLaunchKernelGGL(k_fern_search, dim3((f->n_upper + 3) / 4), dim3(256), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n, codes_dev, good_dev,
                       time, interMap ? 1 : 0, (unsigned long long*)best2_dev);
    DMS_CHECK_LAUNCH();
  }
  return DMS_OK;
}

int dms_ferns_search_blocks(dms_ferns* f, const void* blocks_dev, size_t stride, int count, int skip, size_t codes_offset, size_t good_offset,
                            int time, int interMap, int* best2_dev, int* previous_out, dms_stream st) {
  DMS_REQUIRE(f && blocks_dev && best2_dev && count >= 1, "bad argument");
  DMS_REQUIRE(((uintptr_t)blocks_dev & 7) == 0 && (stride & 7) == 0 && (codes_offset & 7) == 0 && (good_offset & 3) == 0 &&
                  ((uintptr_t)best2_dev & 7) == 0,
              "8-byte aligned blocks, stride, code offset and result required");
  hipStream_t s = (hipStream_t)st;
  if (previous_out) {
    DMS_REQUIRE(((uintptr_t)previous_out & 7) == 0, "8-byte aligned result mirror required");
    // Two word sets alternate (the caller's and one of the handle's): this call searches into the set the previous call's
    // kernel re-armed and hands the other set's results over inside the same launch.  The first call of a sequence (or a change of
    // the caller's buffer / count) arms both the old way.
    if (f->pipe_best == best2_dev && f->pipe_count == count && f->d_best_alt) {
      unsigned long long* cur = (f->pipe_calls & 1) ? f->d_best_alt : (unsigned long long*)best2_dev;
      unsigned long long* other = (f->pipe_calls & 1) ? (unsigned long long*)best2_dev : f->d_best_alt;
      f->pipe_calls += 1;
      const int nu = f->n_upper > 0 ? f->n_upper : 1;
      hipLaunchKernelGGL(k_fern_search_batch, dim3((nu + 3) / 4, count), dim3(256), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n,
                         (const unsigned char*)blocks_dev, stride, codes_offset, good_offset, skip, time, interMap ? 1 : 0, cur, other,
                         (unsigned long long*)previous_out);
      DMS_CHECK_LAUNCH();
      return DMS_OK;
    }
    if (!f->d_best_alt || f->pipe_count != count) {
      if (f->d_best_alt) (void)hipFree(f->d_best_alt);
      f->d_best_alt = nullptr;
      DMS_HIP(hipMalloc((void**)&f->d_best_alt, (size_t)count * 8));
    }
    DMS_HIP(hipMemsetAsync(f->d_best_alt, 0xFF, (size_t)count * 8, s));
    f->pipe_best = best2_dev;
    f->pipe_count = count;
    f->pipe_calls = 1;  // this call searches into the caller's set (armed below); the next one into the handle's
    hipLaunchKernelGGL(k_fern_best_rearm, dim3((count + 63) / 64), dim3(64), 0, s, (unsigned long long*)best2_dev, count,
                       (unsigned long long*)previous_out);
    DMS_CHECK_LAUNCH();
  } else {
    DMS_HIP(hipMemsetAsync(best2_dev, 0xFF, (size_t)count * 8, s));
  }
  if (f->n_upper > 0) {
    hipLaunchKernelGGL(k_fern_search_batch, dim3((f->n_upper + 3) / 4, count), dim3(256), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n,
                       (const unsigned char*)blocks_dev, stride, codes_offset, good_offset, skip, time, interMap ? 1 : 0,
                       (unsigned long long*)best2_dev);
    DMS_CHECK_LAUNCH();
  }
  return DMS_OK;
}

int dms_ferns_search_blocks_hd(dms_ferns* f, const void* blocks_dev, size_t stride, int count, size_t codes_offset, size_t good_offset, int time,
                               int interMap, int* hits4_dev, dms_stream st) {
  return dms_ferns_search_blocks_hd_mirror(f, blocks_dev, stride, count, codes_offset, good_offset, time, interMap, hits4_dev, nullptr, 0, 0, st);
}

int dms_ferns_search_blocks_hd_mirror(dms_ferns* f, const void* blocks_dev, size_t stride, int count, size_t codes_offset, size_t good_offset,
                                      int time, int interMap, int* hits4_dev, void* mirror, size_t mirror_offset, size_t mirror_bytes,
                                      dms_stream st) {
  DMS_REQUIRE(f && blocks_dev && hits4_dev && count >= 1, "bad argument");
  DMS_REQUIRE(!mirror || ((((uintptr_t)mirror | mirror_offset | mirror_bytes) & 3) == 0 && mirror_bytes >= 4), "4-byte aligned mirror, offset and size required");
  DMS_REQUIRE(((uintptr_t)blocks_dev & 7) == 0 && (stride & 7) == 0 && (codes_offset & 7) == 0 && (good_offset & 3) == 0 && ((uintptr_t)hits4_dev & 15) == 0,
              "8-byte aligned blocks, stride and code offset, 16-byte aligned hit rows required");
  hipStream_t s = (hipStream_t)st;
  if (f->n_upper <= kPublishFusedMax && f->publish_fused) {  // a small database: search + test (+ mirror) in one launch
    hipLaunchKernelGGL(k_fern_search_hd_small, dim3(count), dim3(kFernPad), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n,
                       (const unsigned char*)blocks_dev, stride, codes_offset, good_offset, time, interMap ? 1 : 0, f->num, (int4*)hits4_dev,
                       (unsigned*)mirror, mirror_offset, (int)(mirror_bytes / 4));
    DMS_CHECK_LAUNCH();
    return DMS_OK;
  }
  if (f->hd_count < count) {  // the result words of this entry point: armed once, re-armed by k_fern_hd_batch after every read
    if (f->d_hd_best) (void)hipFree(f->d_hd_best);
    f->d_hd_best = nullptr;
    f->hd_count = 0;
    DMS_HIP(hipMalloc((void**)&f->d_hd_best, (size_t)count * 8));
    DMS_HIP(hipMemsetAsync(f->d_hd_best, 0xFF, (size_t)count * 8, s));
    f->hd_count = count;
  }
  if (f->n_upper > 0) {
    hipLaunchKernelGGL(k_fern_search_batch, dim3((f->n_upper + 3) / 4, count), dim3(256), 0, s, f->d_codes, f->d_good, f->d_time, f->d_n,
                       (const unsigned char*)blocks_dev, stride, codes_offset, good_offset, -1, time, interMap ? 1 : 0, f->d_hd_best);
    DMS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_fern_hd_batch, dim3(count), dim3(kFernPad), 0, s, f->d_codes, (const unsigned char*)blocks_dev, stride, codes_offset, f->num,
                     f->d_hd_best, (int4*)hits4_dev, (unsigned*)mirror, mirror_offset, (int)(mirror_bytes / 4));
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int dms_ferns_consume(dms_ferns* dst, dms_ferns* src, const float* T16, float threshold, int* added, dms_stream st) {
  DMS_REQUIRE(dst && src && T16 && added && dst != src, "bad argument");
  DMS_REQUIRE(dst->tw == src->tw && dst->th == src->th, "databases of different thumbnail size");
  hipStream_t s = (hipStream_t)st;
  *added = 0;
  int rc = mirror(src, s);
  if (rc) return rc;
  for (int j = 0; j < src->n_host; ++j) {
    float pose[16];
    mul44(T16, &src->poses[(size_t)j * 16], pose);  // frame->pose = relativeTransform * frame->pose (Ferns.cpp:164)
    rc = stage(dst, nullptr, nullptr, nullptr, src->d_blocks + (size_t)j * src->block_bytes, s);
    if (!rc) rc = add_enqueue(dst, pose, nullptr, src->times[j], threshold, s);
    int one = 0;
    if (!rc) rc = add_result(dst, &one, s);
    if (rc) return rc;
    *added += one;
  }
  return DMS_OK;
}


// ---- the database as records, for a merge across ranks (ReferenceFrame::consumeReferenceFrame, ReferenceFrame.h:125, when the two
// reference frames live in different processes): record = {thumbnail block | pose 16 floats | srcTime | 3 ints of padding}
size_t dms_ferns_record_bytes(dms_ferns* f) { return f ? up256(f->block_bytes + 80) : 0; }

int dms_ferns_export_records(dms_ferns* f, void* dst_dev, int max_count, int* count, dms_stream st) {
  DMS_REQUIRE(f && count && (dst_dev || max_count == 0), "null argument");
  hipStream_t s = (hipStream_t)st;
  int rc = mirror(f, s);
  if (rc) return rc;
  const int n = f->n_host < max_count ? f->n_host : max_count;
  const size_t rb = dms_ferns_record_bytes(f);
  for (int j = 0; j < n; ++j) {
    char* rec = (char*)dst_dev + (size_t)j * rb;
    DMS_HIP(hipMemcpyAsync(rec, f->d_blocks + (size_t)j * f->block_bytes, f->block_bytes, hipMemcpyDeviceToDevice, s));
    float meta[20] = {0};
    memcpy(meta, &f->poses[(size_t)j * 16], 64);
    memcpy(meta + 16, &f->times[j], 4);
    DMS_HIP(hipMemcpyAsync(rec + f->block_bytes, meta, 80, hipMemcpyHostToDevice, s));
    DMS_HIP(hipStreamSynchronize(s));  // (`meta` is a stack buffer)
  }
  *count = n;
  return DMS_OK;
}

int dms_ferns_consume_records(dms_ferns* dst, const void* records_dev, int count, const float* T16, float threshold, int* added, dms_stream st) {
  DMS_REQUIRE(dst && T16 && added && (records_dev || count == 0) && count >= 0, "bad argument");
  hipStream_t s = (hipStream_t)st;
  *added = 0;
  const size_t rb = dms_ferns_record_bytes(dst);
  for (int j = 0; j < count; ++j) {
    const char* rec = (const char*)records_dev + (size_t)j * rb;
    float meta[20], pose[16];
    DMS_HIP(hipMemcpyAsync(meta, rec + dst->block_bytes, 80, hipMemcpyDeviceToHost, s));
    DMS_HIP(hipStreamSynchronize(s));
    int srcTime;
    memcpy(&srcTime, meta + 16, 4);
    mul44(T16, meta, pose);  // frame->pose = relativeTransform * frame->pose (Ferns.cpp:164)
    int rc = stage(dst, nullptr, nullptr, nullptr, rec, s);
    if (!rc) rc = add_enqueue(dst, pose, nullptr, srcTime, threshold, s);
    int one = 0;
    if (!rc) rc = add_result(dst, &one, s);
    if (rc) return rc;
    *added += one;
  }
  return DMS_OK;
}

}  // extern "C"
