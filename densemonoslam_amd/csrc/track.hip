// Object layer: the tracker (reference RGBDOdometry, Utils/RGBDOdometry.{h,cpp}).
//
// MI355X design.  The reference runs ≈67 blocking device→host round trips per call (three
// kernel pairs + a 116-byte copy + a host Eigen solve per Gauss-Newton iteration,
// RGBDOdometry.cpp:425-586).  Here the whole coarse-to-fine loop is enqueued on one HIP
// stream with no host involvement.  Resident form (default): k_track_init, k_so3_level, one k_gn_level per pyramid
// level — ALL iterations of a level in one launch, grid-wide sums through memory-side integer atomics.  Launch-per-phase
// form (DMS_TRACK_MODE=launches; also every level with more than 10 iterations):
//
//   k_track_init     1 lane   prior pose -> state, first-iteration projection parameters
//   k_so3_pass       grid     per-pixel SO3 rows -> per-block integer records; the last block folds and solves
//   k_gn_pass1<I,R>  grid     RGB correspondence (count / Σdiff²) and ICP rows -> records
//   k_gn_pass2       grid     σ from the folded count, photometric rows -> records
//   k_gn_solve       1 block  fold both record sets, fp64 6×6 LDLT, SE3 update, next KRK⁻¹ / Kt
//   k_track_finalize 1 lane   0.3 m jump gate, result block
//
// Every cross-pixel sum is the order-free integer sum of canon.hpp and the scalar section is the fixed operation sequence
// of gn_scalar.hpp, in both forms: a pose does not depend on the execution mode, the grid size or the run, and equals the
// CPU oracle's bit for bit.
//
// The pose, the 4×4 accumulated transform (fp64) and the early-exit flags (SO3
// converged/diverged, rgbOnly break) live in a device state block; kernels past an exit
// return at once, which reproduces the host `break`s.  One D2H copy of the result block
// (pinned) ends the call.
#include <hip/hip_ext.h>
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "internal.hpp"
#include "pixel_ops.hpp"
#include "smallmath.hpp"
#include "gn_scalar.hpp"
#include "canon.hpp"
#include "pyr_body.hpp"
#include "frame_state.hpp"
#include "track_init.hpp"
#include "model_bodies.hpp"
#include "surfel.hpp"
#include <mutex>

namespace dms {

// The tracker object evaluates the Gauss-Newton rows with fused multiply-adds (pixel_ops.hpp madd<true>; what nvcc's default
// -fmad=true does to the reference's kernels) in every execution mode; the operator layer (reduce.hip: dms_icpStep ...)
// keeps every operation rounded.  The oracle restates both forms (orc_set_fused_rows).
constexpr bool kTrackerFma = true;

struct Buf {
  void* p = nullptr;
  size_t pitch = 0;
  int rows = 0, cols = 0;
  dms_image2d img() const {
    dms_image2d i;
    i.data = p;
    i.pitch = pitch;
    i.rows = rows;
    i.cols = cols;
    return i;
  }
};

struct KernelTime {
  double ms = 0;
  int launches = 0;
  std::vector<float> samples;  // per launch, ms (profiling only; "<name>:slow" / ":max" / ":median" of dms_odometry_get_kernel_time)
};

}  // namespace dms

using namespace dms;

struct dms_odometry {
  bool early_exit = false;  // use the resident kernels that leave a level after an iteration without any correspondence
  // Execution switches, fixed per handle: read from the environment once, at dms_odometry_create (DMS_TRACK_MODE,
  // DMS_TRACK_EARLY_EXIT, DMS_PERSIST_BLOCKS), changed only through dms_odometry_set_mode.  (Reading them at
  // every call let a changing environment switch the execution mode in the middle of a session.)
  bool resident = true;       // false: three launches per iteration (DMS_TRACK_MODE=launches)
  int early_exit_force = -1;  // -1: as `early_exit`; 0 / 1: forced (DMS_TRACK_EARLY_EXIT)
  int persist_target = 160;   // largest grid that still gets 1 or 2 pixels per thread (DMS_PERSIST_BLOCKS)
  // Resident kernels spin on each other: ALL blocks of a launch must be on the device at once.  One block (512 threads x
  // up to 256 registers) fills a compute unit, so a grid may have at most as many blocks as the device has compute units
  // (hipDeviceProp_t::multiProcessorCount: 256 on a whole MI355X, fewer on a partition / under a CU mask), checked against the
  // occupancy API at creation; DMS_PERSIST_MAX_BLOCKS lowers it further.  A level that does not fit with <= 4 pixels per
  // thread runs launch-per-phase.
  int max_resident_blocks = 256;
  // dms_odometry_set_resident_budget: a cap below the device's own bound, and the owner's word that every handle that may track at the same
  // time carries a cap such that all their grids fit the device together - such a handle's launches need no chain (PersistSection)
  int budget_cap = 0;
  bool unchained_ok = false;
  bool fell_back = false;     // a resident kernel timed out at a grid-wide wait: this handle has switched to launch-per-phase
  // the model pyramid's last step (level 1 -> 2 of lastDepth / lastImage), left to the next track call's first kernel
  bool deferred_pyr = false;
  // the set-up of the next track call ran inside the model pyramid kernel (odometry_initModel_fused, fold_init): that call then
  // launches no kernel of its own before the SO3 level, which also carries the deferred pyramid step; the promised parameters
  bool init_folded = false;
  // round 6: the call's set-up enqueued AHEAD, as rider blocks of the frame's first kernel (odometry_early_init_args), so that the SO3
  // stage can run inside the model pyramid launch (k_so3_model) instead of after it
  bool early_init = false, so3_in_model = false;
  bool so3_beside_model = false;  // DMS_SO3_BESIDE_MODEL=1: SO3 stage and model pyramid in ONE launch (k_so3_model).  Built, the same bits, 7.6 us less kernel time per
                                  // frame - and no faster frame (the resident levels then wait for the prep stream's kernels instead): off
  int folded_so3 = 0, folded_first_level = 0;
  const float* folded_prior = nullptr;
  unsigned* dense_cnt_zero = nullptr;  // the frame step's 16 dense counters (fill.hpp), read by the model pyramid kernel: zeroed by the next track call's first kernel
  int inject_timeouts = 0;    // dms_odometry_inject_timeout: calls left that start with the timeout flag set
  int width, height;
  float cx, cy, fx, fy, distThres, angleThres;
  float sobelScale, maxDepthDeltaRGB, maxDepthRGB;
  float minGrad[DMS_NUM_PYRS];
  char* arena = nullptr;
  size_t arena_bytes = 0;
  Buf depth_tmp[3], vmaps_g_prev[3], nmaps_g_prev[3], vmaps_curr[3], nmaps_curr[3];
  Buf lastDepth[3], nextDepth[3], lastImage[3], nextImage[3], lastNextImage[3];
  Buf nextdIdx[3], nextdIdy[3], nextGate[3], pointClouds[3], corresImg[3];
  // Live-frame ring (frame pipeline only): the buffers written from the incoming frame alone —
  // depth pyramid, live vertex/normal maps, intensity pyramid and its derivatives — exist three
  // times, so that the next frame can be prepared on a second stream while this one is tracked
  // and the previous intensity pyramid still serves as lastNextImage.  Set 0 is the default set.
  struct LiveSet {
    Buf depth_tmp[3], vmaps_curr[3], nmaps_curr[3], nextImage[3], nextdIdx[3], nextdIdy[3], nextGate[3];
  };
  LiveSet ring[3];
  char* ring_arena = nullptr;
  bool ring_enabled = false;
  const void* deriv_of = nullptr;  // nextImage[0] the derivative pyramid was computed from
  float* vmaps_tmp = nullptr;
  float* nmaps_tmp = nullptr;
  long long* part_icp = nullptr;   // [kMaxPartialBlocks][32] integer records of the launch-per-phase kernels (grid units, canon.hpp)
  long long* part_rgb = nullptr;   // [kMaxPartialBlocks][32]
  long long* part_so3 = nullptr;   // [kMaxPartialBlocks][32]
  int* part_cnt = nullptr;     // [2][1024]
  unsigned* tickets = nullptr; // [4] arrival counters of the last-block-solves hand-off (zero between launches)
  unsigned long long* ar = nullptr;   // [kArSets][kArWords] all-reduce words of the resident kernels, zeroed by k_track_init
  int first_delay = -1;               // integer all-reduce: pause before the first read of the totals (DMS_AR_FIRST_DELAY; -1 = by grid size)
  // the later the last arrival can be after one's own, the longer the pause pays: ~0.4 us on the 150 / 200-block levels, next to
  // nothing on 38 blocks (measured: level 2 is best at 0 - 8 units, levels 1 and 0 at 12 - 20)
  bool long_levels_resident = true;   // levels of more than kArRing iterations (inter-map calls: 50) as resident launches too (DMS_TRACK_LONG_RESIDENT=0: launch-per-phase, as before round 6)
  bool fuse_coarse = false;           // SO3 + level 2 + level 1 in one resident launch (k_track_coarse; DMS_TRACK_FUSE=1).  Built and bit-identical, but SLOWER on the MI355X: off
  unsigned* rider_cnt = nullptr;      // device: rider blocks of k_track_coarse launches that have finished (runs over the life of the handle)
  unsigned rider_issued = 0;          // host: rider blocks enqueued so far
  int first_delay_lvl[4] = {-1, -1, -1, -1};  // per stage (levels 0, 1, 2, SO3): DMS_AR_FIRST_DELAY_BY_LEVEL="l0,l1,l2,so3" (-1 = the rule)
  int first_delay_for(int nb, int stage = -1) const {
    if (stage >= 0 && stage < 4 && first_delay_lvl[stage] >= 0) return first_delay_lvl[stage];
    return first_delay >= 0 ? first_delay : (nb >= 96 ? 24 : 8);
  }
  int depth_bias = 0;                 // production rule (key "depth_exp_bias"): the frame step's depth cut-off raises the static exponents (canon::depth_exp_bias)
  int exp_bias = 0;                   // test hook (dms_odometry_debug_set "exp_bias"): added to the static exponents of a call's first reductions (negative: they do not fit and are repeated)
  long long* prof = nullptr;          // [16] phase clocks of the persistent kernels (profiling only)
  TrackState* state = nullptr;
  TrackState* host_state = nullptr;  // pinned
  bool profiling = false;
  bool profiling_level0_only = false;  // dms_odometry_set_profiling(o, 2): one event pair per call, around the level-0 kernel, nothing else perturbed
  std::map<std::string, KernelTime> times;
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace dms {

__global__ void k_track_init(TrackState* st, Prior prior, const float* __restrict__ prior_pose16, float fx, float fy, float cx, float cy,
                             int so3, int first_level, unsigned long long* sync_words, int n_sync, int inject_timeout, unsigned* zero16) {
  track_init_body(blockIdx.x, gridDim.x, threadIdx.x, blockDim.x, st, prior, prior_pose16, fx, fy, cx, cy, so3, first_level, sync_words, n_sync,
                  inject_timeout, zero16);
}

// The same with the last step of the model-side depth / intensity pyramid (prep.hip k_model_pyr_step, deferred by
// odometry_initModel_fused) in further blocks: the two are independent, one launch boundary less per frame.
// 64 x 4 thread blocks; the first `ni` blocks are k_track_init's, the next gx * gy the pyramid step's.
__global__ __launch_bounds__(256) void k_track_init_pyr(TrackState* st, Prior prior, const float* __restrict__ prior_pose16, float fx, float fy,
                                                        float cx, float cy, int so3, int first_level, unsigned long long* sync_words,
                                                        int n_sync, int inject_timeout, unsigned* zero16, int gx, int gy, int ni, View<const float> dsrc,
                                                        View<float> ddst, View<const unsigned char> isrc, View<unsigned char> idst) {
  const int b = blockIdx.x;
  if (b >= ni) {
    const int c = b - ni, by = c / gx, bx = c - by * gx;
    model_pyr_step_pixel(bx * 64 + threadIdx.x, by * 4 + threadIdx.y, dsrc, ddst, isrc, idst);
    return;
  }
  // (first in the grid: block 0's one-lane state set-up is the longest dependency chain of the launch)
  track_init_body(b, ni, threadIdx.y * 64 + threadIdx.x, 256, st, prior, prior_pose16, fx, fy, cx, cy, so3, first_level, sync_words, n_sync,
                  inject_timeout, zero16);
}

// ---------------------------------------------------------------------------------------
// Launch-per-phase form: shared pieces
// ---------------------------------------------------------------------------------------
constexpr int kRecWords = 32;  // integer record of one block: value k of the reduction in word k (grid units of canon.hpp)

// "Last block folds and solves": every block publishes its record, then takes a ticket;
// the block that draws the last ticket reads all records and runs the scalar solve, so no
// separate launch (and no dependent-launch gap) is needed.  Hand-off protocol of
// cdna_hip_programming.md §6 G16 / §5 split-K recipe: every wave drains its stores, one lane
// issues an agent-scope release, then the relaxed ticket; the last arriver issues one agent-scope
// acquire before the block reads.  Correct for any placement of the blocks over the 8 XCDs.
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == gridDim.x - 1) ? 1 : 0;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// Fold of the integer records by one 256-thread block: thread (g = tid / 32, k = tid % 32) adds word k of records
// g, g + 8, ...; then thread k < 32 adds the 8 group sums.  Integer sums: exact, any order.  totals: LDS [32].
__device__ __forceinline__ void fold_records_i64(const long long* __restrict__ rec, int nblocks, long long* totals) {
  __shared__ long long s_g[8][32];
  const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
  // (eight loads in flight per thread: one dependent round trip per record made this fold 30 us of a 600-record level)
  long long s = 0;
  int b = g;
  for (; b + 56 < nblocks; b += 64) {
    long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = rec[(size_t)(b + 8 * u) * kRecWords + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < nblocks; b += 8) s += rec[(size_t)b * kRecWords + k];
  s_g[g][k] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    long long t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += s_g[q][threadIdx.x];
    totals[threadIdx.x] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ void fold_rows_i2(const int* __restrict__ partials, int stride, int nblocks, int* sums) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (wid < 2) {
    int s = 0;
    for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {  // 8 independent loads in flight
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 64 * u;
        v[u] = partials[(size_t)wid * stride + (b < nblocks ? b : 0)];  // clamp + select: keeps the batch in flight
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (b0 + 64 * u < nblocks) ? v[u] : 0;
    }
    s = wave_sum_to_lane63_i(s);
    if (lane == 63) sums[wid] = s;
  }
}

// block of kBlock threads: the per-thread biased accumulators -> one integer record (words 0 .. NV-1)
template <int N>
__device__ __forceinline__ void store_record(const double (&acc)[canon::Layout<N>::NV], const int* E, long long* rec) {
  __shared__ double s_red[kBlock / kWave][32];
  const double t = canon::block_fold<N, kBlock / kWave>(canon::wave_tree<N>(acc), s_red);
  if (threadIdx.x < 32) rec[threadIdx.x] = (int)threadIdx.x < canon::Layout<N>::NV ? canon::to_units(t, canon::value_exp<N>(E, threadIdx.x)) : 0ll;
}

// integer totals -> violation flag of a reduction with N columns (threads 0 .. 31 hold one value each)
template <int N>
__device__ __forceinline__ bool totals_violate(const long long* totals) {
  bool v = false;
#pragma unroll
  for (int c = 0; c <= N; ++c) v = v || totals[canon::Layout<N>::diag(c)] >= canon::kViolation;
  return v;
}
template <int N>
__device__ __forceinline__ void totals_to_float(const long long* totals, const int* E, float* sums) {
  for (int k = 0; k < canon::Layout<N>::NV; ++k) sums[k] = canon::from_units(totals[k], canon::value_exp<N>(E, k));
}

// ---------------------------------------------------------------------------------------
// SO3 pre-alignment (RGBDOdometry.cpp:297-385), launch-per-phase form
// ---------------------------------------------------------------------------------------
struct SolveCam {
  float fx, fy, cx, cy;
};

__device__ __forceinline__ So3Params so3_params_of(const TrackState* st, int cols, int rows) {
  So3Params p;
  const float* ib = st->imageBasis;
  const float* ki = st->kinv;
  const float* kr = st->krlr;
  p.imageBasis.r0 = mk3(ib[0], ib[1], ib[2]);
  p.imageBasis.r1 = mk3(ib[3], ib[4], ib[5]);
  p.imageBasis.r2 = mk3(ib[6], ib[7], ib[8]);
  p.kinv.r0 = mk3(ki[0], ki[1], ki[2]);
  p.kinv.r1 = mk3(ki[3], ki[4], ki[5]);
  p.kinv.r2 = mk3(ki[6], ki[7], ki[8]);
  p.krlr.r0 = mk3(kr[0], kr[1], kr[2]);
  p.krlr.r1 = mk3(kr[3], kr[4], kr[5]);
  p.krlr.r2 = mk3(kr[6], kr[7], kr[8]);
  p.cols = cols;
  p.rows = rows;
  return p;
}

// the block's threads sweep pixels [first, N) with stride `stride` and leave the per-thread accumulators
__device__ __forceinline__ void so3_accumulate(const So3Params& p, const unsigned char* lastImage, size_t last_pitch,
                                               const unsigned char* nextImage, size_t next_pitch, int first, int stride,
                                               double (&acc)[kSO3]) {
  const int N = p.cols * p.rows;
  for (int i = first; i < N; i += stride) {
    const int y = i / p.cols;
    const int x = i - y * p.cols;
    float row[4];
    const bool found = so3_row(p, lastImage, last_pitch, nextImage, next_pitch, x, y, row);
    canon::acc_add<3>(acc, row, found);
  }
}

__global__ __launch_bounds__(kBlock) void k_so3_pass(TrackState* st, const unsigned char* lastImage, size_t last_pitch,
                                                     const unsigned char* nextImage, size_t next_pitch, int cols, int rows, long long* records,
                                                     unsigned* ticket, SolveCam cam, int iter, int is_last, int first_gn_level, int exp_bias) {
  if (st->so3_done) return;
  __shared__ int s_E[4];
  __shared__ double s_bias[2][32];
  __shared__ long long s_tot[32];
  __shared__ float s_sums[kSO3];
  __shared__ int s_viol;
  const So3Params p = so3_params_of(st, cols, rows);
  if (threadIdx.x == 0) {
    int E[4];
    if (iter == 0) {
      canon::static_so3(cols * rows, E);
      for (int c = 0; c < 4; ++c) E[c] = canon::clamp_e(E[c] + exp_bias);
    } else {
      for (int c = 0; c < 4; ++c) E[c] = st->E_icp[c];  // (the SO3 stage borrows the slot: no Gauss-Newton reduction has run yet)
    }
    for (int c = 0; c < 4; ++c) s_E[c] = E[c];
  }
  __syncthreads();
  if (threadIdx.x < 32) canon::write_bias<3>(s_E, s_bias, threadIdx.x);
  __syncthreads();
  double acc[kSO3];
  canon::acc_init<3>(acc, s_bias);
  so3_accumulate(p, lastImage, last_pitch, nextImage, next_pitch, blockIdx.x * blockDim.x + threadIdx.x, blockDim.x * gridDim.x, acc);
  store_record<3>(acc, s_E, records + (size_t)blockIdx.x * kRecWords);
  if (!last_block_arrives(ticket)) return;
  // ---- last block: fold, check the grid, repeat alone on a coarser one if a diagonal total did not fit, solve ----
  fold_records_i64(records, gridDim.x, s_tot);
  int retries = 0;
  for (;;) {
    if (threadIdx.x == 0) s_viol = totals_violate<3>(s_tot) ? 1 : 0;
    __syncthreads();
    if (!s_viol) break;
    if (++retries > canon::kMaxRetries) break;
    if (threadIdx.x == 0) canon::retry_step<3>(s_E);
    __syncthreads();
    if (threadIdx.x < 32) canon::write_bias<3>(s_E, s_bias, threadIdx.x);
    __syncthreads();
    canon::acc_init<3>(acc, s_bias);
    so3_accumulate(p, lastImage, last_pitch, nextImage, next_pitch, threadIdx.x, blockDim.x, acc);
    __shared__ double s_red[kBlock / kWave][32];
    const double t = canon::block_fold<3, kBlock / kWave>(canon::wave_tree<3>(acc), s_red);
    __syncthreads();
    if (threadIdx.x < 32) s_tot[threadIdx.x] = (int)threadIdx.x < kSO3 ? canon::to_units(t, canon::value_exp<3>(s_E, threadIdx.x)) : 0ll;
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  totals_to_float<3>(s_tot, s_E, s_sums);
  if (retries > canon::kMaxRetries)
    for (int k = 0; k < kSO3 - 1; ++k) s_sums[k] = 0.f;
  st->canon_retries += retries;
  sc::so3_solve_core(st, s_sums, cam.fx, cam.fy, cam.cx, cam.cy, is_last, first_gn_level);
  int E[4];
  for (int c = 0; c < 4; ++c) E[c] = s_E[c];
  canon::next_exponents<3>(s_sums, E);
  for (int c = 0; c < 4; ++c) st->E_icp[c] = E[c];
}

// ---------------------------------------------------------------------------------------
// Gauss-Newton passes (RGBDOdometry.cpp:425-586), launch-per-phase form
// ---------------------------------------------------------------------------------------
using sc::SolveArgs;

// The fused loop's own correspondence record (the 16-byte DataTerm of the operator layer carries
// the live pixel's coordinates, which are the record's position, and a float difference of two
// bytes): half the bytes written by pass 1 and read back by pass 2.
struct Corr8 {
  short zero_x, zero_y;
  short diff;  // nextImage - lastImage, an integer in [-255, 255]
  short valid;
};

struct GnArgs {
  // ICP
  MapPtrs maps;
  float fx, fy, cx, cy, distThres, angleThres, dist2Le, sine2Le;  // (the last two: IcpParams)
  // RGB
  RgbResPtrs rgb;
  Corr8* corres;
  const float* cloud;
  unsigned cloud_pitch;
  float minScale, maxDepthDelta, sobelScale;
  int cols, rows, level;
  int rgbOnly, first_of_call, exp_bias;  // canonical sums: how the level's first exponents are derived (see level_exponents)
};

// Column exponents of a level's first reduction: the call's first Gauss-Newton level starts from the static table, every
// later one from the coarser level's last totals, four times the pixels (canon.hpp).  Uniform: every block derives the same.
__device__ __forceinline__ void level_exponents(const TrackState* st, const GnArgs& a, bool first_iter, int* E_icp, int* E_rgb) {
  if (first_iter && !st->have_E) {
    canon::static_icp(a.cols * a.rows, E_icp);
    canon::static_rgb(a.cols * a.rows, a.fx, a.rgbOnly, E_rgb);
    for (int c = 0; c < 7; ++c) {
      E_icp[c] = canon::clamp_e(E_icp[c] + a.exp_bias);
      E_rgb[c] = canon::clamp_e(E_rgb[c] + a.exp_bias);
    }
    return;
  }
  for (int c = 0; c < 7; ++c) {
    E_icp[c] = st->E_icp[c];
    E_rgb[c] = st->E_rgb[c];
  }
  if (first_iter) {
    canon::level_step<6>(E_icp);
    canon::level_step<6>(E_rgb);
  }
}

__device__ __forceinline__ IcpParams icp_params_of(const float* Rcurr, const float* tcurr, const float* Rprev_inv, const float* tprev, const GnArgs& a) {
  IcpParams ip;
  ip.Rcurr.r0 = mk3(Rcurr[0], Rcurr[1], Rcurr[2]);
  ip.Rcurr.r1 = mk3(Rcurr[3], Rcurr[4], Rcurr[5]);
  ip.Rcurr.r2 = mk3(Rcurr[6], Rcurr[7], Rcurr[8]);
  ip.tcurr = mk3(tcurr[0], tcurr[1], tcurr[2]);
  ip.Rprev_inv.r0 = mk3(Rprev_inv[0], Rprev_inv[1], Rprev_inv[2]);
  ip.Rprev_inv.r1 = mk3(Rprev_inv[3], Rprev_inv[4], Rprev_inv[5]);
  ip.Rprev_inv.r2 = mk3(Rprev_inv[6], Rprev_inv[7], Rprev_inv[8]);
  ip.tprev = mk3(tprev[0], tprev[1], tprev[2]);
  ip.fx = a.fx;
  ip.fy = a.fy;
  ip.cx = a.cx;
  ip.cy = a.cy;
  ip.distThres = a.distThres;
  ip.angleThres = a.angleThres;
  ip.dist2Le = a.dist2Le;
  ip.sine2Le = a.sine2Le;
  ip.cols = a.cols;
  ip.rows = a.rows;
  return ip;
}
__device__ __forceinline__ RgbResParams rgb_params_of(const float* krkinv, const float* kt, const GnArgs& a) {
  RgbResParams rp;
  rp.krkinv.r0 = mk3(krkinv[0], krkinv[1], krkinv[2]);
  rp.krkinv.r1 = mk3(krkinv[3], krkinv[4], krkinv[5]);
  rp.krkinv.r2 = mk3(krkinv[6], krkinv[7], krkinv[8]);
  rp.kt = mk3(kt[0], kt[1], kt[2]);
  rp.minScale = a.minScale;
  rp.maxDepthDelta = a.maxDepthDelta;
  rp.cols = a.cols;
  rp.rows = a.rows;
  return rp;
}

// Pixels are handed out in chunks of kBlock*kPix: thread t of the block owns pixels
// chunk*kBlock*kPix + p*kBlock + t (p < kPix), i.e. kPix coalesced runs, and keeps the loads of all
// kPix pixels (ICP and photometric) in flight together.  Two dependent round trips per chunk.
constexpr int kPix = 2;
inline int track_blocks_for(int n) {
  int b = (n + kBlock * kPix - 1) / (kBlock * kPix);
  return b < 1 ? 1 : (b > kMaxPartialBlocks ? kMaxPartialBlocks : b);
}

// ICP rows of the pixels [first chunk, N) in steps of `chunk_stride` chunks into the per-thread accumulators
__device__ __forceinline__ void icp_accumulate(const IcpParams& ip, const GnArgs& a, int first_chunk, int chunk_stride, double (&acc)[kSE3]) {
  const int N = a.cols * a.rows;
  for (int base = first_chunk * (kBlock * kPix); base < N; base += chunk_stride * (kBlock * kPix)) {
    int idx[kPix];
    IcpOwn io[kPix];
#pragma unroll
    for (int p = 0; p < kPix; ++p) {
      idx[p] = base + p * kBlock + (int)threadIdx.x;
      const int ic = idx[p] < N ? idx[p] : 0;  // lanes past the end shadow pixel 0 and are masked below
      const int py = ic / a.cols;
      io[p] = icp_load_own(a.maps, ic - py * a.cols, py, a.rows);
    }
    IcpProj ir[kPix];
    IcpModel im[kPix];
#pragma unroll
    for (int p = 0; p < kPix; ++p) {
      ir[p] = icp_project<kTrackerFma>(ip, io[p]);
      im[p] = icp_load_model(a.maps, ir[p], a.rows);
    }
#pragma unroll
    for (int p = 0; p < kPix; ++p) {
      float row[7];
      bool found = icp_finish<kTrackerFma>(ip, io[p], ir[p], im[p], row);
      if (idx[p] >= N) {
        found = false;
#pragma unroll
        for (int k = 0; k < 7; ++k) row[k] = 0.f;
      }
      canon::acc_add<6>(acc, row, found);
    }
  }
}

// photometric rows from the correspondence image, same sweep
__device__ __forceinline__ void rgb_accumulate(const RgbStepParams& p_, const GnArgs& a, int first_chunk, int chunk_stride, double (&acc)[kSE3]) {
  const int N = a.cols * a.rows;
  for (int base = first_chunk * (kBlock * kPix); base < N; base += chunk_stride * (kBlock * kPix)) {
    dms_dataterm c[kPix];
    RgbRowIn in[kPix];
#pragma unroll
    for (int p = 0; p < kPix; ++p) {
      const int i = base + p * kBlock + (int)threadIdx.x;
      const int ic = i < N ? i : 0;
      const Corr8 c8 = a.corres[ic];
      const int y = ic / a.cols;
      c[p].zero_x = c8.zero_x;
      c[p].zero_y = c8.zero_y;
      c[p].one_x = (short)(ic - y * a.cols);
      c[p].one_y = (short)y;
      c[p].diff = (float)c8.diff;
      c[p].valid = (i < N) ? (int)c8.valid : 0;
    }
#pragma unroll
    for (int p = 0; p < kPix; ++p) in[p] = rgb_row_load(c[p], a.cloud, a.cloud_pitch, a.rgb.dIdx, a.rgb.dIdy, a.rgb.dI_pitch);
#pragma unroll
    for (int p = 0; p < kPix; ++p) {
      float row[7];
      rgb_row_finish<kTrackerFma>(p_, c[p], in[p], row);
      canon::acc_add<6>(acc, row, c[p].valid != 0);
    }
  }
}

template <bool ICP, bool RGB>
__global__ __launch_bounds__(kBlock) void k_gn_pass1(TrackState* st, GnArgs a, long long* part_icp, int* __restrict__ part_cnt, int stride,
                                                     int first_iter) {
  if (st->level_done[a.level]) return;
  __shared__ int s_E[2][8];
  __shared__ double s_bias[2][32];
  const int N = a.cols * a.rows;
  if (threadIdx.x == 0) level_exponents(st, a, first_iter != 0, s_E[0], s_E[1]);
  __syncthreads();
  if (ICP && threadIdx.x < 32) canon::write_bias<6>(s_E[0], s_bias, threadIdx.x);
  __syncthreads();
  if (ICP) {
    const IcpParams ip = icp_params_of(st->Rcurr, st->tcurr, st->Rprev_inv, st->tprev, a);
    double acc[kSE3];
    canon::acc_init<6>(acc, s_bias);
    icp_accumulate(ip, a, blockIdx.x, gridDim.x, acc);
    store_record<6>(acc, s_E[0], part_icp + (size_t)blockIdx.x * kRecWords);
  }
  if (RGB) {
    const RgbResParams rp = rgb_params_of(st->krkinv, st->kt, a);
    int cnt = 0, sig = 0;
    for (int base = blockIdx.x * (kBlock * kPix); base < N; base += gridDim.x * (kBlock * kPix)) {
      int idx[kPix], px[kPix], py[kPix];
      RgbOwn ro[kPix];
#pragma unroll
      for (int p = 0; p < kPix; ++p) {
        idx[p] = base + p * kBlock + (int)threadIdx.x;
        const int ic = idx[p] < N ? idx[p] : 0;
        py[p] = ic / a.cols;
        px[p] = ic - py[p] * a.cols;
        ro[p] = rgb_load_own_gated(a.rgb, px[p], py[p]);
      }
      RgbProj rr[kPix];
      RgbModel rm[kPix];
#pragma unroll
      for (int p = 0; p < kPix; ++p) {
        rr[p] = rgb_project<kTrackerFma>(rp, ro[p], px[p], py[p]);
        rm[p] = rgb_load_model(a.rgb, rr[p]);
      }
#pragma unroll
      for (int p = 0; p < kPix; ++p) {
        dms_dataterm c;
        int d2;
        const bool ok = rgb_finish(rp, ro[p], rr[p], rm[p], px[p], py[p], c, d2);
        if (idx[p] < N) {
          if (ok) {
            cnt += 1;
            sig += d2;
          }
          Corr8 c8;
          c8.zero_x = c.zero_x;
          c8.zero_y = c.zero_y;
          c8.diff = (short)f2i_rz(c.diff);
          c8.valid = (short)c.valid;
          a.corres[idx[p]] = c8;
        }
      }
    }
    __shared__ int lds[kBlock / kWave][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    cnt = wave_sum_to_lane63_i(cnt);
    sig = wave_sum_to_lane63_i(sig);
    if (lane == 63) {
      lds[wid][0] = cnt;
      lds[wid][1] = sig;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      int s = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += lds[w][threadIdx.x];
      part_cnt[(size_t)threadIdx.x * stride + blockIdx.x] = s;
    }
  }
}

// every block folds the (≤1024) integer partials itself: integer sums are order-free, so
// all blocks agree bit-for-bit and no extra launch is needed to publish σ.
__device__ __forceinline__ void fold_count_pair(const int* __restrict__ part_cnt, int nb_cnt, int stride, int* s_tot /* LDS [2] */) {
  __shared__ int s_cnt[kBlock / kWave][2];
  int c = 0, g = 0;
  for (int b = threadIdx.x; b < nb_cnt; b += blockDim.x) {
    c += part_cnt[b];
    g += part_cnt[(size_t)stride + b];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  c = wave_sum_to_lane63_i(c);
  g = wave_sum_to_lane63_i(g);
  if (lane == 63) {
    s_cnt[wid][0] = c;
    s_cnt[wid][1] = g;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    int s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_cnt[w][threadIdx.x];
    s_tot[threadIdx.x] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_gn_pass2(TrackState* st, GnArgs a, const int* __restrict__ part_cnt, int nb_cnt, int stride,
                                                     int first_iter, long long* part_rgb) {
  if (st->level_done[a.level]) return;
  __shared__ int s_tot[2];
  __shared__ int s_E[2][8];
  __shared__ double s_bias[2][32];
  fold_count_pair(part_cnt, nb_cnt, stride, s_tot);
  const int rgbSize = s_tot[0], sigma = s_tot[1];
  const float lastErr = first_iter ? 3.402823466e+38F : st->lastRGBError;
  if (a.rgbOnly && sc::rgbonly_break(sigma, rgbSize, lastErr)) return;  // host `break`; k_gn_solve records it
  if (threadIdx.x == 0) level_exponents(st, a, first_iter != 0, s_E[0], s_E[1]);
  __syncthreads();
  if (threadIdx.x < 32) canon::write_bias<6>(s_E[1], s_bias, threadIdx.x);
  __syncthreads();
  RgbStepParams p_;
  p_.sigma = a.rgbOnly ? -1.f : sc::sigma_val(sigma, rgbSize);
  p_.fx = a.fx;
  p_.fy = a.fy;
  p_.sobelScale = a.sobelScale;
  double acc[kSE3];
  canon::acc_init<6>(acc, s_bias);
  rgb_accumulate(p_, a, blockIdx.x, gridDim.x, acc);
  store_record<6>(acc, s_E[1], part_rgb + (size_t)blockIdx.x * kRecWords);
}

// One block: fold both record sets, check their grids (a reduction whose diagonal totals did not fit is repeated by this
// block alone on a coarser grid — rare: the static guess of a call's first iteration, or a scene change between
// iterations), then the scalar section.
template <bool ICP, bool RGB>
__global__ __launch_bounds__(kBlock) void k_gn_solve(TrackState* st, GnArgs a, const long long* part_icp, const long long* part_rgb,
                                                      const int* part_cnt, int stride, int nblocks, SolveArgs q) {
  const int level = q.level, first_iter = q.first_iter;
  __shared__ long long s_ti[32], s_tr[32];
  __shared__ float s_icp[32], s_rgb[32];
  __shared__ int s_cnt[2];
  __shared__ int s_E[2][8];
  __shared__ double s_bias[2][32];
  __shared__ double s_red[kBlock / kWave][32];
  __shared__ int s_viol;
  if (st->level_done[level]) return;
  const float lastErr = first_iter ? 3.402823466e+38F : st->lastRGBError;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) level_exponents(st, a, first_iter != 0, s_E[0], s_E[1]);
  __syncthreads();
  if (RGB) fold_rows_i2(part_cnt, stride, nblocks, s_cnt);
  __syncthreads();
  const int rgbSize = s_cnt[0], sigma = s_cnt[1];
  const bool brk = q.rgbOnly && sc::rgbonly_break(sigma, rgbSize, lastErr);
  if (brk) {
    // host `break` (RGBDOdometry.cpp:466-469): the level ends here; the next level that runs
    // needs its own K in the projection parameters
    if (threadIdx.x == 0) {
      st->level_done[level] = 1;
      double Rt[16];
      for (int i = 0; i < 16; ++i) Rt[i] = st->resultRt[i];
      sc::gn_params(Rt, sc::kpre_of(q.fx, q.fy, q.cx, q.cy, q.level_below), st->krkinv, st->kt);
    }
    return;
  }
  if (ICP) fold_records_i64(part_icp, nblocks, s_ti);
  if (RGB) fold_records_i64(part_rgb, nblocks, s_tr);
  int retries = 0;
  bool gave_up_icp = false, gave_up_rgb = false;
  if (ICP) {
    int r = 0;
    for (;;) {
      if (threadIdx.x == 0) s_viol = totals_violate<6>(s_ti) ? 1 : 0;
      __syncthreads();
      if (!s_viol) break;
      if (++r > canon::kMaxRetries) {
        gave_up_icp = true;
        break;
      }
      if (threadIdx.x == 0) canon::retry_step<6>(s_E[0]);
      __syncthreads();
      if (threadIdx.x < 32) canon::write_bias<6>(s_E[0], s_bias, threadIdx.x);
      __syncthreads();
      const IcpParams ip = icp_params_of(st->Rcurr, st->tcurr, st->Rprev_inv, st->tprev, a);
      double acc[kSE3];
      canon::acc_init<6>(acc, s_bias);
      icp_accumulate(ip, a, 0, 1, acc);
      const double t = canon::block_fold<6, kBlock / kWave>(canon::wave_tree<6>(acc), s_red);
      __syncthreads();
      if (threadIdx.x < 32) s_ti[threadIdx.x] = (int)threadIdx.x < kSE3 ? canon::to_units(t, canon::value_exp<6>(s_E[0], threadIdx.x)) : 0ll;
      __syncthreads();
    }
    retries += r > canon::kMaxRetries ? canon::kMaxRetries : r;
  }
  if (RGB) {
    int r = 0;
    for (;;) {
      if (threadIdx.x == 0) s_viol = totals_violate<6>(s_tr) ? 1 : 0;
      __syncthreads();
      if (!s_viol) break;
      if (++r > canon::kMaxRetries) {
        gave_up_rgb = true;
        break;
      }
      if (threadIdx.x == 0) canon::retry_step<6>(s_E[1]);
      __syncthreads();
      if (threadIdx.x < 32) canon::write_bias<6>(s_E[1], s_bias, threadIdx.x);
      __syncthreads();
      RgbStepParams p_;
      p_.sigma = q.rgbOnly ? -1.f : sc::sigma_val(sigma, rgbSize);
      p_.fx = a.fx;
      p_.fy = a.fy;
      p_.sobelScale = a.sobelScale;
      double acc[kSE3];
      canon::acc_init<6>(acc, s_bias);
      rgb_accumulate(p_, a, 0, 1, acc);
      const double t = canon::block_fold<6, kBlock / kWave>(canon::wave_tree<6>(acc), s_red);
      __syncthreads();
      if (threadIdx.x < 32) s_tr[threadIdx.x] = (int)threadIdx.x < kSE3 ? canon::to_units(t, canon::value_exp<6>(s_E[1], threadIdx.x)) : 0ll;
      __syncthreads();
    }
    retries += r > canon::kMaxRetries ? canon::kMaxRetries : r;
  }
  if (threadIdx.x != 0) return;
  if (ICP) {
    totals_to_float<6>(s_ti, s_E[0], s_icp);
    if (gave_up_icp)
      for (int k = 0; k < 28; ++k) s_icp[k] = 0.f;
  }
  if (RGB) {
    totals_to_float<6>(s_tr, s_E[1], s_rgb);
    if (gave_up_rgb)
      for (int k = 0; k < 28; ++k) s_rgb[k] = 0.f;
  }

  sc::GnLocal L;
#pragma unroll
  for (int i = 0; i < 16; ++i) L.resultRt[i] = st->resultRt[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) L.Rprev[i] = st->Rprev[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) L.tprev[i] = st->tprev[i];
  L.iters_run = st->iters_run[level];
  sc::gn_step_core(L, s_icp, s_rgb, rgbSize, sigma, q, sc::kpre_of(q.fx, q.fy, q.cx, q.cy, q.next_level));

  st->iters_run[level] = L.iters_run;
  st->lastRGBError = L.lastRGBError;
  st->lastRGBCount = L.lastRGBCount;
  st->lastICPError = L.lastICPError;
  st->lastICPCount = L.lastICPCount;
#pragma unroll
  for (int i = 0; i < 36; ++i) st->lastA[i] = L.lastA[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) st->lastb[i] = L.lastb[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) st->resultRt[i] = L.resultRt[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) st->Rcurr[i] = L.Rcurr[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) st->tcurr[i] = L.tcurr[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) st->krkinv[i] = L.krkinv[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) st->kt[i] = L.kt[i];
  // exponents of the next reduction from this one's totals
  for (int c = 0; c < 7; ++c) {
    st->E_icp[c] = s_E[0][c];
    st->E_rgb[c] = s_E[1][c];
  }
  if (ICP) canon::next_exponents<6>(s_icp, st->E_icp);
  if (RGB) canon::next_exponents<6>(s_rgb, st->E_rgb);
  st->have_E = 1;
  st->canon_retries += retries;
}

// ---------------------------------------------------------------------------------------
// Persistent Gauss-Newton level: ALL iterations of one pyramid level in one launch
// ---------------------------------------------------------------------------------------
// A GN iteration is two grid-wide reductions deep (sigma needs the global correspondence count;
// the 6x6 system needs every pixel), so the launch path pays three kernel boundaries, three
// dispatch ramps and three rounds of dependent state loads per iteration.  Here one grid of
// <= 256 blocks x 512 threads (at most one block per CU, 1-4 pixels per thread) stays resident for the
// whole level:
//   * everything pose independent is loaded ONCE per level into registers (own vertex / normal,
//     photometric gate, own depth / intensity / gradient);
//   * the correspondences never leave registers (no DataTerm image, no point-cloud image);
//   * both grid-wide sums are integer all-reduces in memory-side atomics: lane k of wave 0 holds the block's exact
//     total of value k in grid units (canon.hpp) and adds it with ONE non-returning 64-bit agent-scope atomic to word
//     [block % 8][k]; the same lane then polls its own 8 shard words until their arrival fields show every block — the
//     total is then in its registers.  No record, no separate barrier, no gather; integer adds are order free, so the
//     totals are bit-identical in every block, from run to run and on any grid.  The words of one shard are contiguous
//     (one coalesced atomic instruction per shard region); with the 8 shards of a value in one cache line the atomics
//     serialise per line (7.1 us instead of 1.9, scripts/bench_allreduce.hip);
//   * every block then runs the 6x6 solve itself on identical totals: all blocks hold bit-identical poses, nothing is
//     broadcast; host `break`s are uniform decisions every block takes from the same sums;
//   * a reduction whose diagonal totals do not fit the grid its exponents promised (canon.hpp) is repeated by the whole
//     grid — a uniform decision again — on a word set from the launch's pool.
// Hand-off rules follow cdna_hip_programming.md G16 (relaxed polling, agent-scope words, bounded spins, words zeroed
// by an earlier kernel on the stream); scripts/bench_gridbarrier.hip is the protocol's stand-alone visibility test.
constexpr int kPB = 512;                 // 8 waves: 256 VGPRs per thread, half the reduction tail of 16 waves
constexpr int kPWaves = kPB / kWave;
constexpr int kMaxPersistBlocks = 256;   // one block per CU
constexpr unsigned kSpinLimit = 1u << 22;

struct LevelArgs {
  int n_iter, level, level_below;
  int rgbOnly;
  float icpWeight;
  float fx, fy, cx, cy;  // full-resolution intrinsics
  unsigned long long* ar;    // kArSetsPerKernel word sets of kArWords, zero on entry: one per iteration, then the retry pool
  int first_delay;           // see ar_wait (units of 64 cycles)
  unsigned* zero16;          // the frame step's 16 dense counters, re-armed by the call's first Gauss-Newton launch when no earlier kernel of the call could (k_so3_model reads them)
  long long* prof;           // optional [3][16] per-level phase clocks of block 0 (null = off)
  // last level of the call: block 0 also does what k_track_finalize does (jump gate, result block,
  // pose write-back, frame bookkeeping) instead of a one-lane launch of its own
  int early_exit;  // host side: selects the EXIT instantiation of k_gn_level
  int finalize, fin_rgb;
  float* pose16_out;
  FrameState* frame;
  float weightMultiplier;
};

// ---- grid-wide integer all-reduce in memory-side atomics ------------------------------------------------------
//   sum word  = arrivals [63:58] | sum of (S + 2^52) [57:0], |S| < 2^52 grid units, <= 32 blocks per shard (canon::pack_word)
//   pair word = arrivals [63:58] | sum of v [57:0], v = the block's correspondence count or sum of squared differences
constexpr int kArShards = 8;
constexpr int kArStride = 64;                       // words per shard: slots 0..28 ICP | 32..60 photometric
constexpr int kArPairBase = kArShards * kArStride;  // then the count / sum-of-squares pair: one 128-byte line per shard (slots 0, 1),
constexpr int kArPairStride = 16;                   // apart from the lines the 58 sums arrive on (its pollers would slow those atomics)
constexpr int kArWords = kArPairBase + kArShards * kArPairStride;  // 5 KB per reduction
constexpr int kArSlotCnt = 0, kArSlotSig = 1;
constexpr int kArPool = 6;                          // extra word sets per resident launch for repeated reductions
constexpr int kArRing = 10;                          // word sets a stage cycles through, one per iteration (re-armed in flight when a level runs more iterations)
constexpr int kArSetsPerKernel = kArRing + kArPool;
constexpr int kArSets = 4 * kArSetsPerKernel;       // SO3 + three levels

__device__ __forceinline__ unsigned long long pair_pack(unsigned long long v) { return (1ull << 58) + v; }

// wave 0, all 64 lanes: every lane with `mine` polls the 8 shard words of `slot` until they show nb arrivals.
// fld = sum of the 8 fields.  Bounded: a timeout sets *timeout and returns false.
// WHEN to poll matters more than how: a read that reaches the words before the last block's atomic has landed costs a whole
// further round trip (~0.8 us), and reads of words that are still receiving atomics slow those down.  The callers therefore
// wait `first_delay` x 64 cycles after their own arrival before the first read of the totals (nobody's sum can be complete
// earlier than one atomic flight after the last arrival).  Measured at 640x480 (level 0 / 1 / 2 launch, us), round 3:
//   probe word, then a sweep of all 58 words, no delay (round 2's form)                        137 / 53 / 41.6
//   ICP words (added a phase earlier, complete: one round trip), then the photometric words    131 / 48 / 40.5
//   the same after a delay of 8 / 16 / 30 units                                                 121.5 / 118.3 / 120.6 (level 0)
//   ICP wave sums kept in registers and folded over the waves together with the photometric
//   ones (one LDS fold and one atomic instruction for both; the pair is polled ~0.5 us earlier),
//   all 58 words polled at once after 24 units (8 on grids under 96 blocks)                     111 / 44 / 38.7
// Tried without gain: pauses between polls (1 - 12 units); the ICP words read before pass 2 and looked at after it (140: the
// early reads delay wave 0's share of the pass); every block adding to 2 or 4 copies of the words and polling one (120 ->
// 121 / 122: the readers' fan-in per line is not what limits); a delay before the pair's poll (+0 - 3 us from 4 units on);
// a block's chunks spread over the image instead of adjacent (no difference: the arrival spread is not a load imbalance);
// 16 shards instead of 8 (no difference: 13 instead of 25 same-address atomics per word against twice the words to read).
// Also without gain: the ICP rows computed after the count pair's arrival instead of before it (pass 1 -0.6 us per iteration,
// the two waits +0.9: the pair's flight was already covered by the ICP wave sums).
__device__ __forceinline__ void poll_pause(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);  // 64 cycles each
}
template <int STRIDE>
__device__ __forceinline__ bool ar_wait(const unsigned long long* w, int slot, bool mine, int nb, unsigned long long& fld_out, int* timeout) {
  unsigned spins = 0;
  for (;;) {
    unsigned long long arr = 0, fld = 0;
    if (mine) {
      unsigned long long q[kArShards];
#pragma unroll
      for (int s = 0; s < kArShards; ++s) q[s] = __hip_atomic_load(w + s * STRIDE + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int s = 0; s < kArShards; ++s) {
        arr += q[s] >> 58;
        fld += q[s] & ((1ull << 58) - 1ull);
      }
    }
    const bool done = !mine || arr == (unsigned long long)nb;
    if (__builtin_amdgcn_ballot_w64(done) == ~0ull) {
      fld_out = fld;
      return true;
    }
    ++spins;
    if (spins > kSpinLimit || ((spins & 1023u) == 0u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // never hang the device
      fld_out = 0;
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// completed sum words -> signed total in grid units
__device__ __forceinline__ long long ar_total(unsigned long long fld, int nb) { return (long long)fld - (long long)nb * (1ll << 52); }

// bit k set: value k of the 29-value layout is a diagonal entry (the check of canon.hpp applies to it)
__device__ __forceinline__ constexpr unsigned se3_diag_mask() {
  unsigned m = 0;
  for (int c = 0; c <= 6; ++c) m |= 1u << canon::Layout<6>::diag(c);
  return m;
}
__device__ __forceinline__ constexpr unsigned so3_diag_mask() {
  unsigned m = 0;
  for (int c = 0; c <= 3; ++c) m |= 1u << canon::Layout<3>::diag(c);
  return m;
}

// EXIT: leave the level after an iteration without any correspondence (below).  A template parameter because the
// mere presence of that exit costs the frame-to-model tracker ~1 % (codegen of the resident loop); the frame step
// instantiates it for the model-to-model pass only.
// The level as a stage: `st` is the call's state block in HBM (the sticky timeout flag lives there), `sv` the state the stage reads
// at its start and leaves at its end - the same block for the stand-alone kernel (block 0 writes it back for the next launch), the
// block's own LDS copy when several stages run in one launch (k_track_coarse: FUSED; every block holds the same bits, so every
// block keeps its copy up to date and nothing crosses blocks between two stages).  nb = blocks taking part in the all-reduces.
template <bool ICP, bool RGB, int P, bool EXIT, bool FUSED>
__device__ __forceinline__ void gn_level_body(TrackState* st, TrackState* sv, const GnArgs& a, const LevelArgs& L, const int nb) {
  constexpr bool kFma = kTrackerFma;
  __shared__ sc::GnLocal s;
  __shared__ int s_redi[kPWaves][2];
  __shared__ double s_red[kPWaves][32];
  __shared__ double s_red2[2][kPWaves][32];  // [ICP | photometric][wave][value]: the two reductions folded at once
  __shared__ double s_bias[2][2][32];  // [ICP | photometric][lanes 0-31 | 32-63][value]
  __shared__ int s_E[2][8];            // column exponents of the two reductions
  __shared__ float s_sums[64];
  __shared__ double s_comb[28];  // the combined 6x6 system (27 unique entries), written by the lanes that hold the totals
  __shared__ float s_held[48];  // finalize step: new pose | previous frame's pose | inverse (frame_state.hpp)
  __shared__ int s_none;
  __shared__ int s_done;
  __shared__ int s_cs[2];
  __shared__ int s_viol;
  __shared__ int s_retries;
  __shared__ sc::KPre s_k[2];
  const int tid = threadIdx.x;
  int eb_icp = 0, eb_rgb = 0;  // wave 0: bound exponents of the values this lane adds (lanes 0-28) / polls (lane k: ICP k, lane 32 + k: photometric k)
  // optional phase clock (block 0, thread 0): wall_clock64 ticks (10 ns) summed per phase into L.prof
  // (accumulated in LDS and flushed once at the end: a global read-modify-write per phase would
  // stall wave 0 for a memory round trip each time and distort what it measures)
  __shared__ long long s_prof[16];
  if (tid < 16) s_prof[tid] = 0;
  long long pc = L.prof ? wall_clock64() : 0;
  auto phase = [&](int i) {
    if (L.prof && blockIdx.x == 0 && tid == 0) {
      const long long now = wall_clock64();
      s_prof[i] += now - pc;
      pc = now;
    }
  };

  if (L.zero16 && blockIdx.x == 0 && tid >= 96 && tid < 112) L.zero16[(tid - 96) * 16] = 0u;
  if (L.finalize && L.frame && blockIdx.x == 0 && tid >= 64 && tid < 80) s_held[16 + tid - 64] = L.frame->lastPose[tid - 64];  // for the finalize step
  // (the pose block's bottom row is not the tracker's to write, see the finalize step: it is carried through)
  if (L.finalize && L.frame && blockIdx.x == 0 && tid >= 80 && tid < 84) s_held[12 + tid - 80] = L.frame->cur.pose[12 + tid - 80];
  if (tid == 0) {
    s_none = 0;
    s_retries = 0;
    s_done = sv->level_done[L.level];
    for (int i = 0; i < 16; ++i) s.resultRt[i] = sv->resultRt[i];
    for (int i = 0; i < 9; ++i) {
      s.Rprev[i] = sv->Rprev[i];
      s.Rprev_inv[i] = sv->Rprev_inv[i];
      s.Rcurr[i] = sv->Rcurr[i];
      s.krkinv[i] = sv->krkinv[i];
    }
    for (int i = 0; i < 3; ++i) {
      s.tprev[i] = sv->tprev[i];
      s.tcurr[i] = sv->tcurr[i];
      s.kt[i] = sv->kt[i];
    }
    s.lastRGBError = sv->lastRGBError;
    s.lastRGBCount = sv->lastRGBCount;
    s.lastICPError = sv->lastICPError;
    s.lastICPCount = sv->lastICPCount;
    s.iters_run = sv->iters_run[L.level];
    for (int i = 0; i < 36; ++i) s.lastA[i] = sv->lastA[i];
    for (int i = 0; i < 6; ++i) s.lastb[i] = sv->lastb[i];
    level_exponents(sv, a, true, s_E[0], s_E[1]);
    // camera matrices of this level and of the next one that runs, for the scalar section (thread 0 uses them)
    // (kept in LDS: 24 more live registers per lane would spill the 256-register pixel loop)
    s_k[0] = sc::kpre_of(L.fx, L.fy, L.cx, L.cy, L.level);
    s_k[1] = sc::kpre_of(L.fx, L.fy, L.cx, L.cy, L.level_below);
  }

  // ---- pose-independent per-pixel data, once per level ----
  const int N = a.cols * a.rows;
  int idx[P], px[P], py[P];
  IcpOwn io[P];
  RgbOwn ro[P];
  short gx[P], gy[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    idx[p] = (blockIdx.x * P + p) * kPB + tid;
    // threads past the end shadow a pixel of the image and are masked.  (Not pixel 0 for all of them: in k_track_coarse whole blocks lie
    // past a small level's image, and tens of thousands of lanes loading one address queue up on its cache line - measured: +5 us
    // at the start of level 2 and +0.7 us per pass.)
    const int ic = idx[p] < N ? idx[p] : idx[p] % N;
    py[p] = ic / a.cols;
    px[p] = ic - py[p] * a.cols;
    if (ICP) io[p] = icp_load_own(a.maps, px[p], py[p], a.rows);
    if (RGB) {
      ro[p] = rgb_load_own_gated(a.rgb, px[p], py[p]);
      gx[p] = trow(a.rgb.dIdx, a.rgb.dI_pitch, py[p])[px[p]];
      gy[p] = trow(a.rgb.dIdy, a.rgb.dI_pitch, py[p])[px[p]];
    }
  }
  const float invFx = 1.0f / a.fx, invFy = 1.0f / a.fy;  // as projectToPointCloud passes them
  __syncthreads();
  if (s_done) return;  // level already ended (uniform: every block read the same flag)
  if (tid < 64) {
    canon::write_bias<6>(s_E[tid >> 5], s_bias[tid >> 5], tid & 31);
    eb_icp = canon::value_exp<6>(s_E[0], tid & 31);
    eb_rgb = canon::value_exp<6>(s_E[1], tid & 31);
  }
  __syncthreads();

  bool ended = false;
  int pool_used = 0;  // word sets of the retry pool consumed so far (uniform)
  bool on_pool = false;
  phase(0);
  for (int it = 0; it < L.n_iter;) {
    // ---- pass 1: correspondences + ICP rows ----
    IcpParams ip;
    RgbResParams rp;
    if (ICP) ip = icp_params_of(s.Rcurr, s.tcurr, s.Rprev_inv, s.tprev, a);
    if (RGB) rp = rgb_params_of(s.krkinv, s.kt, a);
    const float lastErr = (it == 0) ? 3.402823466e+38F : s.lastRGBError;

    IcpProj ir[P];
    RgbProj rr[P];
    IcpModel im[P];
    RgbModel rm[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ICP) {
        ir[p] = icp_project<kFma>(ip, io[p]);
        im[p] = icp_load_model(a.maps, ir[p], a.rows);
      }
      if (RGB) {
        rr[p] = rgb_project<kFma>(rp, ro[p], px[p], py[p]);
        rm[p] = rgb_load_model(a.rgb, rr[p]);
      }
    }
    float rows[P][7];
    bool found[P];
    int cnt = 0, sig = 0;
    dms_dataterm c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const bool live = idx[p] < N;
      if (RGB) {
        int d2;
        const bool ok = rgb_finish(rp, ro[p], rr[p], rm[p], px[p], py[p], c[p], d2) && live;
        if (!live) c[p].valid = 0;
        if (ok) {
          cnt += 1;
          sig += d2;
        }
      }
      if (ICP) {
        found[p] = icp_finish<kFma>(ip, io[p], ir[p], im[p], rows[p]);
        if (!live) {
          found[p] = false;
#pragma unroll
          for (int k = 0; k < 7; ++k) rows[p][k] = 0.f;
        }
      }
    }
    phase(1);
    // per-block stamps of the level's middle iteration (profiling: arrival skew and completion latency of the two
    // grid-wide reductions): pass 1 done | count pair complete | pass 2 + block sum done | totals complete
    const bool stamp = L.prof && L.level == 0 && it == L.n_iter / 2 && tid == 0;
    if (stamp) L.prof[48 + blockIdx.x * 8 + 0] = wall_clock64();
    // word set of this iteration: one of a RING of kArRing sets (a level may run more iterations than the ring has sets - the 50 of an
    // inter-map call: the set of iteration it - 1 is re-armed during iteration it, below), or the next set of the retry pool
    const size_t set = (size_t)(on_pool ? kArRing + pool_used - 1 : it % kArRing) * kArWords;
    unsigned long long* arw = L.ar + set + (size_t)(blockIdx.x & (kArShards - 1)) * kArStride;
    const unsigned long long* arp = L.ar + set;
    unsigned long long* arq = L.ar + set + kArPairBase + (size_t)(blockIdx.x & (kArShards - 1)) * kArPairStride;
    int rgbSize = 0, sigma = 0;
    if (RGB) {
      // ---- count / sum-of-squares pair, arrival ----
      const int lane = tid & 63, wid = tid >> 6;
      cnt = wave_sum_to_lane63_i(cnt);
      sig = wave_sum_to_lane63_i(sig);
      if (lane == 63) {
        s_redi[wid][0] = cnt;
        s_redi[wid][1] = sig;
      }
      __syncthreads();
      if (tid < 64) {  // lanes 0-7: the 8 wave counts, lanes 8-15: the 8 wave sums (block sums fit an int)
        const int t = row8_sum_i(tid < 16 ? s_redi[tid & 7][tid >> 3] : 0);
        const unsigned long long cb = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(t, 0);
        const unsigned long long sb = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(t, 8);
        if (tid < 2) __hip_atomic_fetch_add(arq + tid, pair_pack(tid == kArSlotCnt ? cb : sb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    phase(2);
    double r_icp = 0.0;  // (both terms on) this wave's ICP sums, one value per lane pair: folded over the waves together with the photometric ones
    if (ICP) {  // the ICP sums overlap the other blocks' arrivals
      if (RGB) {
        r_icp = canon::wave_sum<6, P>(rows, found, s_bias[0]);
      } else {
        const double p_icp = canon::block_sum<6, P, kPWaves>(rows, found, s_bias[0], s_red);
        if (tid < kSE3) __hip_atomic_fetch_add(arw + tid, canon::pack_word(canon::to_units(p_icp, eb_icp)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // what the photometric rows need that does not depend on sigma — the model pixel's cloud point, exactly as k_projectPoints
    // builds it from lastDepth, and 1 / z — is taken while the count is in flight (the waves other than wave 0 only wait here)
    f3 cpt[P];
    float cinvz[P];
    if (RGB) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float z = rm[p].d0;
        cpt[p] = mk3((((float)c[p].zero_x - a.cx) * z) * invFx, (((float)c[p].zero_y - a.cy) * z) * invFy, z);
        cinvz[p] = rgb_row_invz(cpt[p]);
      }
    }
    if (RGB) {
      if (tid < 64) {
        unsigned long long fld;
        if (stamp) L.prof[48 + blockIdx.x * 8 + 4] = wall_clock64();  // poll start
        ar_wait<kArPairStride>(arp + kArPairBase, tid, tid < 2, nb, fld, &st->sync_timeout);
        const int c0 = __builtin_amdgcn_readlane((int)fld, kArSlotCnt), s0 = __builtin_amdgcn_readlane((int)fld, kArSlotSig);
        if (tid == 0) {
          s_cs[0] = c0;
          s_cs[1] = s0;  // (the low 32 bits: an int sum, as in the reference)
        }
      }
      __syncthreads();
      rgbSize = s_cs[0];
      sigma = s_cs[1];
      phase(3);
      if (stamp) L.prof[48 + blockIdx.x * 8 + 1] = wall_clock64();
      if (L.rgbOnly && sc::rgbonly_break(sigma, rgbSize, lastErr)) {
        // host `break` (RGBDOdometry.cpp:466-469): the level ends; the next level needs its own K
        if (tid == 0) {
          double Rt[16];
          for (int i = 0; i < 16; ++i) Rt[i] = s.resultRt[i];
          sc::gn_params(Rt, s_k[1], s.krkinv, s.kt);
        }
        ended = true;
        break;
      }
      // ---- pass 2: photometric rows from the registers ----
      RgbStepParams p_;
      p_.sigma = L.rgbOnly ? -1.f : sc::sigma_val(sigma, rgbSize);
      p_.fx = a.fx;
      p_.fy = a.fy;
      p_.sobelScale = a.sobelScale;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        RgbRowIn in;
        in.pt = cpt[p];
        in.gx = gx[p];
        in.gy = gy[p];
        rgb_row_finish<kFma>(p_, c[p], in, rows[p], cinvz[p], true);
        found[p] = c[p].valid != 0;
      }
      if (ICP) {
        // one fold for both reductions: lane k of wave 0 ends with the block's ICP total of value k, lane 32 + k with the
        // photometric one — exactly the slot (tid) their words have in the shard
        const double r_rgb = canon::wave_sum<6, P>(rows, found, s_bias[1]);
        const int lane = tid & 63, wid = tid >> 6;
        // (s_red2 was last read before the barrier that follows the totals: no barrier needed in front of these stores)
        if ((lane & 1) == 0) {
          const int sl = canon::slot_of_lane<32>(lane);
          s_red2[0][wid][sl] = r_icp;
          s_red2[1][wid][sl] = r_rgb;
        }
        __syncthreads();
        if (tid < 64 && (tid & 31) < kSE3) {
          double t = 0.0;
#pragma unroll
          for (int w = 0; w < kPWaves; ++w) t += s_red2[tid >> 5][w][tid & 31];
          __hip_atomic_fetch_add(arw + tid, canon::pack_word(canon::to_units(t, tid < 32 ? eb_icp : eb_rgb)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
      const double p_rgb = canon::block_sum<6, P, kPWaves>(rows, found, s_bias[1], s_red);
      if (tid < kSE3) __hip_atomic_fetch_add(arw + 32 + tid, canon::pack_word(canon::to_units(p_rgb, eb_rgb)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    phase(4);
    if (stamp) L.prof[48 + blockIdx.x * 8 + 2] = wall_clock64();
    // ---- the totals arrive in the lanes that poll them ----
    if (tid < 64) {
      const int k = tid & 31;
      const bool mine = k < kSE3 && (tid < 32 ? ICP : RGB);
      unsigned long long fld;
      poll_pause(L.first_delay);
      ar_wait<kArStride>(arp, tid, mine, nb, fld, &st->sync_timeout);
      const long long tot = ar_total(fld, nb);
      const float val = mine ? canon::from_units(tot, tid < 32 ? eb_icp : eb_rgb) : 0.f;
      s_sums[tid] = val;
      {  // entry k of the combined system in lane k < 27: its photometric total sits in lane k + 32 (gn_scalar.hpp, comb_entry)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(val), __float_as_uint(val), false, false);
        if (tid < 27) s_comb[tid] = sc::comb_entry(tid, ICP, RGB, L.icpWeight, val, __uint_as_float(sw[1]));
      }
      const bool bad = mine && ((se3_diag_mask() >> k) & 1u) && tot >= canon::kViolation;
      const unsigned long long any = __builtin_amdgcn_ballot_w64(bad);
      if (tid == 0) s_viol = ((any & 0xffffffffull) ? 1 : 0) | ((any >> 32) ? 2 : 0);
    }
    __syncthreads();
    phase(5);
    if (stamp) L.prof[48 + blockIdx.x * 8 + 3] = wall_clock64();
    // Re-arm the ring: every block has arrived in THIS iteration's totals, so every block finished reading the previous iteration's set
    // (a block adds to a set only after it has taken the totals of the set before).  Block 0 zeroes it - only when the level will come
    // round to it again - and releases the stores; the set is next used kArRing - 1 iterations (>= 60 us) later, by blocks that have seen
    // block 0's later arrivals.
    if (L.n_iter > kArRing && blockIdx.x == 0 && it >= 1 && it - 1 + kArRing < L.n_iter) {
      unsigned long long* prev = L.ar + (size_t)((it - 1) % kArRing) * kArWords;
      for (int w = tid; w < kArWords; w += kPB) prev[w] = 0ull;
      __threadfence();
    }
    if (s_viol) {
      // a diagonal total does not fit the grid its exponents promised: every block saw the same totals, so the whole
      // grid raises the exponents of that reduction and repeats the iteration on a word set of the pool (uniform)
      const int viol = s_viol;
      __syncthreads();  // (s_viol is rewritten by the repeated iteration)
      pool_used += 1;
      on_pool = true;
      if (pool_used > kArPool) {  // no word set left: the call fails like a barrier timeout (prior pose kept, nothing fused)
        if (tid == 0) __hip_atomic_store(&st->sync_timeout, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ended = true;
        break;
      }
      if (tid == 0) {
        if (viol & 1) canon::retry_step<6>(s_E[0]);
        if (viol & 2) canon::retry_step<6>(s_E[1]);
        s_retries += (viol & 1) + ((viol >> 1) & 1);  // (counted per reduction, as the oracle does)
      }
      __syncthreads();
      if (tid < 64) {
        canon::write_bias<6>(s_E[tid >> 5], s_bias[tid >> 5], tid & 31);
        eb_icp = canon::value_exp<6>(s_E[0], tid & 31);
        eb_rgb = canon::value_exp<6>(s_E[1], tid & 31);
      }
      __syncthreads();
      continue;
    }
    on_pool = false;
    if (tid == 0) {
      SolveArgs q;
      q.icp = ICP ? 1 : 0;
      q.rgb = RGB ? 1 : 0;
      q.rgbOnly = L.rgbOnly;
      q.icpWeight = L.icpWeight;
      q.level = L.level;
      q.first_iter = it == 0;
      // No correspondence of either kind: the system is all zeros and the update exactly the identity, so every
      // remaining iteration of this level would reproduce this one bit for bit (same state, same sums, same side
      // outputs) — e.g. the model-to-model pass of the full frame step while the INACTIVE view is empty.  This
      // iteration then prepares the next level's projection parameters like the level's last one, the remaining
      // ones are accounted for, and every block leaves (uniform: s_none is read after the barrier below).
      const bool none = EXIT && !L.rgbOnly && it < L.n_iter - 1 && (!ICP || s_sums[28] == 0.f) && (!RGB || rgbSize == 0);
      q.next_level = (it == L.n_iter - 1 || none) ? L.level_below : L.level;
      q.level_below = L.level_below;
      q.fx = L.fx;
      q.fy = L.fy;
      q.cx = L.cx;
      q.cy = L.cy;
      const bool last = it == L.n_iter - 1 || none;
      const sc::KPre kp = s_k[last ? 1 : 0];
      // (with rgbOnly the level may end at any iteration's break: the side outputs are then the previous iteration's)
      sc::gn_step_combined(s, s_comb, s_sums[27], s_sums[28], rgbSize, sigma, q, kp, last || L.rgbOnly);
      if (none) {
        s.iters_run += L.n_iter - 1 - it;
        s_none = 1;
      }
    } else if (tid >= 64 && tid < 128) {
      // wave 1, beside the solve: exponents and bias table of the next reduction from this one's totals (canon.hpp)
      const int set_ = (tid - 64) >> 5, k = tid & 31;
      if ((set_ == 0 && ICP) || (set_ == 1 && RGB)) {
        int E[7];
#pragma unroll
        for (int c2 = 0; c2 < 7; ++c2) E[c2] = s_E[set_][c2];
        canon::next_exponents<6>(s_sums + 32 * set_, E);
        canon::write_bias<6>(E, s_bias[set_], k);
        if (k < 7) s_E[set_][k] = E[k];
      }
    }
    __syncthreads();
    if (tid < 64) {
      eb_icp = canon::value_exp<6>(s_E[0], tid & 31);
      eb_rgb = canon::value_exp<6>(s_E[1], tid & 31);
    }
    phase(7);
    ++it;
    if constexpr (EXIT) {
      if (s_none) break;
    }
  }

  // ---- block 0 hands the state to the next kernel on the stream ----
  __syncthreads();
  if ((FUSED || blockIdx.x == 0) && tid == 0) {
    if (ended) sv->level_done[L.level] = 1;
    for (int i = 0; i < 16; ++i) sv->resultRt[i] = s.resultRt[i];
    for (int i = 0; i < 9; ++i) {
      sv->Rcurr[i] = s.Rcurr[i];
      sv->krkinv[i] = s.krkinv[i];
    }
    for (int i = 0; i < 3; ++i) {
      sv->tcurr[i] = s.tcurr[i];
      sv->kt[i] = s.kt[i];
    }
    sv->lastRGBError = s.lastRGBError;
    sv->lastRGBCount = s.lastRGBCount;
    sv->lastICPError = s.lastICPError;
    sv->lastICPCount = s.lastICPCount;
    sv->iters_run[L.level] = s.iters_run;
    for (int i = 0; i < 36; ++i) sv->lastA[i] = s.lastA[i];
    for (int i = 0; i < 6; ++i) sv->lastb[i] = s.lastb[i];
    for (int c2 = 0; c2 < 7; ++c2) {
      sv->E_icp[c2] = s_E[0][c2];
      sv->E_rgb[c2] = s_E[1][c2];
    }
    sv->have_E = 1;
    sv->canon_retries += s_retries;
    if (L.finalize && blockIdx.x == 0) {  // == k_track_finalize, from the values this thread holds
      float tc[3], Rc[9];
      for (int i = 0; i < 3; ++i) tc[i] = s.tcurr[i];
      for (int i = 0; i < 9; ++i) Rc[i] = s.Rcurr[i];
      const float dx = tc[0] - s.tprev[0], dy = tc[1] - s.tprev[1], dz = tc[2] - s.tprev[2];
      const float n = sqrtf(dx * dx + dy * dy + dz * dz);
      // a grid-barrier timeout anywhere in this call (sticky flag, set before the waiting block gave up) leaves sums
      // that are not the sums of the image: the prior pose is kept and the frame step fuses nothing
      const bool timed_out = __hip_atomic_load(&st->sync_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (timed_out || (L.fin_rgb && (double)n > 0.3)) {  // RGBDOdometry.cpp:589-593
        for (int i = 0; i < 9; ++i) Rc[i] = sv->Rcurr[i] = s.Rprev[i];
        for (int i = 0; i < 3; ++i) tc[i] = sv->tcurr[i] = s.tprev[i];
        if (!timed_out) sv->rejected_jump = 1;
      }
      for (int i = 0; i < 3; ++i) sv->out_trans[i] = tc[i];
      for (int i = 0; i < 9; ++i) sv->out_rot[i] = Rc[i];
      // The frame step's pose block IS context.currPose(): the reference assigns the tracker's result to its top three rows only
      // (`currPose.topRightCorner(3, 1) = trans; currPose.topLeftCorner(3, 3) = rot`, ElasticFusion.cpp:246-247), so a bottom row
      // that is not exactly (0 0 0 1) - after a map merge, currPose = relativeTransform * currPose with relativeTransform =
      // recoveryPose * currPose.inverse(), a general 4 x 4 inverse (ReferenceFrame.h:95, :131) - stays with the camera, and
      // pose.inverse() (every shader's t_inv) sees it.  A pose matrix of the caller's own gets a fresh bottom row.
      const bool own = L.frame && L.pose16_out == L.frame->cur.pose;
      if (L.pose16_out) {
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 3; ++j) L.pose16_out[i * 4 + j] = Rc[i * 3 + j];
          L.pose16_out[i * 4 + 3] = tc[i];
        }
        if (!own) {
          L.pose16_out[12] = 0.f;
          L.pose16_out[13] = 0.f;
          L.pose16_out[14] = 0.f;
          L.pose16_out[15] = 1.f;
        }
      }
      if (L.frame) {
        // (the pose block the frame step hands in IS the frame state's current pose: then this block holds both matrices;
        // s_held[12..15] took the block's bottom row at the start of the kernel)
        if (own) {
          for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) s_held[i * 4 + j] = Rc[i * 3 + j];
            s_held[i * 4 + 3] = tc[i];
          }
        }
        frame_after_track_body(L.frame, L.weightMultiplier, __hip_atomic_load(&st->sync_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), own ? s_held : nullptr);
      }
    }
  }
  phase(8);
  if (L.prof && blockIdx.x == 0 && tid == 0) {
    phase(9);  // = cost of one phase() call itself
    for (int i = 0; i < 10; ++i) L.prof[L.level * 16 + i] += s_prof[i];
  }
}

template <bool ICP, bool RGB, int P, bool EXIT>
__global__ __launch_bounds__(kPB) void k_gn_level(TrackState* st, GnArgs a, LevelArgs L) {
  gn_level_body<ICP, RGB, P, EXIT, false>(st, st, a, L, (int)gridDim.x);
}

// Independent work riding on the SO3 launch when the call's set-up was folded into the model pyramid kernel (frame step):
// the pyramid's deferred last step in gx * gy blocks of 64 x 8 pixels behind the nb_so3 resident blocks (gx = 0: none),
// and the re-arming of the frame step's dense counters.
struct So3Extra {
  int nb_so3, gx, gy;
  unsigned* zero16;
  View<const float> dsrc;
  View<float> ddst;
  View<const unsigned char> isrc;
  View<unsigned char> idst;
};

// ---------------------------------------------------------------------------------------
// Persistent SO3 pre-alignment: all (<= 10) iterations in one launch, same protocol as k_gn_level
// with one all-reduce per iteration.  The state block is copied into LDS
// and the unchanged scalar code runs on the copy; block 0 writes it back at the end.
// ---------------------------------------------------------------------------------------
struct So3Args {
  const unsigned char* lastImage;
  size_t last_pitch;
  const unsigned char* nextImage;
  size_t next_pitch;
  int cols, rows;
  unsigned long long* ar;
  SolveCam cam;
  int first_gn_level, max_iter, exp_bias, first_delay;
};

// The SO3 stage: `s` is the block's LDS copy of the call's state (every block holds the same bits); stand-alone, it is loaded from
// `st` at the start and block 0 writes it back at the end; inside k_track_coarse (FUSED) it simply stays where it is for the next stage.
template <bool FUSED>
__device__ __forceinline__ void so3_body(TrackState* st, TrackState& s, const So3Args& q, const int nb) {
  const unsigned char* lastImage = q.lastImage;
  const unsigned char* nextImage = q.nextImage;
  const size_t last_pitch = q.last_pitch, next_pitch = q.next_pitch;
  const int cols = q.cols, rows = q.rows, first_gn_level = q.first_gn_level, max_iter = q.max_iter, exp_bias = q.exp_bias, first_delay = q.first_delay;
  unsigned long long* ar = q.ar;
  const SolveCam cam = q.cam;
  __shared__ int s_viol;
  __shared__ int s_E[4];
  __shared__ double s_bias[2][32];
  __shared__ double s_red[kPWaves][32];
  __shared__ float s_sums[32];
  int eb_mine = 0;  // wave 0, lane k < 11: bound exponent of value k
  const int tid = threadIdx.x;
  if (!FUSED) {
    const int* src = reinterpret_cast<const int*>(st);
    int* dst = reinterpret_cast<int*>(&s);
    for (int i = tid; i < (int)(sizeof(TrackState) / 4); i += kPB) dst[i] = src[i];
  }
  if (tid == 0) {
    int E[4];
    canon::static_so3(cols * rows, E);
    for (int c = 0; c < 4; ++c) s_E[c] = canon::clamp_e(E[c] + exp_bias);
  }
  __syncthreads();
  if (s.so3_done) return;
  if (tid < 32) {
    canon::write_bias<3>(s_E, s_bias, tid);
    eb_mine = canon::value_exp<3>(s_E, tid);
  }
  __syncthreads();
  const int N = cols * rows;
  const int i = blockIdx.x * kPB + tid;
  const bool live = i < N;
  const int ic = live ? i : i % N;  // (masked lanes shadow distinct pixels: see gn_level_body)
  const int y = ic / cols, x = ic - y * cols;
  int pool_used = 0, retries = 0;
  bool on_pool = false, failed = false;
  for (int it = 0; it < max_iter;) {
    const So3Params p = so3_params_of(&s, cols, rows);
    float rows_[1][4];
    bool found[1];
    found[0] = so3_row(p, lastImage, last_pitch, nextImage, next_pitch, x, y, rows_[0]);
    if (!live) {
      found[0] = false;
      rows_[0][0] = rows_[0][1] = rows_[0][2] = rows_[0][3] = 0.f;
    }
    const size_t set = (size_t)(on_pool ? kArRing + pool_used - 1 : it) * kArWords;  // (max_iter <= kArRing)
    const double pt = canon::block_sum<3, 1, kPWaves>(rows_, found, s_bias, s_red);
    if (tid < kSO3)
      __hip_atomic_fetch_add(ar + set + (size_t)(blockIdx.x & (kArShards - 1)) * kArStride + tid, canon::pack_word(canon::to_units(pt, eb_mine)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      unsigned long long fld;
      poll_pause(first_delay);
      ar_wait<kArStride>(ar + set, tid, tid < kSO3, nb, fld, &st->sync_timeout);
      const long long tot = ar_total(fld, nb);
      if (tid < 32) s_sums[tid] = tid < kSO3 ? canon::from_units(tot, eb_mine) : 0.f;
      const bool bad = tid < kSO3 && ((so3_diag_mask() >> tid) & 1u) && tot >= canon::kViolation;
      const unsigned long long any = __builtin_amdgcn_ballot_w64(bad);
      if (tid == 0) s_viol = any != 0ull ? 1 : 0;
    }
    __syncthreads();
    if (s_viol) {
      __syncthreads();
      pool_used += 1;
      retries += 1;
      on_pool = true;
      if (pool_used > kArPool) {
        failed = true;
        break;
      }
      if (tid == 0) canon::retry_step<3>(s_E);
      __syncthreads();
      if (tid < 32) {
        canon::write_bias<3>(s_E, s_bias, tid);
        eb_mine = canon::value_exp<3>(s_E, tid);
      }
      __syncthreads();
      continue;
    }
    on_pool = false;
    if (tid == 0) {
      sc::so3_solve_core(&s, s_sums, cam.fx, cam.fy, cam.cx, cam.cy, it == max_iter - 1 ? 1 : 0, first_gn_level);
    } else if (tid >= 64 && tid < 96) {
      int E[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) E[c] = s_E[c];
      canon::next_exponents<3>(s_sums, E);
      canon::write_bias<3>(E, s_bias, tid - 64);
      if (tid - 64 < 4) s_E[tid - 64] = E[tid - 64];
    }
    __syncthreads();
    if (s.so3_done) break;
    if (tid < 32) eb_mine = canon::value_exp<3>(s_E, tid);
    ++it;
  }
  __syncthreads();
  if (FUSED || blockIdx.x == 0) {
    if (tid == 0) {
      if (failed) __hip_atomic_store(&st->sync_timeout, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s.sync_timeout = __hip_atomic_load(&st->sync_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s.canon_retries += retries;
      if (failed && !s.so3_done) {  // the Gauss-Newton levels still need their projection parameters
        s.so3_done = 1;
        double Rt[16];
        for (int q2 = 0; q2 < 16; ++q2) Rt[q2] = s.resultRt[q2];
        sc::gn_params(Rt, sc::kpre_of(cam.fx, cam.fy, cam.cx, cam.cy, first_gn_level), s.krkinv, s.kt);
      }
    }
    __syncthreads();
    if (!FUSED) {
      const int* src = reinterpret_cast<const int*>(&s);
      int* dst = reinterpret_cast<int*>(st);
      for (int k = tid; k < (int)(sizeof(TrackState) / 4); k += kPB) dst[k] = src[k];
    }
  }
}

__global__ __launch_bounds__(kPB) void k_so3_level(TrackState* st, const unsigned char* lastImage, size_t last_pitch,
                                                   const unsigned char* nextImage, size_t next_pitch, int cols, int rows,
                                                   unsigned long long* ar, SolveCam cam, int first_gn_level, int max_iter, int exp_bias, int first_delay, So3Extra ex) {
  // Blocks past the resident grid carry independent work of the same call: the model pyramid's last step, whose output the
  // first Gauss-Newton level reads (not this kernel) — see So3Extra.  They take no part in the all-reduce.
  if ((int)blockIdx.x >= ex.nb_so3) {
    const int c = (int)blockIdx.x - ex.nb_so3, by = c / ex.gx, bx = c - by * ex.gx;
    model_pyr_step_pixel(bx * 64 + (int)(threadIdx.x & 63), by * (kPB / 64) + (int)(threadIdx.x >> 6), ex.dsrc, ex.ddst, ex.isrc, ex.idst);
    return;
  }
  if (ex.zero16 && blockIdx.x == 0 && threadIdx.x < 16) ex.zero16[threadIdx.x * 16] = 0u;  // the frame step's dense counters (read before this launch)
  __shared__ TrackState s;
  const So3Args q = {lastImage, last_pitch, nextImage, next_pitch, cols, rows, ar, cam, first_gn_level, max_iter, exp_bias, first_delay};
  so3_body<false>(st, s, q, ex.nb_so3);
}

// ---------------------------------------------------------------------------------------
// SO3 BESIDE the model pyramid (round 6).  The SO3 pre-alignment compares two LIVE image pyramids and starts from the prior pose: it
// needs nothing of the model prediction - yet it ran after the model pyramid kernel, a launch of ~16 us on the frame's critical path
// behind one of ~14.  Here the resident SO3 blocks are the first blocks of ONE launch whose other blocks are the model pyramid's
// groups (model_bodies.hpp, as 64 x 8 blocks): the two halves run side by side, the launch ends when the longer one does.  What made
// the old order necessary is arranged elsewhere: the call's set-up (state block, all-reduce words, timeout flag) rides on the
// frame's FIRST kernel (the tracking prediction's resolve pass: odometry_early_init_args), so it is complete - across a kernel
// boundary - before this launch starts; the pyramid's last step, which rode on the SO3 launch, is taken straight from the sources
// (model_pyr_step2_body); the dense counters this launch reads are re-armed by the first Gauss-Newton launch (LevelArgs::zero16).
struct ModelGroups {
  ModelSrc m;
  int g0x, g0y, g12x, g12y, gsx, gsy, g2x, g2y;
  int rows0, cols0, rows1, cols1, rows2, cols2;
  float cutOff;
  View<float> v0, n0, depth0, v1, n1, v2, n2, depth1, depth2;
  View<unsigned char> inten0, inten1, inten2;
};

__global__ __launch_bounds__(kPB) void k_so3_model(TrackState* st, So3Args q, int nb_so3, ModelGroups G) {
  if ((int)blockIdx.x < nb_so3) {
    __shared__ TrackState s;
    so3_body<false>(st, s, q, nb_so3);
    return;
  }
  const int t = (int)threadIdx.x, tx = t & 63, ty = t >> 6;
  constexpr int BYv = kPB / 64;
  const int b = (int)blockIdx.x - nb_so3;
  if (G.m.dense_cnt && b == 0 && t == 0) *G.m.flag_out = model_use_b(G.m) ? 1 : 0;
  const int nb12 = G.g12x * G.g12y, nbs = G.gsx * G.gsy, nb2 = G.g2x * G.g2y;
  if (b < nb12) {
    model_levels12_body(tx, ty, BYv, b % G.g12x, b / G.g12x, G.m, G.cols0, G.rows1, G.cols1, G.rows2, G.cols2, G.v1, G.n1, G.v2, G.n2);
  } else if (b < nb12 + nbs) {
    const int c = b - nb12;
    model_pyr_step1_body<BYv>(tx, ty, c % G.gsx, c / G.gsx, G.m, G.rows0, G.cols0, G.cutOff, G.depth1, G.inten1);
  } else if (b < nb12 + nbs + nb2) {
    const int c = b - nb12 - nbs;
    model_pyr_step2_body<kPB>(t, c % G.g2x, c / G.g2x, G.m, G.rows0, G.cols0, G.cutOff, G.rows1, G.cols1, G.depth2, G.inten2);
  } else {
    const int c = b - nb12 - nbs - nb2;
    model_level0_body(tx, ty, BYv, c % G.g0x, c / G.g0x, G.m, G.rows0, G.cols0, G.v0, G.n0, G.depth0, G.inten0, G.cutOff);
  }
}

// ---------------------------------------------------------------------------------------
// The coarse half of a call in ONE resident launch (round 6): SO3 pre-alignment, level 2, level 1.  Every dependent launch on a
// stream costs ~3.5 us of dispatch on this device plus the stage's own hand-off through HBM (state written back by block 0, reloaded
// by every block of the next launch, ~2 us) - 92 us of launches for a quarter of level 0's pixel work.  The stages need nothing from
// each other but the state block, and every block already holds it bit for bit (each block solves on identical totals): so the
// stages simply follow one another inside one launch, the state stays in LDS, and the blocks never meet between two stages (each
// stage's all-reduces have word sets of their own).  Grid = the largest stage's (level 1); a stage with fewer pixels leaves the
// blocks past its image empty-handed - they still arrive in its all-reduces (with zeros).  Level 0 keeps its launch (three pixels
// per thread on a larger grid).  The rider blocks of the SO3 launch (So3Extra: the model pyramid's deferred last step, which LEVEL 2
// reads) are riders of this launch: they release their output and count themselves in, and the resident blocks wait for that count
// between the SO3 stage and level 2 (long complete by then: a formality with a bounded spin).
struct CoarseArgs {
  So3Args so3;
  GnArgs a2, a1;
  LevelArgs L2, L1;
  int run_so3, run_l2, run_l1;
  unsigned* rider_cnt;     // counts rider blocks that have finished, over the life of the handle
  unsigned rider_target;   // its value once this launch's riders are done
};

template <bool ICP, bool RGB, int P1, bool EXIT>
__global__ __launch_bounds__(kPB) void k_track_coarse(TrackState* st, CoarseArgs F, So3Extra ex) {
  if ((int)blockIdx.x >= ex.nb_so3) {
    const int c = (int)blockIdx.x - ex.nb_so3, by = c / ex.gx, bx = c - by * ex.gx;
    model_pyr_step_pixel(bx * 64 + (int)(threadIdx.x & 63), by * (kPB / 64) + (int)(threadIdx.x >> 6), ex.dsrc, ex.ddst, ex.isrc, ex.idst);
    __threadfence();  // (release: the stores above reach memory every XCD reads from)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(F.rider_cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (ex.zero16 && blockIdx.x == 0 && threadIdx.x < 16) ex.zero16[threadIdx.x * 16] = 0u;
  __shared__ TrackState s;
  const int tid = threadIdx.x, nb = ex.nb_so3;
  {
    const int* src = reinterpret_cast<const int*>(st);
    int* dst = reinterpret_cast<int*>(&s);
    for (int i = tid; i < (int)(sizeof(TrackState) / 4); i += kPB) dst[i] = src[i];
  }
  __syncthreads();
  if (F.run_so3) so3_body<true>(st, s, F.so3, nb);
  __syncthreads();
  if (ex.gx > 0) {  // level 2 reads what this launch's rider blocks wrote
    if (tid == 0) {
      unsigned spins = 0;
      // (wrap-safe comparison: the counter runs over the life of the handle)
      while ((int)(__hip_atomic_load(F.rider_cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - F.rider_target) < 0) {
        if (++spins > kSpinLimit) {
          __hip_atomic_store(&st->sync_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // never hang the device
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (F.run_l2) gn_level_body<ICP, RGB, 1, EXIT, true>(st, &s, F.a2, F.L2, nb);
  __syncthreads();
  if (F.run_l1) gn_level_body<ICP, RGB, P1, EXIT, true>(st, &s, F.a1, F.L1, nb);
  __syncthreads();
  if (blockIdx.x == 0) {  // the state goes back to HBM for the level-0 launch
    if (tid == 0) s.sync_timeout = __hip_atomic_load(&st->sync_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int* src = reinterpret_cast<const int*>(&s);
    int* dst = reinterpret_cast<int*>(st);
    for (int k = tid; k < (int)(sizeof(TrackState) / 4); k += kPB) dst[k] = src[k];
  }
}

__global__ void k_track_finalize(TrackState* st, int rgb, float* __restrict__ pose16_out, FrameState* frame, float weightMultiplier) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float dx = st->tcurr[0] - st->tprev[0], dy = st->tcurr[1] - st->tprev[1], dz = st->tcurr[2] - st->tprev[2];
  const float n = sqrtf(dx * dx + dy * dy + dz * dz);
  const bool timed_out = st->sync_timeout != 0;  // (see the finalize step of k_gn_level)
  if (timed_out || (rgb && (double)n > 0.3)) {  // RGBDOdometry.cpp:589-593
    for (int i = 0; i < 9; ++i) st->Rcurr[i] = st->Rprev[i];
    for (int i = 0; i < 3; ++i) st->tcurr[i] = st->tprev[i];
    if (!timed_out) st->rejected_jump = 1;
  }
  for (int i = 0; i < 3; ++i) st->out_trans[i] = st->tcurr[i];
  for (int i = 0; i < 9; ++i) st->out_rot[i] = st->Rcurr[i];
  if (pose16_out) {  // frame step: the pose goes straight into the context's pose block (row-major 4x4)
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) pose16_out[i * 4 + j] = st->out_rot[i * 3 + j];
      pose16_out[i * 4 + 3] = st->out_trans[i];
    }
    if (!(frame && pose16_out == frame->cur.pose)) {  // (currPose keeps its bottom row: see the finalize step of k_gn_level)
      pose16_out[12] = 0.f;
      pose16_out[13] = 0.f;
      pose16_out[14] = 0.f;
      pose16_out[15] = 1.f;
    }
  }
  // frame step: pose16_out is frame->cur.pose; derive its inverse and the velocity weight here
  // instead of in a launch of their own
  if (frame) frame_after_track_body(frame, weightMultiplier, st->sync_timeout);
}

}  // namespace dms

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
  size_t off = 0;
  char* base = nullptr;
  void* take(size_t bytes) {
    off = align_up(off, 256);
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

void layout(dms_odometry* o, Carver& c) {
  const int W = o->width, H = o->height;
  auto mk = [&](Buf& b, int rows, int cols, size_t elem) {
    b.rows = rows;
    b.cols = cols;
    b.pitch = (size_t)cols * elem;
    b.p = c.take(b.pitch * rows);
  };
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    const int r = H >> i, w = W >> i;
    mk(o->depth_tmp[i], r, w, 2);
    mk(o->vmaps_g_prev[i], r * 3, w, 4);
    mk(o->nmaps_g_prev[i], r * 3, w, 4);
    mk(o->vmaps_curr[i], r * 3, w, 4);
    mk(o->nmaps_curr[i], r * 3, w, 4);
    mk(o->lastDepth[i], r, w, 4);
    mk(o->nextDepth[i], r, w, 4);
    mk(o->lastImage[i], r, w, 1);
    mk(o->nextImage[i], r, w, 1);
    mk(o->lastNextImage[i], r, w, 1);
    mk(o->nextdIdx[i], r, w, 2);
    mk(o->nextdIdy[i], r, w, 2);
    mk(o->nextGate[i], r, w, 1);
    mk(o->pointClouds[i], r, w, 12);
    mk(o->corresImg[i], r, w, sizeof(dms_dataterm));
  }
  o->vmaps_tmp = (float*)c.take((size_t)W * H * 16);
  o->nmaps_tmp = (float*)c.take((size_t)W * H * 16);
  o->part_icp = (long long*)c.take((size_t)kRecWords * kMaxPartialBlocks * 8);
  o->part_rgb = (long long*)c.take((size_t)kRecWords * kMaxPartialBlocks * 8);
  o->part_so3 = (long long*)c.take((size_t)kRecWords * kMaxPartialBlocks * 8);
  o->part_cnt = (int*)c.take((size_t)2 * kMaxPartialBlocks * 4);
  o->tickets = (unsigned*)c.take(64);
  o->rider_cnt = (unsigned*)c.take(64);
  o->ar = (unsigned long long*)c.take((size_t)kArSets * kArWords * 8);
  o->prof = (long long*)c.take((3 * 16 + 256 * 8) * 8);
  o->state = (TrackState*)c.take(sizeof(TrackState));
}

struct Timer {
  dms_odometry* o;
  hipStream_t s;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  bool on() const { return o->profiling && (!o->profiling_level0_only || strcmp(name, "gn_level0") == 0); }
  Timer(dms_odometry* o_, hipStream_t s_, const char* n) : o(o_), s(s_), name(n) {
    if (!on()) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!o->event_pool.empty()) {
        e = o->event_pool.back();
        o->event_pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, s);
  }
  ~Timer() {
    if (!on()) return;
    (void)hipEventRecord(b, s);
    o->pending.push_back({name, {a, b}});
  }
};

void drain_timers(dms_odometry* o) {
  for (auto& p : o->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
      KernelTime& t = o->times[p.first];
      t.ms += ms;
      t.launches += 1;
      if (t.samples.size() < 100000) t.samples.push_back(ms);
    }
    o->event_pool.push_back(p.second.first);
    o->event_pool.push_back(p.second.second);
  }
  o->pending.clear();
}

// minimum squared gradient of level l (RGBDOdometry.cpp:287-291 / :453)
float rgb_min_scale(const dms_odometry* o, int l) { return (float)(pow((double)o->minGrad[l], 2.0) / pow((double)o->sobelScale, 2.0)); }

int populateDepth(dms_odometry* o, Buf* destDepths, hipStream_t s) {
  int rc;
  dms_image2d d0 = destDepths[0].img();
  if ((rc = verticesToDepth(o->vmaps_tmp, &d0, o->maxDepthRGB, s))) return rc;
  for (int i = 0; i + 1 < DMS_NUM_PYRS; i++) {
    dms_image2d a = destDepths[i].img(), b = destDepths[i + 1].img();
    if ((rc = pyrDownGaussF(&a, &b, s))) return rc;
  }
  return DMS_OK;
}

int populateImage(const dms_image2d* rgba, Buf* destImages, hipStream_t s) {
  int rc;
  dms_image2d i0 = destImages[0].img();
  if ((rc = imageToIntensity(rgba, &i0, s))) return rc;
  for (int i = 0; i + 1 < DMS_NUM_PYRS; i++) {
    dms_image2d a = destImages[i].img(), b = destImages[i + 1].img();
    if ((rc = pyrDownUcharGauss(&a, &b, s))) return rc;
  }
  return DMS_OK;
}

// RGBDOdometry::populateRGBDData (RGBDOdometry.cpp:139-160): depth from the LAST vertex map that
// initICPModel uploaded (vmaps_tmp), intensity from the given image
int populateRGBDData(dms_odometry* o, const dms_image2d* rgba, Buf* destDepths, Buf* destImages, hipStream_t s) {
  int rc;
  if ((rc = populateDepth(o, destDepths, s))) return rc;
  return populateImage(rgba, destImages, s);
}

// Sobel pyramid of nextImage (RGBDOdometry.cpp:279-283).  The reference computes it inside
// getIncrementalTransformation; here it is computed where nextImage is written and only redone
// by the tracker when nextImage is no longer the image it was computed from (SO3 handle swap).
int nextDerivatives(dms_odometry* o, hipStream_t s) {
  int rc;
  for (int i = 0; i < DMS_NUM_PYRS; i++) {
    dms_image2d a = o->nextImage[i].img(), dx = o->nextdIdx[i].img(), dy = o->nextdIdy[i].img(), g = o->nextGate[i].img();
    if ((rc = derivativeGate(&a, &dx, &dy, &g, rgb_min_scale(o, i), s))) return rc;
  }
  o->deriv_of = o->nextImage[0].p;
  return DMS_OK;
}

}  // namespace

extern "C" {

int dms_odometry_create(dms_odometry** out, int width, int height, float cx, float cy, float fx, float fy, float distThresh,
                        float angleThresh) {
  DMS_REQUIRE(out, "null out");
  DMS_REQUIRE(width >= 16 && height >= 16 && width <= 32767 && height <= 32767, "bad resolution");
  dms_odometry* o = new dms_odometry();
  o->width = width;
  o->height = height;
  o->cx = cx;
  o->cy = cy;
  o->fx = fx;
  o->fy = fy;
  o->distThres = distThresh > 0.f ? distThresh : 0.10f;
  o->angleThres = angleThresh > 0.f ? angleThresh : (float)sin(20.f * 3.14159254f / 180.f);
  o->sobelScale = (float)(1.0 / pow(2.0, 3));  // RGBDOdometry.cpp:34-35
  o->maxDepthDeltaRGB = 0.07f;
  o->maxDepthRGB = 6.0f;
  o->minGrad[0] = 5;
  o->minGrad[1] = 3;
  o->minGrad[2] = 1;
  Carver sz;
  layout(o, sz);
  o->arena_bytes = align_up(sz.off, 256);
  hipError_t e = hipMalloc((void**)&o->arena, o->arena_bytes);
  if (e != hipSuccess) {
    delete o;
    return hip_fail(e, "hipMalloc(arena)", __FILE__, __LINE__);
  }
  // pyramids start as zeros (the reference's cudaMallocPitch memory is uninitialised;
  // zero keeps the never-initialised-RGB case of GPUTest.cpp:247-286 deterministic)
  (void)hipMemset(o->arena, 0, o->arena_bytes);
  Carver c;
  c.base = o->arena;
  layout(o, c);
  e = hipHostMalloc((void**)&o->host_state, sizeof(TrackState), hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipFree(o->arena);
    delete o;
    return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
  }
  memset(o->host_state, 0, sizeof(TrackState));
  {  // execution switches: the environment is consulted here and nowhere else
    const char* e = getenv("DMS_TRACK_MODE");
    o->resident = !(e && strcmp(e, "launches") == 0);
    e = getenv("DMS_TRACK_EARLY_EXIT");
    o->early_exit_force = e ? (e[0] != '0' ? 1 : 0) : -1;
    e = getenv("DMS_PERSIST_BLOCKS");
    if (e && atoi(e) > 0) o->persist_target = atoi(e);
    {  // how many blocks of a resident kernel fit the device at once (see max_resident_blocks)
      int dev = 0, cus = 0, per_cu = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_gn_level<true, true, 3, false>, kPB, 0) != hipSuccess) per_cu = 0;
      (void)hipGetLastError();
      // (the API is known to answer one block per CU too many near register-file edges, MI355X_MICROARCH.md: a 512-thread block
      // at > 128 registers can only ever be alone on its CU, so one per CU is what is relied on)
      o->max_resident_blocks = per_cu >= 1 ? (cus < kMaxPersistBlocks ? cus : kMaxPersistBlocks) : 0;
      if (o->max_resident_blocks <= 0) o->resident = false;
    }
    e = getenv("DMS_PERSIST_MAX_BLOCKS");
    if (e && atoi(e) > 0 && atoi(e) < o->max_resident_blocks) o->max_resident_blocks = atoi(e);
    e = getenv("DMS_AR_FIRST_DELAY");  // units of 64 cycles before the first read of the totals; default: by grid size
    if (e && atoi(e) >= 0 && atoi(e) <= 1000) o->first_delay = atoi(e);
    e = getenv("DMS_TRACK_LONG_RESIDENT");
    if (e) o->long_levels_resident = atoi(e) != 0;
    e = getenv("DMS_SO3_BESIDE_MODEL");
    if (e) o->so3_beside_model = atoi(e) != 0;
    e = getenv("DMS_TRACK_FUSE");
    if (e) o->fuse_coarse = atoi(e) != 0;
    e = getenv("DMS_AR_FIRST_DELAY_BY_LEVEL");  // "l0,l1,l2,so3" in the same units (-1 keeps the rule for that stage): the per-level sweep
    if (e) {
      int v[4] = {-1, -1, -1, -1};
      if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) >= 1)
        for (int i = 0; i < 4; ++i)
          if (v[i] >= 0 && v[i] <= 1000) o->first_delay_lvl[i] = v[i];
    }
  }
  *out = o;
  return DMS_OK;
}

int dms_odometry_debug_set(dms_odometry* o, const char* key, int value) {
  DMS_REQUIRE(o && key, "null argument");
  if (strcmp(key, "exp_bias") == 0) {
    DMS_REQUIRE(value >= -100 && value <= 100, "exp_bias out of range");
    o->exp_bias = value;
    return DMS_OK;
  }
  if (strcmp(key, "depth_exp_bias") == 0) {  // the frame step's rule, kept apart from the test hook: the two add
    DMS_REQUIRE(value >= 0 && value <= 100, "depth_exp_bias out of range");
    o->depth_bias = value;
    return DMS_OK;
  }
  DMS_REQUIRE(false, "unknown key");
}

int dms_odometry_set_resident_budget(dms_odometry* o, int max_blocks, int unchained) {
  DMS_REQUIRE(o && max_blocks >= 0, "bad argument");
  o->budget_cap = max_blocks;
  o->unchained_ok = unchained != 0 && max_blocks > 0;
  return DMS_OK;
}

int dms_odometry_get_mode(dms_odometry* o, int* resident, int* max_resident_blocks, int* fell_back) {
  DMS_REQUIRE(o, "null argument");
  if (resident) *resident = o->resident ? 1 : 0;
  if (max_resident_blocks) *max_resident_blocks = o->max_resident_blocks;
  if (fell_back) *fell_back = o->fell_back ? 1 : 0;
  return DMS_OK;
}

int dms_odometry_canon_retries(dms_odometry* o, int* retries) {
  DMS_REQUIRE(o && retries, "null argument");
  *retries = o->host_state->canon_retries;
  return DMS_OK;
}

// host builds of the scalar section (test hooks, see dmslam.h)
int dms_debug_scalar_gn(const float* sums_icp, const float* sums_rgb, float icpWeight, const float* Rprev, const float* tprev,
                        double* resultRt, float fx, float fy, float cx, float cy, int next_level, double* A, double* b, float* Rcurr,
                        float* tcurr, float* krkinv, float* kt) {
  DMS_REQUIRE((sums_icp || sums_rgb) && Rprev && tprev && resultRt && A && b && Rcurr && tcurr && krkinv && kt, "null argument");
  sc::GnLocal L;
  memset(&L, 0, sizeof(L));
  for (int i = 0; i < 16; ++i) L.resultRt[i] = resultRt[i];
  for (int i = 0; i < 9; ++i) L.Rprev[i] = Rprev[i];
  for (int i = 0; i < 3; ++i) L.tprev[i] = tprev[i];
  sc::SolveArgs q;
  memset(&q, 0, sizeof(q));
  q.icp = sums_icp ? 1 : 0;
  q.rgb = sums_rgb ? 1 : 0;
  q.icpWeight = icpWeight;
  float zero[29] = {0};
  sc::gn_step_core(L, sums_icp ? sums_icp : zero, sums_rgb ? sums_rgb : zero, 1, 1, q, sc::kpre_of(fx, fy, cx, cy, next_level), true);
  for (int i = 0; i < 16; ++i) resultRt[i] = L.resultRt[i];
  for (int i = 0; i < 36; ++i) A[i] = L.lastA[i];
  for (int i = 0; i < 6; ++i) b[i] = L.lastb[i];
  for (int i = 0; i < 9; ++i) {
    Rcurr[i] = L.Rcurr[i];
    krkinv[i] = L.krkinv[i];
  }
  for (int i = 0; i < 3; ++i) {
    tcurr[i] = L.tcurr[i];
    kt[i] = L.kt[i];
  }
  return DMS_OK;
}

int dms_debug_scalar_so3(const float* sums, float* R_lr, double* resultR, float fx, float fy, float cx, float cy, float* imageBasis,
                         float* kinv, float* krlr) {
  DMS_REQUIRE(sums && R_lr && resultR && imageBasis && kinv && krlr, "null argument");
  TrackState st;
  memset(&st, 0, sizeof(st));
  for (int i = 0; i < 9; ++i) {
    st.R_lr[i] = R_lr[i];
    st.resultR[i] = st.lastResultR[i] = resultR[i];
  }
  st.so3_lastError = 3.402823466e+38F / 2;
  st.so3_lastCount = 3.402823466e+38F / 2;
  sc::so3_solve_core(&st, sums, fx, fy, cx, cy, 0, 0);
  for (int i = 0; i < 9; ++i) {
    R_lr[i] = st.R_lr[i];
    resultR[i] = st.resultR[i];
    imageBasis[i] = st.imageBasis[i];
    kinv[i] = st.kinv[i];
    krlr[i] = st.krlr[i];
  }
  return DMS_OK;
}

int dms_odometry_inject_timeout(dms_odometry* o, int calls) {
  DMS_REQUIRE(o && calls >= 0, "bad argument");
  o->inject_timeouts = calls;
  return DMS_OK;
}

int dms_odometry_set_exec(dms_odometry* o, int resident, int early_exit, int coarse_launch) {
  DMS_REQUIRE(o, "null argument");
  // resident = 1 on a handle whose device cannot hold a resident kernel's blocks at once (max_resident_blocks = 0) stays off
  if (resident >= 0) o->resident = resident != 0 && o->max_resident_blocks > 0;
  if (early_exit >= 0) o->early_exit_force = early_exit ? 1 : 0;  // -1: unchanged (keeps a DMS_TRACK_EARLY_EXIT choice made at creation)
  if (coarse_launch >= 0) o->fuse_coarse = coarse_launch != 0;
  return DMS_OK;
}

// deprecated form (rounds 1-2 had summation variants: fp64 block sums, a record protocol; every sum is the exact integer sum of canon.hpp now)
int dms_odometry_set_mode(dms_odometry* o, int resident, int fp64_sums, int early_exit, int atomic_reduce) {
  (void)fp64_sums;
  (void)atomic_reduce;
  return dms_odometry_set_exec(o, resident, early_exit, -1);
}

// the frame step's reaction to a resident kernel that timed out: launch-per-phase from here on, nothing else touched
int dms_odometry_fall_back_to_launches(dms_odometry* o) {
  DMS_REQUIRE(o, "null argument");
  o->resident = false;
  o->fell_back = true;
  return DMS_OK;
}

int dms_odometry_destroy(dms_odometry* o) {
  if (!o) return DMS_OK;
  drain_timers(o);
  for (auto e : o->event_pool) (void)hipEventDestroy(e);
  if (o->arena) (void)hipFree(o->arena);
  if (o->ring_arena) (void)hipFree(o->ring_arena);
  if (o->host_state) (void)hipHostFree(o->host_state);
  delete o;
  return DMS_OK;
}

int dms_odometry_initICP_depth(dms_odometry* o, const dms_image2d* depth, float depthCutoff, dms_stream st) {
  DMS_REQUIRE(o && depth && depth->data, "null argument");
  DMS_REQUIRE(depth->rows == o->height && depth->cols == o->width, "depth must be full resolution");
  hipStream_t s = (hipStream_t)st;
  DMS_HIP(hipMemcpy2DAsync(o->depth_tmp[0].p, o->depth_tmp[0].pitch, depth->data, depth->pitch, (size_t)o->width * 2, o->height,
                           hipMemcpyDeviceToDevice, s));
  int rc;
  for (int i = 1; i < DMS_NUM_PYRS; ++i) {
    dms_image2d a = o->depth_tmp[i - 1].img(), b = o->depth_tmp[i].img();
    if ((rc = pyrDown(&a, &b, s))) return rc;
  }
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    const int div = 1 << i;
    dms_camera k = {o->fx / div, o->fy / div, o->cx / div, o->cy / div};
    dms_image2d d = o->depth_tmp[i].img(), v = o->vmaps_curr[i].img(), n = o->nmaps_curr[i].img();
    if ((rc = createVMap(&k, &d, &v, depthCutoff, s))) return rc;
    if ((rc = createNMap(&v, &n, s))) return rc;
  }
  return DMS_OK;
}

int dms_odometry_initICP_maps(dms_odometry* o, const float* verts, const float* norms, float depthCutoff, dms_stream st) {
  (void)depthCutoff;
  DMS_REQUIRE(o && verts && norms, "null argument");
  hipStream_t s = (hipStream_t)st;
  const size_t bytes = (size_t)o->width * o->height * 16;
  DMS_HIP(hipMemcpyAsync(o->vmaps_tmp, verts, bytes, hipMemcpyDeviceToDevice, s));
  DMS_HIP(hipMemcpyAsync(o->nmaps_tmp, norms, bytes, hipMemcpyDeviceToDevice, s));
  int rc;
  dms_image2d v0 = o->vmaps_curr[0].img(), n0 = o->nmaps_curr[0].img();
  if ((rc = copyMaps(o->vmaps_tmp, o->nmaps_tmp, &v0, &n0, s))) return rc;
  for (int i = 1; i < DMS_NUM_PYRS; ++i) {
    dms_image2d va = o->vmaps_curr[i - 1].img(), vb = o->vmaps_curr[i].img();
    dms_image2d na = o->nmaps_curr[i - 1].img(), nb = o->nmaps_curr[i].img();
    if ((rc = resizeMap(&va, &vb, false, s))) return rc;
    if ((rc = resizeMap(&na, &nb, true, s))) return rc;
  }
  return DMS_OK;
}

int dms_odometry_initICPModel(dms_odometry* o, const float* verts, const float* norms, float depthCutoff, const float* modelPose,
                              dms_stream st) {
  (void)depthCutoff;
  DMS_REQUIRE(o && verts && norms && modelPose, "null argument");
  hipStream_t s = (hipStream_t)st;
  const size_t bytes = (size_t)o->width * o->height * 16;
  DMS_HIP(hipMemcpyAsync(o->vmaps_tmp, verts, bytes, hipMemcpyDeviceToDevice, s));
  DMS_HIP(hipMemcpyAsync(o->nmaps_tmp, norms, bytes, hipMemcpyDeviceToDevice, s));
  int rc;
  dms_image2d v0 = o->vmaps_g_prev[0].img(), n0 = o->nmaps_g_prev[0].img();
  if ((rc = copyMaps(o->vmaps_tmp, o->nmaps_tmp, &v0, &n0, s))) return rc;
  for (int i = 1; i < DMS_NUM_PYRS; ++i) {
    dms_image2d va = o->vmaps_g_prev[i - 1].img(), vb = o->vmaps_g_prev[i].img();
    dms_image2d na = o->nmaps_g_prev[i - 1].img(), nb = o->nmaps_g_prev[i].img();
    if ((rc = resizeMap(&va, &vb, false, s))) return rc;
    if ((rc = resizeMap(&na, &nb, true, s))) return rc;
  }
  dms_mat33 R;
  dms_float3 t;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R.m[i * 3 + j] = modelPose[i * 4 + j];
  t.x = modelPose[3];
  t.y = modelPose[7];
  t.z = modelPose[11];
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    dms_image2d v = o->vmaps_g_prev[i].img(), n = o->nmaps_g_prev[i].img();
    if ((rc = transformMaps(&v, &n, &R, &t, &v, &n, s))) return rc;
  }
  return DMS_OK;
}

int dms_odometry_initRGB(dms_odometry* o, const dms_image2d* rgba, dms_stream s) {
  DMS_REQUIRE(o && rgba && rgba->data, "null argument");
  DMS_REQUIRE(rgba->rows == o->height && rgba->cols == o->width, "rgba must be full resolution");
  int rc = populateRGBDData(o, rgba, o->nextDepth, o->nextImage, (hipStream_t)s);
  if (rc) return rc;
  return nextDerivatives(o, (hipStream_t)s);
}
int dms_odometry_initRGBModel(dms_odometry* o, const dms_image2d* rgba, dms_stream s) {
  DMS_REQUIRE(o && rgba && rgba->data, "null argument");
  DMS_REQUIRE(rgba->rows == o->height && rgba->cols == o->width, "rgba must be full resolution");
  return populateRGBDData(o, rgba, o->lastDepth, o->lastImage, (hipStream_t)s);
}
int dms_odometry_initFirstRGB(dms_odometry* o, const dms_image2d* rgba, dms_stream st) {
  DMS_REQUIRE(o && rgba && rgba->data, "null argument");
  DMS_REQUIRE(rgba->rows == o->height && rgba->cols == o->width, "rgba must be full resolution");
  hipStream_t s = (hipStream_t)st;
  int rc;
  dms_image2d i0 = o->lastNextImage[0].img();
  if ((rc = imageToIntensity(rgba, &i0, s))) return rc;
  for (int i = 0; i + 1 < DMS_NUM_PYRS; i++) {
    dms_image2d a = o->lastNextImage[i].img(), b = o->lastNextImage[i + 1].img();
    if ((rc = pyrDownUcharGauss(&a, &b, s))) return rc;
  }
  return DMS_OK;
}

}  // extern "C"

namespace dms {
struct TrackFold {  // the track call a model pyramid launch prepares (odometry_initModel_fused)
  const float* prior_pose16;
  int pyramid, fastOdom, so3, interMap;
};
int odometry_track_enqueue(dms_odometry* o, const float* trans, const float* rot, const float* prior_pose16_dev, int rgbOnly,
                           float icpWeight, int pyramid, int fastOdom, int so3, int interMap, hipStream_t s, FrameState* frame = nullptr,
                           float weightMultiplier = 1.f);
}

extern "C" {

int dms_odometry_track_async(dms_odometry* o, const float* trans, const float* rot, int rgbOnly, float icpWeight, int pyramid,
                             int fastOdom, int so3, int interMap, dms_stream st) {
  DMS_REQUIRE(o && trans && rot, "null argument");
  return odometry_track_enqueue(o, trans, rot, nullptr, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap, (hipStream_t)st);
}

}  // extern "C"

namespace dms {

// ---- persistent level kernels: launch shape and cross-stream serialisation ----
// dms_odometry::resident = false (DMS_TRACK_MODE=launches at creation) selects the three-launches-per-iteration path.

// pixels per thread (1 - 5) and grid of k_gn_level for an n-pixel level; 0 blocks = not eligible
static void persistent_shape(int n, int target, int max_blocks, int& P, int& nb, int reserve = 56) {
  // 1 or 2 pixels per thread if that keeps the grid at <= `target` blocks (cheap barriers win on the small levels; 96 while
  // the cross-block sums went through per-block records and a gather, 160 since the integer all-reduce: level 1 of
  // 640x480 on 150 blocks with one pixel per thread instead of 75 with two, +1 % frame rate);
  // otherwise 3 pixels per thread on up to 256 blocks (the full-resolution
  // level is bound by its per-CU arithmetic: measured 221 us at 200 blocks vs 230 us at 150), else 4
  auto blocks = [&](int p) { return (n + kPB * p - 1) / (kPB * p); };
  const int cap = max_blocks < kMaxPersistBlocks ? max_blocks : kMaxPersistBlocks;  // every block must be resident at once
  const int small = target < cap ? target : cap;
  // ... and among 3 / 4 / 5 the fewest that leaves `reserve` compute units to whatever overlaps the tracker (the frame step's
  // prep stream holds whole units for the length of its depth filter: 1241x376 on 228 blocks beside 97 filter blocks started
  // incomplete; 5 pixels per thread = 183 blocks leave 73), else the fewest that fits at all
  if (blocks(1) <= small)
    P = 1;
  else if (blocks(2) <= small)
    P = 2;
  else {
    P = 0;
    for (int p = 3; p <= 5 && !P; ++p)
      if (blocks(p) <= cap - reserve) P = p;
    for (int p = 3; p <= 5 && !P; ++p)
      if (blocks(p) <= cap) P = p;
    if (!P) P = 5;
  }
  nb = (n + kPB * P - 1) / (kPB * P);
  if (nb > cap || n >= (1 << 19)) nb = 0;  // does not fit the device (or the 19-bit count of the pair word): launch-per-phase
}

// Blocks of a persistent kernel spin on each other, so two of them must never share the device
// half-resident.  Every persistent section of this process is therefore chained on one event per
// device: stream-ordered, free for a single stream, and it serialises trackers of different
// cameras / streams against each other.
// A process that runs all its trackers on one stream (one camera per process, the usual case) needs no event at all:
// the stream orders the sections, and the unconditional record after every tracker call cost ~6 us of idle stream per
// frame in the rocprof trace.  The first time a section arrives on a stream other than the previous one, the device is
// synchronised once (the previous stream may have been destroyed since: its handle must not be touched any more) and the
// device switches to the chained form: from then on every section ends with an event recorded on its own — live —
// stream, and a section on another stream waits for it.
struct PersistChain {
  std::mutex mu;
  hipEvent_t ev[64] = {};
  hipStream_t last[64] = {};
  bool used[64] = {};
  bool chained[64] = {};   // several streams have run sections on this device: events from here on
  bool ev_valid[64] = {};
  // (round 6) ... until one stream has had the device's sections to itself for kQuietAfter calls in a row (a front end that changed
  // streams between phases; bench.py's loops): the events stop, and the next section from another stream pays one device
  // synchronisation as the very first change of stream did.  Two cameras that alternate never get there and keep their events.
  int run[64] = {};
  bool quiet[64] = {};
};
static constexpr int kQuietAfter = 8;
static PersistChain g_persist;
// DMS_PERSIST_UNCHAINED=1: the caller guarantees that the resident grids of all handles that may track at the same time fit the
// device TOGETHER (DMS_PERSIST_MAX_BLOCKS of each, summed, <= compute units; a handle whose bound exceeds half the device stays chained) — then every grid completes whatever the
// interleaving and no section has to wait for another camera's.  Read once.
static bool unchained() {
  static const bool u = [] {
    const char* e = getenv("DMS_PERSIST_UNCHAINED");
    return e && atoi(e) != 0;
  }();
  return u;
}
struct PersistSection {
  hipStream_t s;
  int dev = 0;
  bool active = false;
  int budget;  // largest resident grid of the handle this section belongs to
  bool handle_unchained;
  PersistSection(hipStream_t s_, int budget_, bool handle_unchained_ = false) : s(s_), budget(budget_), handle_unchained(handle_unchained_) {}
  void begin() {
    if (active) return;
    if (unchained() || handle_unchained) {  // honoured only for handles that hold at most half the device: two of them always fit together
      static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? n : 0;
      }();
      if (budget > 0 && 2 * budget <= cus) return;
    }
    g_persist.mu.lock();
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (g_persist.used[dev] && g_persist.last[dev] != s) {
      if (!g_persist.chained[dev] || g_persist.quiet[dev]) {
        (void)hipDeviceSynchronize();
        g_persist.chained[dev] = true;
      } else if (g_persist.ev_valid[dev]) {
        (void)hipStreamWaitEvent(s, g_persist.ev[dev], 0);
      }
      g_persist.run[dev] = 0;
      g_persist.quiet[dev] = false;
    } else if (g_persist.chained[dev] && !g_persist.quiet[dev] && ++g_persist.run[dev] >= kQuietAfter) {
      g_persist.quiet[dev] = true;
      g_persist.ev_valid[dev] = false;
    }
    g_persist.last[dev] = s;
    g_persist.used[dev] = true;
    active = true;
  }
  // The section's LAST resident launch can carry the chain's event itself (its completion signal) instead of a marker recorded behind
  // it: between two kernels of one stream a marker is a gap of about 7 us (rocprofv3, the session loop: level 0 -> index map).  Returns
  // null when no event is owed (one stream only so far) - then nothing changes.
  bool closed = false;
  hipEvent_t closing_event() {
    if (!active || !g_persist.chained[dev] || g_persist.quiet[dev] || !ext_events()) return nullptr;
    if (!g_persist.ev[dev] && hipEventCreateWithFlags(&g_persist.ev[dev], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_persist.ev[dev] = nullptr;
      return nullptr;
    }
    closed = true;
    g_persist.ev_valid[dev] = true;
    return g_persist.ev[dev];
  }
  static bool ext_events() {
    static const bool on = [] {
      const char* e = getenv("DMS_PERSIST_EXT_EVENT");
      return !e || atoi(e) != 0;
    }();
    return on;
  }
  ~PersistSection() {
    if (!active) return;
    if (closed) {
      g_persist.mu.unlock();
      return;
    }
    if (g_persist.chained[dev] && !g_persist.quiet[dev]) {
      if (!g_persist.ev[dev]) (void)hipEventCreateWithFlags(&g_persist.ev[dev], hipEventDisableTiming);
      g_persist.ev_valid[dev] = g_persist.ev[dev] && hipEventRecord(g_persist.ev[dev], s) == hipSuccess;
      if (!g_persist.ev_valid[dev]) (void)hipGetLastError();
    }
    g_persist.mu.unlock();
  }
};

// `done`: an event that fires with THIS launch's completion signal (hipExtLaunchKernelGGL) - no marker packet behind the kernel
template <bool ICP, bool RGB, int P, bool EXIT>
static void launch_gn_level_p(int nb, hipStream_t s, TrackState* st, const GnArgs& a, const LevelArgs& L, hipEvent_t done) {
  if (done)
    hipExtLaunchKernelGGL((k_gn_level<ICP, RGB, P, EXIT>), dim3(nb), dim3(kPB), 0, s, nullptr, done, 0, st, a, L);
  else
    hipLaunchKernelGGL((k_gn_level<ICP, RGB, P, EXIT>), dim3(nb), dim3(kPB), 0, s, st, a, L);
}
template <bool ICP, bool RGB, bool EXIT>
static void launch_gn_level_f(int P, int nb, hipStream_t s, TrackState* st, const GnArgs& a, const LevelArgs& L, hipEvent_t done) {
  if (P == 1)
    launch_gn_level_p<ICP, RGB, 1, EXIT>(nb, s, st, a, L, done);
  else if (P == 2)
    launch_gn_level_p<ICP, RGB, 2, EXIT>(nb, s, st, a, L, done);
  else if (P == 3)
    launch_gn_level_p<ICP, RGB, 3, EXIT>(nb, s, st, a, L, done);
  else if (P == 4)
    launch_gn_level_p<ICP, RGB, 4, EXIT>(nb, s, st, a, L, done);
  else
    launch_gn_level_p<ICP, RGB, 5, EXIT>(nb, s, st, a, L, done);
}
template <bool ICP, bool RGB>
static void launch_gn_level(int P, int nb, hipStream_t s, TrackState* st, const GnArgs& a, const LevelArgs& L, hipEvent_t done = nullptr) {
  if (L.early_exit)
    launch_gn_level_f<ICP, RGB, true>(P, nb, s, st, a, L, done);
  else
    launch_gn_level_f<ICP, RGB, false>(P, nb, s, st, a, L, done);
}

template <bool ICP, bool RGB>
static void launch_track_coarse(int P1, bool early_exit, int grid, hipStream_t s, TrackState* st, const CoarseArgs& F, const So3Extra& ex) {
  if (P1 == 1) {
    if (early_exit)
      hipLaunchKernelGGL((k_track_coarse<ICP, RGB, 1, true>), dim3(grid), dim3(kPB), 0, s, st, F, ex);
    else
      hipLaunchKernelGGL((k_track_coarse<ICP, RGB, 1, false>), dim3(grid), dim3(kPB), 0, s, st, F, ex);
  } else {
    if (early_exit)
      hipLaunchKernelGGL((k_track_coarse<ICP, RGB, 2, true>), dim3(grid), dim3(kPB), 0, s, st, F, ex);
    else
      hipLaunchKernelGGL((k_track_coarse<ICP, RGB, 2, false>), dim3(grid), dim3(kPB), 0, s, st, F, ex);
  }
}

static inline int budget_of(const dms_odometry* o) {
  return (o->budget_cap > 0 && o->budget_cap < o->max_resident_blocks) ? o->budget_cap : o->max_resident_blocks;
}

// grid of the resident SO3 kernel for this handle; 0 = the SO3 stage runs launch-per-phase
static int so3_resident_blocks(const dms_odometry* o) {
  const Buf& li = o->lastNextImage[2];
  const int nbp = (li.rows * li.cols + kPB - 1) / kPB;
  return (o->resident && nbp <= kMaxPersistBlocks && nbp <= budget_of(o)) ? nbp : 0;
}

int odometry_track_enqueue(dms_odometry* o, const float* trans, const float* rot, const float* prior_pose16_dev, int rgbOnly,
                           float icpWeight, int pyramid, int fastOdom, int so3, int interMap, hipStream_t s, FrameState* frame,
                           float weightMultiplier) {
  DMS_REQUIRE(o && ((trans && rot) || prior_pose16_dev), "null argument");
  const bool icp = !rgbOnly && icpWeight > 0;
  const bool rgb = rgbOnly || icpWeight < 100;
  DMS_REQUIRE(icp || rgb, "neither ICP nor RGB term active");
  int rc;

  Prior prior;
  memset(&prior, 0, sizeof(prior));
  if (!prior_pose16_dev) {
    memcpy(prior.v, trans, 3 * sizeof(float));
    memcpy(prior.v + 3, rot, 9 * sizeof(float));
  }

  if (rgb && o->deriv_of != o->nextImage[0].p) {
    if ((rc = nextDerivatives(o, s))) return rc;
  }

  int iterations[DMS_NUM_PYRS];
  iterations[0] = interMap ? 50 : fastOdom ? 3 : 10;
  iterations[1] = interMap ? 50 : pyramid ? 5 : 0;
  iterations[2] = interMap ? 50 : pyramid ? 4 : 0;
  int first_level = 0;
  for (int l = DMS_NUM_PYRS - 1; l >= 0; --l)
    if (iterations[l] > 0) {
      first_level = l;
      break;
    }

  // set-up already done inside the model pyramid kernel (frame step): nothing to launch here; the SO3 launch below carries the
  // deferred pyramid step and re-arms the dense counters
  const bool folded = o->init_folded;
  So3Extra ex;
  memset(&ex, 0, sizeof(ex));
  const bool so3_ran = o->so3_in_model;  // the SO3 stage of this call ran inside the model pyramid launch (k_so3_model)
  unsigned* first_gn_zero16 = nullptr;
  if (so3_ran) {
    o->so3_in_model = false;
    o->early_init = false;
    DMS_REQUIRE(so3 && o->folded_so3 == 1 && o->folded_first_level == first_level && o->folded_prior == prior_pose16_dev && prior_pose16_dev && !o->deferred_pyr,
                "track call differs from the one its early set-up and SO3 stage were prepared for");
    first_gn_zero16 = o->dense_cnt_zero;
    o->dense_cnt_zero = nullptr;
  } else if (folded) {
    o->init_folded = false;
    DMS_REQUIRE(so3 && so3_resident_blocks(o) > 0 && o->folded_so3 == (so3 ? 1 : 0) && o->folded_first_level == first_level &&
                    o->folded_prior == prior_pose16_dev && prior_pose16_dev,
                "track call differs from the one its folded set-up was prepared for");
    ex.zero16 = o->dense_cnt_zero;
    o->dense_cnt_zero = nullptr;
    if (o->deferred_pyr) {
      o->deferred_pyr = false;
      dms_image2d d1 = o->lastDepth[1].img(), d2 = o->lastDepth[2].img(), i1 = o->lastImage[1].img(), i2 = o->lastImage[2].img();
      ex.gx = (d2.cols + 63) / 64;
      ex.gy = (d2.rows + kPB / 64 - 1) / (kPB / 64);
      ex.dsrc = view<const float>(&d1);
      ex.ddst = view<float>(&d2);
      ex.isrc = view<const unsigned char>(&i1);
      ex.idst = view<unsigned char>(&i2);
    }
  } else {
    Timer t(o, s, "track_init");
    // all-reduce words of the call's resident kernels (resident mode only)
    const int zero_pairs = o->resident ? (int)((size_t)kArSets * kArWords / 2) : 0;
    const int ni = o->resident ? 16 : 1;
    unsigned* zero16 = o->dense_cnt_zero;
    o->dense_cnt_zero = nullptr;
    if (o->deferred_pyr) {
      o->deferred_pyr = false;
      dms_image2d d1 = o->lastDepth[1].img(), d2 = o->lastDepth[2].img(), i1 = o->lastImage[1].img(), i2 = o->lastImage[2].img();
      const int gx = (d2.cols + 63) / 64, gy = (d2.rows + 3) / 4;
      hipLaunchKernelGGL(k_track_init_pyr, dim3(gx * gy + ni), dim3(64, 4), 0, s, o->state, prior, prior_pose16_dev, o->fx, o->fy, o->cx, o->cy,
                         so3 ? 1 : 0, first_level, o->ar, zero_pairs, o->inject_timeouts > 0 ? 1 : 0, zero16, gx, gy, ni, view<const float>(&d1),
                         view<float>(&d2), view<const unsigned char>(&i1), view<unsigned char>(&i2));
    } else {
      hipLaunchKernelGGL(k_track_init, dim3(ni), dim3(256), 0, s, o->state, prior, prior_pose16_dev, o->fx, o->fy, o->cx, o->cy, so3 ? 1 : 0,
                         first_level, o->ar, zero_pairs, o->inject_timeouts > 0 ? 1 : 0, zero16);
    }
    DMS_CHECK_LAUNCH();
    if (o->inject_timeouts > 0) o->inject_timeouts -= 1;
  }

  PersistSection persist(s, budget_of(o), o->unchained_ok);

  auto level_below_of = [&](int l) {
    for (int q = l - 1; q >= 0; --q)
      if (iterations[q] > 0) return q;
    return l;
  };
  auto gn_args_of = [&](int l) {
    const int div = 1 << l;
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.maps.vcurr = (const float*)o->vmaps_curr[l].p;
    a.maps.vcurr_pitch = o->vmaps_curr[l].pitch;
    a.maps.ncurr = (const float*)o->nmaps_curr[l].p;
    a.maps.ncurr_pitch = o->nmaps_curr[l].pitch;
    a.maps.vprev = (const float*)o->vmaps_g_prev[l].p;
    a.maps.vprev_pitch = o->vmaps_g_prev[l].pitch;
    a.maps.nprev = (const float*)o->nmaps_g_prev[l].p;
    a.maps.nprev_pitch = o->nmaps_g_prev[l].pitch;
    a.fx = o->fx / div;
    a.fy = o->fy / div;
    a.cx = o->cx / div;
    a.cy = o->cy / div;
    a.distThres = o->distThres;
    a.angleThres = o->angleThres;
    a.dist2Le = sqrt_le_bound(o->distThres);
    a.sine2Le = sqrt_lt_bound(o->angleThres);
    a.rgb.dIdx = (const short*)o->nextdIdx[l].p;
    a.rgb.dIdy = (const short*)o->nextdIdy[l].p;
    a.rgb.dI_pitch = o->nextdIdx[l].pitch;
    a.rgb.lastDepth = (const float*)o->lastDepth[l].p;
    a.rgb.lastDepth_pitch = o->lastDepth[l].pitch;
    a.rgb.nextDepth = (const float*)o->nextDepth[l].p;
    a.rgb.nextDepth_pitch = o->nextDepth[l].pitch;
    a.rgb.lastImage = (const unsigned char*)o->lastImage[l].p;
    a.rgb.lastImage_pitch = o->lastImage[l].pitch;
    a.rgb.nextImage = (const unsigned char*)o->nextImage[l].p;
    a.rgb.nextImage_pitch = o->nextImage[l].pitch;
    a.rgb.gate = (const unsigned char*)o->nextGate[l].p;
    a.rgb.gate_pitch = o->nextGate[l].pitch;
    a.corres = (Corr8*)o->corresImg[l].p;
    a.cloud = (const float*)o->pointClouds[l].p;
    a.cloud_pitch = o->pointClouds[l].pitch;
    // pow(minimumGradientMagnitudes[i], 2.0) / pow(sobelScale, 2.0) (RGBDOdometry.cpp:445)
    a.minScale = rgb_min_scale(o, l);
    a.maxDepthDelta = o->maxDepthDeltaRGB;
    a.sobelScale = o->sobelScale;
    a.cols = o->vmaps_curr[l].cols;
    a.rows = o->vmaps_curr[l].rows / 3;
    a.level = l;
    a.rgbOnly = rgbOnly ? 1 : 0;
    a.exp_bias = o->exp_bias + o->depth_bias;
    return a;
  };
  auto level_args_of = [&](int l, int pnb) {
    LevelArgs L;
    L.n_iter = iterations[l];
    L.level = l;
    L.level_below = level_below_of(l);
    L.rgbOnly = rgbOnly ? 1 : 0;
    L.icpWeight = icpWeight;
    L.fx = o->fx;
    L.fy = o->fy;
    L.cx = o->cx;
    L.cy = o->cy;
    L.ar = o->ar + (size_t)(1 + l) * kArSetsPerKernel * kArWords;
    L.first_delay = o->first_delay_for(pnb, l);
    L.zero16 = nullptr;
    L.prof = (o->profiling && !o->profiling_level0_only) ? o->prof : nullptr;
    // on for trackers that ask for it (the frame step's model-to-model pass) unless forced either way
    L.early_exit = o->early_exit_force >= 0 ? o->early_exit_force : (o->early_exit ? 1 : 0);
    L.finalize = (l == 0) ? 1 : 0;  // level 0 always runs last
    L.fin_rgb = rgb ? 1 : 0;
    L.pose16_out = frame ? const_cast<float*>(prior_pose16_dev) : nullptr;
    L.frame = frame;
    L.weightMultiplier = weightMultiplier;
    return L;
  };
  auto level_shape = [&](int l, int& pP, int& pnb) {
    pP = 1;
    pnb = 0;
    if (o->resident && (iterations[l] <= kArRing || o->long_levels_resident))  // (more iterations than ring sets: the ring is re-armed in flight)
      persistent_shape(o->vmaps_curr[l].cols * (o->vmaps_curr[l].rows / 3), o->persist_target, budget_of(o), pP, pnb);
  };

  // SO3 + level 2 + level 1 in one resident launch (k_track_coarse) when all three are resident stages of this call and their shapes
  // fit one grid: level 1's, with one pixel per thread at level 2
  bool coarse = false;
  int cP1 = 1, cnb = 0;
  if (o->fuse_coarse && o->resident && so3 && !so3_ran && iterations[2] > 0 && iterations[1] > 0 && iterations[2] <= kArRing && iterations[1] <= kArRing) {
    int P2 = 1, nb2 = 0;
    level_shape(2, P2, nb2);
    level_shape(1, cP1, cnb);
    const int nbs = so3_resident_blocks(o);
    coarse = nbs > 0 && nb2 > 0 && cnb > 0 && P2 == 1 && cP1 <= 2 && nb2 <= cnb && nbs <= cnb;
  }
  if (coarse) {
    const Buf& li = o->lastNextImage[2];
    const Buf& ni = o->nextImage[2];
    CoarseArgs F;
    memset(&F, 0, sizeof(F));
    F.so3 = {(const unsigned char*)li.p, li.pitch, (const unsigned char*)ni.p, ni.pitch, ni.cols, ni.rows, o->ar, SolveCam{o->fx, o->fy, o->cx, o->cy},
             first_level, 10, o->exp_bias + o->depth_bias, o->first_delay_for(cnb, 3)};
    F.a2 = gn_args_of(2);
    F.a1 = gn_args_of(1);
    F.L2 = level_args_of(2, cnb);
    F.L1 = level_args_of(1, cnb);
    F.run_so3 = F.run_l2 = F.run_l1 = 1;
    F.rider_cnt = o->rider_cnt;
    o->rider_issued += (unsigned)(ex.gx * ex.gy);
    F.rider_target = o->rider_issued;
    ex.nb_so3 = cnb;
    persist.begin();
    Timer t(o, s, "track_coarse");
    const bool ee = F.L1.early_exit != 0;
    if (icp && rgb)
      launch_track_coarse<true, true>(cP1, ee, cnb + ex.gx * ex.gy, s, o->state, F, ex);
    else if (icp)
      launch_track_coarse<true, false>(cP1, ee, cnb + ex.gx * ex.gy, s, o->state, F, ex);
    else
      launch_track_coarse<false, true>(cP1, ee, cnb + ex.gx * ex.gy, s, o->state, F, ex);
    DMS_CHECK_LAUNCH();
  } else if (so3 && !so3_ran) {
    const int L = 2;
    const Buf& li = o->lastNextImage[L];
    const Buf& ni = o->nextImage[L];
    const int nb = reduce_blocks_for(li.rows * li.cols);
    const int nbp = so3_resident_blocks(o);
    if (nbp > 0) {
      persist.begin();
      Timer t(o, s, "so3_level");
      SolveCam cam = {o->fx, o->fy, o->cx, o->cy};
      ex.nb_so3 = nbp;
      hipLaunchKernelGGL(k_so3_level, dim3(nbp + ex.gx * ex.gy), dim3(kPB), 0, s, o->state, (const unsigned char*)li.p, li.pitch,
                         (const unsigned char*)ni.p, ni.pitch, ni.cols, ni.rows, o->ar, cam, first_level, 10, o->exp_bias + o->depth_bias, o->first_delay_for(nbp, 3), ex);
      DMS_CHECK_LAUNCH();
    } else
    for (int i = 0; i < 10; ++i) {
      {
        Timer t(o, s, "so3_pass");
        SolveCam cam = {o->fx, o->fy, o->cx, o->cy};
        hipLaunchKernelGGL(k_so3_pass, dim3(nb), dim3(kBlock), 0, s, o->state, (const unsigned char*)li.p, li.pitch,
                           (const unsigned char*)ni.p, ni.pitch, ni.cols, ni.rows, o->part_so3, o->tickets, cam, i, i == 9 ? 1 : 0, first_level,
                           o->exp_bias + o->depth_bias);
        DMS_CHECK_LAUNCH();
      }
    }
  }

  bool finalized_in_kernel = false;
  for (int l = DMS_NUM_PYRS - 1; l >= 0; --l) {
    if (coarse && l >= 1) continue;  // (ran inside k_track_coarse; resident levels rebuild the cloud point themselves)
    int pP = 1, pnb = 0;
    level_shape(l, pP, pnb);
    const bool persistent = pnb > 0;
    if (rgb && !persistent) {  // the persistent kernel rebuilds the cloud point from lastDepth itself
      dms_camera k = {o->fx, o->fy, o->cx, o->cy};
      dms_image2d d = o->lastDepth[l].img(), c = o->pointClouds[l].img();
      if ((rc = projectToPointCloud(&d, &c, &k, l, s))) return rc;
    }
    if (iterations[l] == 0) continue;
    const GnArgs a = gn_args_of(l);
    const int nb = track_blocks_for(a.cols * a.rows);
    const int level_below = level_below_of(l);
    if (persistent) {
      LevelArgs L = level_args_of(l, pnb);
      if (first_gn_zero16) {
        L.zero16 = first_gn_zero16;
        first_gn_zero16 = nullptr;
      }
      if (l == 0) finalized_in_kernel = true;
      persist.begin();
      static const char* const kLevelTimer[3] = {"gn_level0", "gn_level1", "gn_level2"};
      Timer t(o, s, kLevelTimer[l]);
      hipEvent_t done = (l == 0) ? persist.closing_event() : nullptr;  // level 0 is the section's last resident launch
      if (icp && rgb)
        launch_gn_level<true, true>(pP, pnb, s, o->state, a, L, done);
      else if (icp)
        launch_gn_level<true, false>(pP, pnb, s, o->state, a, L, done);
      else
        launch_gn_level<false, true>(pP, pnb, s, o->state, a, L, done);
      DMS_CHECK_LAUNCH();
      continue;
    }
    for (int j = 0; j < iterations[l]; ++j) {
      const int next_level = (j == iterations[l] - 1) ? level_below : l;
      SolveArgs q;
      q.icp = icp ? 1 : 0;
      q.rgb = rgb ? 1 : 0;
      q.rgbOnly = rgbOnly ? 1 : 0;
      q.icpWeight = icpWeight;
      q.level = l;
      q.first_iter = j == 0 ? 1 : 0;
      q.next_level = next_level;
      q.level_below = level_below;
      q.fx = o->fx;
      q.fy = o->fy;
      q.cx = o->cx;
      q.cy = o->cy;
      {
        Timer t(o, s, "gn_pass1");
        if (icp && rgb)
          hipLaunchKernelGGL((k_gn_pass1<true, true>), dim3(nb), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_cnt, kMaxPartialBlocks, j == 0 ? 1 : 0);
        else if (icp)
          hipLaunchKernelGGL((k_gn_pass1<true, false>), dim3(nb), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_cnt, kMaxPartialBlocks, j == 0 ? 1 : 0);
        else
          hipLaunchKernelGGL((k_gn_pass1<false, true>), dim3(nb), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_cnt, kMaxPartialBlocks, j == 0 ? 1 : 0);
        DMS_CHECK_LAUNCH();
      }
      if (rgb) {
        Timer t(o, s, "gn_pass2");
        hipLaunchKernelGGL(k_gn_pass2, dim3(nb), dim3(kBlock), 0, s, o->state, a, o->part_cnt, nb, kMaxPartialBlocks, j == 0 ? 1 : 0, o->part_rgb);
        DMS_CHECK_LAUNCH();
      }
      {
        Timer t(o, s, "gn_solve");
        if (icp && rgb)
          hipLaunchKernelGGL((k_gn_solve<true, true>), dim3(1), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_rgb, o->part_cnt, kMaxPartialBlocks, nb, q);
        else if (icp)
          hipLaunchKernelGGL((k_gn_solve<true, false>), dim3(1), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_rgb, o->part_cnt, kMaxPartialBlocks, nb, q);
        else
          hipLaunchKernelGGL((k_gn_solve<false, true>), dim3(1), dim3(kBlock), 0, s, o->state, a, o->part_icp, o->part_rgb, o->part_cnt, kMaxPartialBlocks, nb, q);
        DMS_CHECK_LAUNCH();
      }
    }
  }

  if (first_gn_zero16) {  // (no resident level took it: launch-per-phase levels)
    DMS_HIP(hipMemsetAsync(first_gn_zero16, 0, 16 * 16 * sizeof(unsigned), s));  // (the 16 counters, 64 bytes apart)
  }
  if (!finalized_in_kernel) {
    Timer t(o, s, "track_finalize");
    // frame step (frame state given): the result is written back into the pose block the prior came from;
    // a device prior without frame state (model-to-model tracking) is read-only
    hipLaunchKernelGGL(k_track_finalize, dim3(1), dim3(64), 0, s, o->state, rgb ? 1 : 0,
                       frame ? const_cast<float*>(prior_pose16_dev) : nullptr, frame, weightMultiplier);
    DMS_CHECK_LAUNCH();
  }
  // (the result block is copied to the host by dms_odometry_fetch_result, not once per call)

  if (so3) {  // RGBDOdometry.cpp:595-601 — stream-ordered, so swapping the handles is enough
    for (int i = 0; i < DMS_NUM_PYRS; i++) std::swap(o->lastNextImage[i], o->nextImage[i]);
  }
  return DMS_OK;
}

// ---- device-resident variants used by the frame step (fusion_frame.hip) --------------------
__global__ void k_track_pose_out(const TrackState* __restrict__ st, float* __restrict__ pose16) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pose16[i * 4 + j] = st->out_rot[i * 3 + j];
    pose16[i * 4 + 3] = st->out_trans[i];
  }
  pose16[12] = 0.f;
  pose16[13] = 0.f;
  pose16[14] = 0.f;
  pose16[15] = 1.f;
}

int odometry_result_pose(dms_odometry* o, float* pose16_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_track_pose_out, dim3(1), dim3(64), 0, s, o->state, pose16_dev);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

// ---- local loop-closure candidate (ElasticFusion.cpp:427-474) --------------------------------
// One block, after the model-to-model tracker: thread 0 takes the acceptance decision from the
// tracker's side outputs (getCovariance = inverse of lastA by partially pivoted elimination in fp64,
// RGBDOdometry.cpp:607-610; thresholds of ElasticFusion.cpp:428-442), then the block samples the
// W/20 x H/20 nearest-neighbour grid of the ACTIVE vertex map and the INACTIVE time map
// (Resize::vertex / Resize::time, :443-444) and compacts the surface constraints in the reference's
// order (columns outer, rows inner, :446-447).
__global__ __launch_bounds__(256) void k_loop_candidate(const TrackState* __restrict__ st, FrameState* __restrict__ frame,
                                                        const float4* __restrict__ vertex, const unsigned short* __restrict__ oldTime,
                                                        int cols, int rows, float maxDepth, LoopState* __restrict__ out,
                                                        float* __restrict__ cons) {
  __shared__ int s_ok;
  __shared__ int s_wave[4];
  __shared__ int s_base;
  // Gauss-Jordan on the augmented 6x12 matrix in LDS, one thread per element: the same operations
  // in the same order per element as the scalar loop of dms_odometry_getCovariance
  __shared__ double s_m[6][12];
  const int er = threadIdx.x / 12, ej = threadIdx.x % 12;
  const bool elem = threadIdx.x < 72;
  if (elem) s_m[er][ej] = ej < 6 ? st->lastA[er * 6 + ej] : ((ej - 6 == er) ? 1.0 : 0.0);
  __syncthreads();
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(s_m[r][c]) > fabs(s_m[p][c])) p = r;
    __syncthreads();
    if (p != c && threadIdx.x < 12) {
      const double t0 = s_m[c][threadIdx.x];
      s_m[c][threadIdx.x] = s_m[p][threadIdx.x];
      s_m[p][threadIdx.x] = t0;
    }
    __syncthreads();
    const double d = s_m[c][c];
    __syncthreads();
    if (threadIdx.x < 12) s_m[c][threadIdx.x] /= d;
    __syncthreads();
    const double f = elem ? s_m[er][c] : 0.0;
    __syncthreads();
    if (elem && er != c) s_m[er][ej] -= f * s_m[c][ej];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    bool covOk = true;
    for (int i = 0; i < 6; ++i) {
      const double cii = s_m[i][6 + i];
      out->cov_diag[i] = cii;
      if (cii > 8e-05) covOk = false;
    }
    const bool timed_out = st->sync_timeout != 0;  // model-to-model pass invalid: no candidate, counted like a tracker timeout
    if (st->sync_timeout == 2)
      frame->track_range_failures += 1;
    else if (timed_out)
      frame->track_timeouts += 1;
    const int ok = (!timed_out && covOk && st->lastICPCount > 15000.f && st->lastICPError < 0.0003f) ? 1 : 0;
    out->ok = ok;
    out->icp_error = st->lastICPError;
    out->icp_count = st->lastICPCount;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) out->est_pose[i * 4 + j] = st->out_rot[i * 3 + j];
      out->est_pose[i * 4 + 3] = st->out_trans[i];
    }
    out->est_pose[12] = out->est_pose[13] = out->est_pose[14] = 0.f;
    out->est_pose[15] = 1.f;
    s_ok = ok;
    s_base = 0;
  }
  __syncthreads();
  if (!s_ok) {
    if (threadIdx.x == 0) out->n_constraints = 0;
    return;
  }
  const int dw = cols / 20, dh = rows / 20, n = dw * dh;
  const float* P = frame->cur.pose;
  for (int base = 0; base < n; base += 256) {
    const int k = base + (int)threadIdx.x;
    bool valid = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned short t = 0;
    if (k < n) {
      const int i = k / dh, j = k - i * dh;
      const float u = ((float)i + 0.5f) / (float)dw, v = ((float)j + 0.5f) / (float)dh;
      const int sx = texel(u, (float)cols, cols), sy = texel(v, (float)rows, rows);
      p = vertex[(size_t)sy * cols + sx];
      t = oldTime[(size_t)sy * cols + sx];
      valid = p.z > 0.f && p.z < maxDepth && t > 0;
    }
    const unsigned long long m = __ballot(valid);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_wave[w] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int q = 0; q < w; ++q) off += s_wave[q];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (valid) {
      float* c = cons + (size_t)off * 8;
      for (int r = 0; r < 3; ++r) {
        c[r] = P[r * 4 + 0] * p.x + P[r * 4 + 1] * p.y + P[r * 4 + 2] * p.z + P[r * 4 + 3];
        c[3 + r] = st->out_rot[r * 3 + 0] * p.x + st->out_rot[r * 3 + 1] * p.y + st->out_rot[r * 3 + 2] * p.z + st->out_trans[r];
      }
      c[6] = (float)t;
      c[7] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) out->n_constraints = s_base;
}

int odometry_loop_candidate(dms_odometry* o, FrameState* frame, const dms_image2d* vertex, const dms_image2d* oldTime, float maxDepth,
                            LoopState* out, float* cons, hipStream_t s) {
  hipLaunchKernelGGL(k_loop_candidate, dim3(1), dim3(256), 0, s, o->state, frame, (const float4*)vertex->data,
                     (const unsigned short*)oldTime->data, vertex->cols, vertex->rows, maxDepth, out, cons);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int selectCopy16(void* dst, const void* a, const void* b, const int* flag_dev, int force_b, size_t n16, hipStream_t s);
int transformMapsDev(dms_image2d* v, dms_image2d* n, const float* pose16_dev, hipStream_t s);

// ---- live-frame ring (see dms_odometry::LiveSet) ----
void odometry_set_early_exit(dms_odometry* o, int on) { o->early_exit = on != 0; }

// compute units the resident tracker kernels of this handle leave free at their widest level (-1: launch-per-phase, nothing
// is held): what an overlapping stream can use without keeping a resident block off the device
int odometry_free_cus(const dms_odometry* o) {
  if (!o->resident) return -1;
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
  int widest = so3_resident_blocks(o);
  for (int l = 0; l < DMS_NUM_PYRS; ++l) {
    int P = 1, nb = 0;
    persistent_shape(o->vmaps_curr[l].cols * (o->vmaps_curr[l].rows / 3), o->persist_target, o->max_resident_blocks, P, nb);
    widest = nb > widest ? nb : widest;
  }
  return cus - widest;
}

int odometry_enable_ring(dms_odometry* o) {
  if (o->ring_enabled) return DMS_OK;
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    o->ring[0].depth_tmp[i] = o->depth_tmp[i];
    o->ring[0].vmaps_curr[i] = o->vmaps_curr[i];
    o->ring[0].nmaps_curr[i] = o->nmaps_curr[i];
    o->ring[0].nextImage[i] = o->nextImage[i];
    o->ring[0].nextdIdx[i] = o->nextdIdx[i];
    o->ring[0].nextdIdy[i] = o->nextdIdy[i];
    o->ring[0].nextGate[i] = o->nextGate[i];
  }
  for (int pass = 0; pass < 2; ++pass) {
    Carver c;
    c.base = pass ? o->ring_arena : nullptr;
    const int W = o->width, H = o->height;
    auto mk = [&](Buf& b, int rows, int cols, size_t elem) {
      b.rows = rows;
      b.cols = cols;
      b.pitch = (size_t)cols * elem;
      b.p = c.take(b.pitch * rows);
    };
    for (int k = 1; k < 3; ++k)
      for (int i = 0; i < DMS_NUM_PYRS; ++i) {
        const int r = H >> i, w = W >> i;
        mk(o->ring[k].depth_tmp[i], r, w, 2);
        mk(o->ring[k].vmaps_curr[i], r * 3, w, 4);
        mk(o->ring[k].nmaps_curr[i], r * 3, w, 4);
        mk(o->ring[k].nextImage[i], r, w, 1);
        mk(o->ring[k].nextdIdx[i], r, w, 2);
        mk(o->ring[k].nextdIdy[i], r, w, 2);
        mk(o->ring[k].nextGate[i], r, w, 1);
      }
    if (!pass) {
      const size_t bytes = align_up(c.off, 256);
      hipError_t e = hipMalloc((void**)&o->ring_arena, bytes);
      if (e == hipSuccess) e = hipMemset(o->ring_arena, 0, bytes);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc(live ring)", __FILE__, __LINE__);
    }
  }
  o->ring_enabled = true;
  return DMS_OK;
}

// point the tracker's live-frame buffers at ring set k (host-side handle swap, stream-ordered use)
void odometry_bind_live(dms_odometry* o, int k) {
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    o->depth_tmp[i] = o->ring[k].depth_tmp[i];
    o->vmaps_curr[i] = o->ring[k].vmaps_curr[i];
    o->nmaps_curr[i] = o->ring[k].nmaps_curr[i];
    o->nextImage[i] = o->ring[k].nextImage[i];
    o->nextdIdx[i] = o->ring[k].nextdIdx[i];
    o->nextdIdy[i] = o->ring[k].nextdIdy[i];
    o->nextGate[i] = o->ring[k].nextGate[i];
  }
  o->deriv_of = o->nextImage[0].p;  // every ring set carries the derivatives of its own image
}

void odometry_bind_lastnext(dms_odometry* o, int k) {
  for (int i = 0; i < DMS_NUM_PYRS; ++i) o->lastNextImage[i] = o->ring[k].nextImage[i];
}

// The bound live set's three pyramid levels, for the frame step's fused live half (prep.hip liveLevelsFused): it writes them
// all itself and the derivative pyramid with them (`mark_derivatives`: the bookkeeping of nextDerivatives)
int odometry_live_views(dms_odometry* o, dms_image2d* depth, dms_image2d* vmap, dms_image2d* nmap, dms_image2d* image, dms_image2d* dx,
                        dms_image2d* dy, dms_image2d* gate, float* minScale, bool mark_derivatives) {
  for (int l = 0; l < DMS_NUM_PYRS; ++l) {
    depth[l] = o->depth_tmp[l].img();
    vmap[l] = o->vmaps_curr[l].img();
    nmap[l] = o->nmaps_curr[l].img();
    image[l] = o->nextImage[l].img();
    dx[l] = o->nextdIdx[l].img();
    dy[l] = o->nextdIdy[l].img();
    gate[l] = o->nextGate[l].img();
    minScale[l] = rgb_min_scale(o, l);
  }
  if (mark_derivatives) o->deriv_of = o->nextImage[0].p;
  return DMS_OK;
}

// live half of initRGB: intensity pyramid + derivative pyramid of the bound live set
int odometry_initRGB_image(dms_odometry* o, const dms_image2d* rgba, hipStream_t s) {
  int rc = populateImage(rgba, o->nextImage, s);
  if (rc) return rc;
  return nextDerivatives(o, s);
}

// model half of initRGB.  populateRGBDData derives nextDepth from the same vmaps_tmp (and the same
// cutoff) as initRGBModel derived lastDepth a moment earlier, so the two pyramids are identical:
// the handles are aliased instead of recomputing three kernels.
void odometry_alias_next_depth(dms_odometry* o) {
  for (int i = 0; i < DMS_NUM_PYRS; ++i) o->nextDepth[i] = o->lastDepth[i];
}

// views of the live intensity and depth pyramids at `level` (RGBDOdometry::nextImg / nextD, RGBDOdometry.h:79-87)
int odometry_next_buffers(dms_odometry* o, int level, dms_image2d* nextImage, dms_image2d* nextDepth) {
  DMS_REQUIRE(o && level >= 0 && level < DMS_NUM_PYRS, "bad argument");
  if (nextImage) *nextImage = o->nextImage[level].img();
  if (nextDepth) *nextDepth = o->nextDepth[level].img();
  return DMS_OK;
}

// The set-up of the frame step's NEXT tracker call as a block group of an EARLIER kernel of the frame (the tracking prediction's
// resolve pass, fusion_map.hip): fills `ti` and returns 1 when this handle can take it (resident SO3 stage, the call odometry_initModel_fused
// + odometry_track_enqueue are about to make), 0 otherwise (ti->blocks = 0: the set-up stays folded into the model pyramid kernel).
// With the set-up done a kernel boundary ahead, odometry_initModel_fused runs the SO3 stage beside the model pyramid (k_so3_model).
int odometry_early_init_args(dms_odometry* o, const TrackFold* fold, TrackInitArgs* ti) {
  if (!o) return 0;
  if (!fold && o->early_init && ti->inject_timeout) o->inject_timeouts += 1;  // (cancelled: the injected fault goes to the call's real set-up)
  memset(ti, 0, sizeof(*ti));
  o->early_init = false;
  if (!fold || !o->so3_beside_model || !o->resident || !fold->prior_pose16 || !fold->so3 || fold->interMap || so3_resident_blocks(o) <= 0) return 0;
  const int it1 = fold->pyramid ? 5 : 0, it2 = fold->pyramid ? 4 : 0;
  ti->st = o->state;
  ti->prior_pose16 = fold->prior_pose16;
  ti->fx = o->fx;
  ti->fy = o->fy;
  ti->cx = o->cx;
  ti->cy = o->cy;
  ti->so3 = 1;
  ti->first_level = it2 > 0 ? 2 : (it1 > 0 ? 1 : 0);
  ti->sync_words = o->ar;
  ti->n_sync = (int)((size_t)kArSets * kArWords / 2);
  ti->inject_timeout = o->inject_timeouts > 0 ? 1 : 0;
  if (o->inject_timeouts > 0) o->inject_timeouts -= 1;
  ti->blocks = 16;
  o->early_init = true;
  o->folded_so3 = 1;
  o->folded_first_level = ti->first_level;
  o->folded_prior = fold->prior_pose16;
  return 1;
}

// initICPModel + initRGBModel of the frame step in four launches (prep.hip, modelPyramidFused).
// The operator-layer staging copy vmaps_tmp is not written on this path; nextDepth is aliased to
// lastDepth by the caller, so nothing reads it.
// defer_last_step: the pyramid's last step is left to the first kernel of the NEXT odometry_track_enqueue on this object
// (k_track_init_pyr), which the caller must enqueue on the same stream before anything else reads level 2 of lastDepth / lastImage
// fold: the set-up of the NEXT odometry_track_enqueue on this object (prior = the pose block fold->prior_pose16, the given
// so3 / pyramid / fastOdom switches) runs as a block group of the pyramid kernel; that call must follow on the same stream
// with the same arguments.  Taken only where the call's first launch is the resident SO3 level (which then carries the
// deferred pyramid step); otherwise ignored.
int odometry_initModel_fused(dms_odometry* o, const void* vA, const void* nA, const void* iA, const void* vB, const void* nB,
                             const void* iB, const int* flag_dev, int force_b_img, const float* pose16_dev, hipStream_t s,
                             int defer_last_step, unsigned* dense_cnt, int dense_samples, const TrackFold* fold) {
  dms_image2d v[DMS_NUM_PYRS], n[DMS_NUM_PYRS], d[DMS_NUM_PYRS], im[DMS_NUM_PYRS];
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    v[i] = o->vmaps_g_prev[i].img();
    n[i] = o->nmaps_g_prev[i].img();
    d[i] = o->lastDepth[i].img();
    im[i] = o->lastImage[i].img();
  }
  o->deferred_pyr = defer_last_step != 0;
  // dense_cnt: the source choice comes from the 16 counters of the prediction's resolve pass instead of *flag_dev, which then
  // receives the decision; the next odometry_track_enqueue on this object zeroes the counters
  DMS_REQUIRE(!dense_cnt || defer_last_step, "dense counters are zeroed by the track call that takes the deferred pyramid step");
  o->dense_cnt_zero = dense_cnt;
  TrackInitArgs ti;
  memset(&ti, 0, sizeof(ti));
  o->init_folded = false;
  if (fold && fold->prior_pose16 && fold->so3 && !fold->interMap && defer_last_step && so3_resident_blocks(o) > 0) {
    const int it1 = fold->pyramid ? 5 : 0, it2 = fold->pyramid ? 4 : 0;
    ti.st = o->state;
    ti.prior_pose16 = fold->prior_pose16;
    ti.fx = o->fx;
    ti.fy = o->fy;
    ti.cx = o->cx;
    ti.cy = o->cy;
    ti.so3 = 1;
    ti.first_level = it2 > 0 ? 2 : (it1 > 0 ? 1 : 0);
    ti.sync_words = o->ar;
    ti.n_sync = (int)((size_t)kArSets * kArWords / 2);
    ti.inject_timeout = (!o->early_init && o->inject_timeouts > 0) ? 1 : 0;
    if (!o->early_init && o->inject_timeouts > 0) o->inject_timeouts -= 1;
    ti.blocks = 16;
    o->init_folded = true;
    o->folded_so3 = 1;
    o->folded_first_level = ti.first_level;
    o->folded_prior = fold->prior_pose16;
  }
  if (o->early_init && o->init_folded) {
    // The set-up ran a kernel ahead: the resident SO3 stage goes beside the model pyramid's groups in ONE launch (k_so3_model), the
    // pyramid's last step comes straight from the sources in that launch, nothing is deferred.
    o->init_folded = false;
    o->deferred_pyr = false;
    o->so3_in_model = true;
    ModelGroups G;
    memset(&G, 0, sizeof(G));
    G.m.vA = (const float4*)vA;
    G.m.nA = (const float4*)nA;
    G.m.iA = (const uchar4*)iA;
    G.m.vB = (const float4*)vB;
    G.m.nB = (const float4*)nB;
    G.m.iB = (const uchar4*)iB;
    G.m.flag = flag_dev;
    G.m.dense_cnt = dense_cnt;
    G.m.dense_samples = dense_samples;
    G.m.flag_out = const_cast<int*>(flag_dev);
    G.m.force_b_img = force_b_img;
    G.m.pose16 = pose16_dev;
    G.rows0 = v[0].rows / 3;
    G.cols0 = v[0].cols;
    G.rows1 = v[1].rows / 3;
    G.cols1 = v[1].cols;
    G.rows2 = v[2].rows / 3;
    G.cols2 = v[2].cols;
    DMS_REQUIRE(G.rows1 == G.rows0 / 2 && G.cols1 == G.cols0 / 2 && G.rows2 == G.rows1 / 2 && G.cols2 == G.cols1 / 2, "pyramid shapes");
    constexpr int BYv = kPB / 64;
    G.g0x = (G.cols0 + 63) / 64;
    G.g0y = (G.rows0 + BYv - 1) / BYv;
    G.g12x = ((G.cols1 + 1) / 2 + 15) / 16;
    G.g12y = ((G.rows1 + 1) / 2 + BYv - 1) / BYv;
    G.gsx = (d[1].cols + 63) / 64;
    G.gsy = (d[1].rows + BYv - 1) / BYv;
    G.g2x = (d[2].cols + 15) / 16;
    G.g2y = (d[2].rows + 7) / 8;
    G.cutOff = o->maxDepthRGB;
    G.v0 = view<float>(&v[0]);
    G.n0 = view<float>(&n[0]);
    G.depth0 = view<float>(&d[0]);
    G.inten0 = view<unsigned char>(&im[0]);
    G.v1 = view<float>(&v[1]);
    G.n1 = view<float>(&n[1]);
    G.v2 = view<float>(&v[2]);
    G.n2 = view<float>(&n[2]);
    G.depth1 = view<float>(&d[1]);
    G.inten1 = view<unsigned char>(&im[1]);
    G.depth2 = view<float>(&d[2]);
    G.inten2 = view<unsigned char>(&im[2]);
    const Buf& li = o->lastNextImage[2];
    const Buf& ni = o->nextImage[2];
    const int nbs = so3_resident_blocks(o);
    const So3Args q = {(const unsigned char*)li.p, li.pitch, (const unsigned char*)ni.p, ni.pitch, ni.cols, ni.rows, o->ar, SolveCam{o->fx, o->fy, o->cx, o->cy},
                       o->folded_first_level, 10, o->exp_bias + o->depth_bias, o->first_delay_for(nbs, 3)};
    PersistSection persist(s, budget_of(o), o->unchained_ok);
    persist.begin();
    Timer t(o, s, "so3_model");
    hipLaunchKernelGGL(k_so3_model, dim3(nbs + G.g12x * G.g12y + G.gsx * G.gsy + G.g2x * G.g2y + G.g0x * G.g0y), dim3(kPB), 0, s, o->state, q, nbs, G);
    DMS_CHECK_LAUNCH();
    return DMS_OK;
  }
  o->early_init = false;
  return modelPyramidFused(vA, nA, iA, vB, nB, iB, flag_dev, force_b_img, pose16_dev, v, n, d, im, o->maxDepthRGB, s, o->deferred_pyr, dense_cnt,
                           dense_samples, const_cast<int*>(flag_dev), o->init_folded ? &ti : nullptr);
}

// initICP(vertex map, normal map) + initRGB(image) of the live side in the fused form (RGBDOdometry.cpp:118-137,
// :162-191 through populateRGBDData): no transform, nextDepth from the same vertex map, then the Sobel / gate pyramid
int odometry_initLive_fused(dms_odometry* o, const void* verts, const void* norms, const void* rgba, const int* any_flag_dev,
                            hipStream_t s) {
  dms_image2d v[DMS_NUM_PYRS], n[DMS_NUM_PYRS], d[DMS_NUM_PYRS], im[DMS_NUM_PYRS];
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    v[i] = o->vmaps_curr[i].img();
    n[i] = o->nmaps_curr[i].img();
    d[i] = o->nextDepth[i].img();
    im[i] = o->nextImage[i].img();
  }
  int rc = modelPyramidFused(verts, norms, rgba, verts, norms, rgba, any_flag_dev, 0, nullptr, v, n, d, im, o->maxDepthRGB, s);
  if (rc) return rc;
  return nextDerivatives(o, s);
}

// initICPModel with the source chosen on device: (*flag ? fill-in maps : predicted maps), pose read from HBM
int odometry_initICPModel_sel(dms_odometry* o, const float* vA, const float* nA, const float* vB, const float* nB, const int* flag_dev,
                              const float* pose16_dev, hipStream_t s) {
  const size_t n16 = (size_t)o->width * o->height;
  int rc;
  if ((rc = selectCopy16(o->vmaps_tmp, vA, vB, flag_dev, 0, n16, s))) return rc;
  if ((rc = selectCopy16(o->nmaps_tmp, nA, nB, flag_dev, 0, n16, s))) return rc;
  dms_image2d v0 = o->vmaps_g_prev[0].img(), n0 = o->nmaps_g_prev[0].img();
  if ((rc = copyMaps(o->vmaps_tmp, o->nmaps_tmp, &v0, &n0, s))) return rc;
  for (int i = 1; i < DMS_NUM_PYRS; ++i) {
    dms_image2d va = o->vmaps_g_prev[i - 1].img(), vb = o->vmaps_g_prev[i].img();
    dms_image2d na = o->nmaps_g_prev[i - 1].img(), nb = o->nmaps_g_prev[i].img();
    if ((rc = resizeMap(&va, &vb, false, s))) return rc;
    if ((rc = resizeMap(&na, &nb, true, s))) return rc;
  }
  for (int i = 0; i < DMS_NUM_PYRS; ++i) {
    dms_image2d v = o->vmaps_g_prev[i].img(), n = o->nmaps_g_prev[i].img();
    if ((rc = transformMapsDev(&v, &n, pose16_dev, s))) return rc;
  }
  return DMS_OK;
}

// initRGBModel with the image chosen on device; rgba_tmp: W*H*4-byte scratch
int odometry_initRGBModel_sel(dms_odometry* o, const void* rgbaA, const void* rgbaB, const int* flag_dev, int force_b, void* rgba_tmp,
                              hipStream_t s) {
  const size_t n16 = ((size_t)o->width * o->height + 3) / 4;  // 4 pixels per 16 bytes (buffers are padded to 16 B)
  int rc;
  if ((rc = selectCopy16(rgba_tmp, rgbaA, rgbaB, flag_dev, force_b, n16, s))) return rc;
  dms_image2d img;
  img.data = rgba_tmp;
  img.pitch = (size_t)o->width * 4;
  img.rows = o->height;
  img.cols = o->width;
  return populateRGBDData(o, &img, o->lastDepth, o->lastImage, s);
}

}  // namespace dms

extern "C" {

int dms_odometry_initModelFused(dms_odometry* o, const void* vertA, const void* normA, const void* rgbaA, const void* vertB,
                                const void* normB, const void* rgbaB, const int* use_b_dev, int force_b_image,
                                const float* modelPose16_dev, dms_stream s) {
  DMS_REQUIRE(o, "null odometry");
  return odometry_initModel_fused(o, vertA, normA, rgbaA, vertB, normB, rgbaB, use_b_dev, force_b_image, modelPose16_dev, (hipStream_t)s, 0, nullptr, 0, nullptr);
}

int dms_odometry_fetch_result(dms_odometry* o, dms_track_result* r, dms_stream st) {
  DMS_REQUIRE(o && r, "null argument");
  DMS_HIP(hipMemcpyAsync(o->host_state, o->state, sizeof(TrackState), hipMemcpyDeviceToHost, (hipStream_t)st));
  DMS_HIP(hipStreamSynchronize((hipStream_t)st));
  drain_timers(o);
  const TrackState* h = o->host_state;
  memcpy(r->trans, h->out_trans, sizeof(r->trans));
  memcpy(r->rot, h->out_rot, sizeof(r->rot));
  r->lastICPError = h->lastICPError;
  r->lastICPCount = h->lastICPCount;
  r->lastRGBError = h->lastRGBError;
  r->lastRGBCount = h->lastRGBCount;
  r->lastSO3Error = h->lastSO3Error;
  r->lastSO3Count = h->lastSO3Count;
  memcpy(r->lastA, h->lastA, sizeof(r->lastA));
  memcpy(r->lastb, h->lastb, sizeof(r->lastb));
  for (int l = 0; l < DMS_NUM_PYRS; ++l) r->iterations_run[l] = h->iters_run[l];
  r->so3_iterations_run = h->so3_iters;
  r->rejected_jump = h->rejected_jump;
  if (h->sync_timeout) {
    if (h->sync_timeout == 1 && o->resident) {  // its blocks were not all resident: never again on this handle
      o->resident = false;
      o->fell_back = true;
    }
    if (h->sync_timeout == 2)
      set_error("dms_odometry_fetch_result: no fixed-point range found for a cross-pixel sum (the retry pool of a resident kernel is "
                "exhausted: non-finite input maps?)");
    else
      set_error("dms_odometry_fetch_result: a resident tracker kernel timed out at a grid-wide wait (its blocks were not all on the device: "
                "another process on this GPU?); this call's result is invalid and the handle has switched to DMS_TRACK_MODE=launches");
    return DMS_ERR_TIMEOUT;
  }
  return DMS_OK;
}

int dms_odometry_getIncrementalTransformation(dms_odometry* o, float* trans, float* rot, int rgbOnly, float icpWeight, int pyramid,
                                              int fastOdom, int so3, int interMap, dms_track_result* result, dms_stream s) {
  DMS_REQUIRE(o && trans && rot, "null argument");
  int rc = dms_odometry_track_async(o, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap, s);
  if (rc) return rc;
  dms_track_result local;
  dms_track_result* r = result ? result : &local;
  const bool was_resident = o->resident;
  rc = dms_odometry_fetch_result(o, r, s);
  if (rc == DMS_ERR_TIMEOUT && was_resident && !o->resident) {
    // The resident kernels could not be co-resident: the same call again, launch-per-phase (the inputs are untouched; the
    // SO3 image swap of the first attempt, RGBDOdometry.cpp:595-601, is undone first).  Same bits as a resident run would give.
    if (so3)
      for (int i = 0; i < DMS_NUM_PYRS; i++) std::swap(o->lastNextImage[i], o->nextImage[i]);
    rc = dms_odometry_track_async(o, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap, s);
    if (rc) return rc;
    rc = dms_odometry_fetch_result(o, r, s);
  }
  if (rc) return rc;
  memcpy(trans, r->trans, 3 * sizeof(float));
  memcpy(rot, r->rot, 9 * sizeof(float));
  return DMS_OK;
}

int dms_odometry_getCovariance(dms_odometry* o, double* cov36) {
  DMS_REQUIRE(o && cov36, "null argument");
  // lastA.lu().inverse() (RGBDOdometry.cpp:607-610): Gauss-Jordan with partial pivoting
  double a[36], inv[36];
  memcpy(a, o->host_state->lastA, sizeof(a));
  for (int i = 0; i < 36; ++i) inv[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(a[r * 6 + c]) > fabs(a[p * 6 + c])) p = r;
    if (p != c)
      for (int j = 0; j < 6; ++j) {
        std::swap(a[c * 6 + j], a[p * 6 + j]);
        std::swap(inv[c * 6 + j], inv[p * 6 + j]);
      }
    const double d = a[c * 6 + c];
    for (int j = 0; j < 6; ++j) {
      a[c * 6 + j] /= d;
      inv[c * 6 + j] /= d;
    }
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[r * 6 + c];
      for (int j = 0; j < 6; ++j) {
        a[r * 6 + j] -= f * a[c * 6 + j];
        inv[r * 6 + j] -= f * inv[c * 6 + j];
      }
    }
  }
  memcpy(cov36, inv, sizeof(inv));
  return DMS_OK;
}

int dms_odometry_get_buffer(dms_odometry* o, int which, int level, dms_image2d* view) {
  DMS_REQUIRE(o && view, "null argument");
  DMS_REQUIRE(level >= 0 && level < DMS_NUM_PYRS, "bad level");
  const Buf* b = nullptr;
  switch (which) {
    case 0: b = &o->vmaps_curr[level]; break;
    case 1: b = &o->nmaps_curr[level]; break;
    case 2: b = &o->vmaps_g_prev[level]; break;
    case 3: b = &o->nmaps_g_prev[level]; break;
    case 4: b = &o->lastDepth[level]; break;
    case 5: b = &o->nextDepth[level]; break;
    case 6: b = &o->lastImage[level]; break;
    case 7: b = &o->nextImage[level]; break;
    case 8: b = &o->lastNextImage[level]; break;
    case 9: b = &o->nextdIdx[level]; break;
    case 10: b = &o->nextdIdy[level]; break;
    case 11: b = &o->pointClouds[level]; break;
    case 12: b = &o->depth_tmp[level]; break;
    case 13: b = &o->corresImg[level]; break;
    case 14: b = &o->nextGate[level]; break;
    default: DMS_REQUIRE(false, "bad buffer id");
  }
  *view = b->img();
  return DMS_OK;
}

int dms_odometry_set_profiling(dms_odometry* o, int enabled) {
  DMS_REQUIRE(o, "null argument");
  o->profiling = enabled != 0;
  o->profiling_level0_only = enabled == 2;
  if (enabled) {
    drain_timers(o);
    o->times.clear();
    DMS_HIP(hipDeviceSynchronize());  // (the phase clocks are accumulated by the resident kernels)
    DMS_HIP(hipMemset(o->prof, 0, (3 * 16 + 256 * 8) * 8));
  }
  return DMS_OK;
}

int dms_odometry_get_kernel_time(dms_odometry* o, const char* name, double* total_ms, int* launches) {
  DMS_REQUIRE(o && name && total_ms && launches, "null argument");
  // "phase:<i>": accumulated in-kernel clock of phase i of the persistent level kernels (block 0)
  if (strncmp(name, "phase:", 6) == 0) {
    const int i = atoi(name + 6);  // level * 16 + phase
    DMS_REQUIRE(i >= 0 && i < 48 + 256 * 8, "bad phase index");  // from 48 on: per-block stamps [block][8] of level 0's middle iteration
    long long v = 0;
    DMS_HIP(hipMemcpy(&v, o->prof + i, sizeof(v), hipMemcpyDeviceToHost));
    *total_ms = i < 48 ? (double)v * 1e-5 : (double)(v % 100000000ll) * 1e-5;  // (counts: value * 1e-5)  // 10 ns ticks (stamps: modulo 1 s)
    *launches = 1;
    return DMS_OK;
  }
  {  // "<kernel>:median" / ":max" (ms) and ":slow" (launches = how many took more than 1.5 x the median)
    const char* colon = strchr(name, ':');
    if (colon) {
      auto it2 = o->times.find(std::string(name, colon - name));
      *total_ms = 0;
      *launches = 0;
      if (it2 == o->times.end() || it2->second.samples.empty()) return DMS_OK;
      std::vector<float> v = it2->second.samples;
      std::sort(v.begin(), v.end());
      const float med = v[v.size() / 2];
      if (strcmp(colon, ":median") == 0) *total_ms = med;
      else if (strcmp(colon, ":max") == 0) *total_ms = v.back();
      else if (strcmp(colon, ":slow") == 0) {
        int n = 0;
        double extra = 0;
        for (float x : v)
          if (x > 1.5f * med) {
            ++n;
            extra += x - med;
          }
        *launches = n;
        *total_ms = extra;  // what the slow launches took beyond the median, summed
      } else DMS_REQUIRE(false, "unknown suffix");
      if (strcmp(colon, ":slow") != 0) *launches = (int)v.size();
      return DMS_OK;
    }
  }
  auto it = o->times.find(name);
  if (it == o->times.end()) {
    *total_ms = 0;
    *launches = 0;
    return DMS_OK;
  }
  *total_ms = it->second.ms;
  *launches = it->second.launches;
  return DMS_OK;
}

}  // extern "C"
