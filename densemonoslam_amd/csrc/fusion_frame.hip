// Object layer: one camera's frame step — the part of ElasticFusion::processFrame
// (Core/src/ElasticFusion.cpp:99-637) that runs per frame with loop closure off (--o) and NID
// keyframing off (--nkf) — plus the extern "C" wrappers of the fusion operators.
//
// MI355X design: the whole frame is enqueued on one HIP stream without a single host
// synchronisation.  The camera pose never leaves HBM between tracking and fusion: the tracker
// writes it into the context's pose block, a one-lane kernel derives the inverse and the
// velocity weight (ElasticFusion.cpp:252-268), and every later kernel reads the block through a
// pointer.  The fill-in decision (`denseEnough`, ElasticFusion.cpp:84-97,166-167), which the
// reference takes on the host after a glReadPixels, is a device flag consumed by a select-copy.
#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "internal.hpp"
#include "smallmath.hpp"
#include "surfel.hpp"
#include "canon.hpp"
#include "frame_state.hpp"
#include "live_bodies.hpp"
#include "fill.hpp"
#include "track_init.hpp"

struct dms_odometry;

namespace dms {
// fusion_pre.hip
int depth_bilateral(const dms_image2d* src, dms_image2d* dst, float maxD, hipStream_t s, int narrow_blocks = 0,
                    const dms_image2d* metric_filtered = nullptr, const dms_image2d* depth_l0 = nullptr, const dms_image2d* vmap_l0 = nullptr,
                    const dms_camera* cam_l0 = nullptr, float vmap_cutoff = 0.f);
int depth_metric(const dms_image2d* src, dms_image2d* dst, float maxD, hipStream_t s);
int fill_in(const dms_predict_out* ex, const dms_image2d* depth, const dms_image2d* rgba, const dms_camera* cam, int pass_geom,
            int pass_rgb, dms_predict_out* out, hipStream_t s, const void* mirror_src = nullptr, void* mirror_dst = nullptr,
            int mirror_bytes = 0, int* dense_flag = nullptr);
int fill_args(const dms_predict_out* ex, const dms_image2d* depth, const dms_image2d* rgba, const dms_camera* cam, int pass_geom, int pass_rgb,
              dms_predict_out* out, const void* mirror_src, void* mirror_dst, int mirror_bytes, int* dense_flag, FillArgs* res);
int resize_nn(const dms_image2d* src, dms_image2d* dst, int elem, hipStream_t s);
// fusion_map.hip
int model_initialise(dms_model* m, const dms_image2d* rgba, const dms_image2d* dm, const dms_image2d* dmf, const dms_camera* cam, int time,
                     int timeIdx, float maxDepth, hipStream_t s);
int index_map(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, int time, int timeIdx, float maxDepth, int timeDelta,
              unsigned long long* zbuf, dms_indexmap_out* out, int transposed, int zclean, hipStream_t s);
int clear_zbuf(unsigned long long* zbuf, int n, hipStream_t s);
int untranspose(const void* src, void* dst, int cols, int rows, int elem, hipStream_t s);
int splat_predict(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, float confThreshold, int time,
                  int timeIdx, int maxTime, int timeDelta, int active, unsigned long long* zbuf, dms_predict_out* out,
                  dms_image2d* depth_out, int zclean, hipStream_t s, const float* second_conf_time_maxtime = nullptr,
                  unsigned long long* zbuf2 = nullptr, int resolve_only = 0, const FillArgs* fill = nullptr, const TrackInitArgs* init = nullptr);
int model_sample_graph(dms_model* m, int sampleRate, float* rows4_host, int max_rows, int* n_host, hipStream_t s);
int model_flush_pending(dms_model* m, hipStream_t s);
// fusion_fuse.hip
int model_fuse(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_image2d* rgba, const dms_image2d* dr,
               const dms_image2d* drf, const dms_indexmap_out* im, const dms_camera* cam, float depthCutoff, float weighting,
               const float* weighting_dev, int transposed, hipStream_t s, int defer_update = 0);
int model_clean(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_indexmap_out* im, const dms_image2d* depth_synth,
                const dms_camera* cam, float confThreshold, const float* graph_host, int graph_nodes, int timeDelta, float maxDepth,
                int isFern, int transposed, unsigned* count_out2, hipStream_t s);
// track.hip
struct FrameState;
int odometry_track_enqueue(dms_odometry* o, const float* trans, const float* rot, const float* prior_pose16_dev, int rgbOnly,
                           float icpWeight, int pyramid, int fastOdom, int so3, int interMap, hipStream_t s, FrameState* frame,
                           float weightMultiplier);
int odometry_result_pose(dms_odometry* o, float* pose16_dev, hipStream_t s);
int odometry_initICPModel_sel(dms_odometry* o, const float* vA, const float* nA, const float* vB, const float* nB, const int* flag_dev,
                              const float* pose16_dev, hipStream_t s);
int odometry_initRGBModel_sel(dms_odometry* o, const void* rgbaA, const void* rgbaB, const int* flag_dev, int force_b, void* rgba_tmp,
                              hipStream_t s);
struct TrackFold {  // the track call a model pyramid launch prepares (track.hip)
  const float* prior_pose16;
  int pyramid, fastOdom, so3, interMap;
};
int odometry_early_init_args(dms_odometry* o, const TrackFold* fold, TrackInitArgs* ti);
int odometry_initModel_fused(dms_odometry* o, const void* vA, const void* nA, const void* iA, const void* vB, const void* nB,
                             const void* iB, const int* flag_dev, int force_b_img, const float* pose16_dev, hipStream_t s,
                             int defer_last_step = 0, unsigned* dense_cnt = nullptr, int dense_samples = 0, const TrackFold* fold = nullptr);
int odometry_initLive_fused(dms_odometry* o, const void* verts, const void* norms, const void* rgba, const int* any_flag_dev,
                            hipStream_t s);
int odometry_enable_ring(dms_odometry* o);
int odometry_free_cus(const dms_odometry* o);
void odometry_set_early_exit(dms_odometry* o, int on);
int odometry_next_buffers(dms_odometry* o, int level, dms_image2d* nextImage, dms_image2d* nextDepth);
struct LoopState;
int odometry_loop_candidate(dms_odometry* o, FrameState* frame, const dms_image2d* vertex, const dms_image2d* oldTime, float maxDepth,
                            LoopState* out, float* cons, hipStream_t s);
// nid.hip
size_t nid_workspace_bytes(int num_bins);
int computeNIDImg(const dms_image2d* img_kf, const dms_image2d* img_kf_old, const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old,
                  const dms_image2d* img_curr, int num_bins, void* workspace, size_t workspace_bytes, float* nid_host, float* nid_dev,
                  hipStream_t s);
int computeNIDDepth(const dms_image2d* dmap_kf, const dms_image2d* dmap_kf_old, const dms_image2d* dmap_curr, int num_bins, float max_depth,
                    void* workspace, size_t workspace_bytes, float* nid_host, float* nid_dev, hipStream_t s);
void odometry_bind_live(dms_odometry* o, int k);
void odometry_bind_lastnext(dms_odometry* o, int k);
int odometry_initRGB_image(dms_odometry* o, const dms_image2d* rgba, hipStream_t s);
int odometry_live_views(dms_odometry* o, dms_image2d* depth, dms_image2d* vmap, dms_image2d* nmap, dms_image2d* image, dms_image2d* dx,
                        dms_image2d* dy, dms_image2d* gate, float* minScale, bool mark_derivatives);
int liveLevelsFused(const dms_image2d* depth, const dms_image2d* vmap, const dms_image2d* nmap, const dms_image2d* image, const dms_image2d* dx,
                    const dms_image2d* dy, const dms_image2d* gate, const dms_camera* cam0, float cutoff, const float* minScale, hipStream_t s);
void odometry_alias_next_depth(dms_odometry* o);

// RGB8 (3 B/px, as the reference uploads, ElasticFusion.cpp:111) -> RGBA8 texture
__global__ void k_rgb_to_rgba(const unsigned char* __restrict__ rgb, uchar4* __restrict__ rgba, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x)
    rgba[i] = make_uchar4(rgb[3 * i + 0], rgb[3 * i + 1], rgb[3 * i + 2], 255);
}

// The whole "upload" of a frame in one pass over its pixels (ElasticFusion.cpp:111-119 + the first step of initRGB): the
// colour texel (RGB8 -> RGBA8, or an RGBA8 copy), the raw depth, its metric form (metriciseDepth of the raw image,
// depth_metric.frag:28-39) and — when the tracker wants it — the level-0 intensity of that texel (bgr2Intensity).
__global__ __launch_bounds__(256) void k_live_ingest(const unsigned char* __restrict__ rgb, int channels, const unsigned short* __restrict__ depth,
                                                     uchar4* __restrict__ rgba, unsigned short* __restrict__ depth_raw,
                                                     float* __restrict__ depth_metric, unsigned char* __restrict__ intensity, int n, float maxD) {
  const unsigned gate = (unsigned)f2i_rz(maxD * 1000.0f);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    const uchar4 c = channels == 3 ? make_uchar4(rgb[3 * i + 0], rgb[3 * i + 1], rgb[3 * i + 2], 255) : reinterpret_cast<const uchar4*>(rgb)[i];
    rgba[i] = c;
    const unsigned v = depth[i];
    depth_raw[i] = (unsigned short)v;
    depth_metric[i] = (v > gate || v < 300U) ? 0.f : (float)v / 1000.0f;
    if (intensity) intensity[i] = live::intensity_of(c);
  }
}

struct Pose16 {
  float v[16];
};

// currPose = prior (host value) or keep the device pose; lastPose = pose before this frame
__global__ void k_frame_begin(FrameState* st, Pose16 prior, int have_prior) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) st->lastPose[i] = st->cur.pose[i];
  if (have_prior)
    for (int i = 0; i < 16; ++i) st->cur.pose[i] = prior.v[i];
  sm::inv4t<float>(st->cur.pose, st->cur.t_inv);
}

__global__ void k_pose_set(FrameState* st, Pose16 p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) st->cur.pose[i] = st->lastPose[i] = p.v[i];
  sm::inv4t<float>(st->cur.pose, st->cur.t_inv);
  st->weighting = 1.f;
  st->fill_in = 0;
}

// context.currPose() = estPose after an accepted deformation (ElasticFusion.cpp:489): the velocity weight of
// this frame is already fixed (:252-268); the next frame's lastPose is the new pose (:158)
__global__ void k_pose_override(FrameState* st, Pose16 p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) st->cur.pose[i] = st->lastPose[i] = p.v[i];
  sm::inv4t<float>(st->cur.pose, st->cur.t_inv);
}

__global__ void k_frame_after_track(FrameState* st, float weightMultiplier) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  frame_after_track_body(st, weightMultiplier);
}

// The W/8 x H/8 thumbnails a camera publishes per frame in collaborative mode (the fern matcher's
// inputs: Resize of the fill-in image / vertex / normal textures, Ferns.cpp:277-423) in
// one launch: block = [RGBA8 image | RGBA32F vertex | RGBA32F normal], each tw x th, NEAREST as resize.frag
__global__ __launch_bounds__(256) void k_thumbnails(const uchar4* __restrict__ image, const float4* __restrict__ vertex,
                                                    const float4* __restrict__ normal, int cols, int rows, int tw, int th,
                                                    unsigned char* __restrict__ block, const float* __restrict__ pose16, float* pose_dst,
                                                    int* tick_dst, int tick) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (pose_dst && k < 16) pose_dst[k] = pose16[k];  // (the frame block of collaborative mode: pose and tick ride along)
  if (tick_dst && k == 0) *tick_dst = tick;
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

// ORB-triggered global loop closure (ElasticFusion.cpp:292-326 inside processFrame; ElasticFusion::applyGlobalLoop, :1148-1200):
// the two poses the ORB-SLAM3 front end hands over, as pose blocks (pose + inverse) in HBM
__global__ void k_pose_blocks2(dms_pose_block* pb, Pose16 a, Pose16 b) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) {
    pb[0].pose[i] = a.v[i];
    pb[1].pose[i] = b.v[i];
  }
  sm::inv4t<float>(pb[0].pose, pb[0].t_inv);
  sm::inv4t<float>(pb[1].pose, pb[1].t_inv);
}

// One block: the W/20 x H/20 NEAREST samples of the ACTIVE vertex map (Resize::vertex -> consBuff) and of the INACTIVE time
// map (Resize::time -> timesBuff), walked columns outer / rows inner (:303-304); a sample with 0 < z < maxDepth (and, in
// applyGlobalLoop, a non-zero time) gives one constraint row {orbTcwOld * p, orbTcwNew * p, time, 0} = the arguments of
// Deformation::addConstraint (:319-323).  pb[0] = orbTcwOld, pb[1] = orbTcwNew.  out[0] = number of rows.
__global__ __launch_bounds__(256) void k_global_loop_constraints(const float4* __restrict__ vertex, const unsigned short* __restrict__ oldTime,
                                                                 int cols, int rows, float maxDepth, const dms_pose_block* __restrict__ pb,
                                                                 int require_time, int* __restrict__ out, float* __restrict__ cons) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const int dw = cols / 20, dh = rows / 20, n = dw * dh;
  const float* Po = pb[0].pose;
  const float* Pn = pb[1].pose;
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
      valid = p.z > 0.f && p.z < maxDepth && (!require_time || t > 0);
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
        c[r] = Po[r * 4 + 0] * p.x + Po[r * 4 + 1] * p.y + Po[r * 4 + 2] * p.z + Po[r * 4 + 3];
        c[3 + r] = Pn[r * 4 + 0] * p.x + Pn[r * 4 + 1] * p.y + Pn[r * 4 + 2] * p.z + Pn[r * 4 + 3];
      }
      c[6] = (float)t;
      c[7] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s_base;
}

__global__ void k_frame_end(FrameState* st, const unsigned* __restrict__ d_count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->surfels = d_count[0];
}

struct FKernelTime {
  double ms = 0;
  int launches = 0;
};

}  // namespace dms

using namespace dms;

struct dms_fusion {
  dms_fusion_params p;
  dms_camera cam;
  // dms_fusion_arm_frame_block: the frame block the NEXT frame's final resolve + fill-in pass writes (collaborative mode)
  bool thumb_masks_ok = false;
  struct ArmedBlock {
    unsigned char* block = nullptr;
    float* pose_dst = nullptr;
    int* tick_dst = nullptr;
    int tick = 0;
  } armed, armed_now;  // armed: for the next frame; armed_now: taken over by the frame in progress (process_frame_begin)
  bool armed_written = false;
  dms_model* model = nullptr;
  dms_model* own_model = nullptr;  // the map this context created; != model once it has joined another camera's map
  bool adopting = false;           // dms_fusion_import_camera: the next frame seeds the live state and touches no map
  // GlobalModel's per-cluster surfel buffers (GlobalModel.h:99-109; ground-truth-clusters mode).  The constructor makes cluster 0
  // current (GlobalModel.cpp:56-59); a frame that fuses under an id the map does not know adds buffers for it, fills them from
  // the context's FEEDBACK buffers and makes them current (:266-277, ElasticFusion.cpp:508-515) — nothing switches back, and
  // every other member works on the current cluster.  The feedback buffers are computed on the first frame only
  // (ElasticFusion.cpp:133) unless the caller asks again (Context::computeFeedbackBuffers, MainController.cpp:476): what is kept
  // here is their input — colour, raw and filtered metric depth, tick — from which model_initialise makes the same surfels.
  // nullptr = the constructor's empty cluster 0 of a context whose first frame arrived under another id.
  std::map<int, dms_model*> clusters;
  int cur_cluster = 0, req_cluster = 0;
  dms_image2d fb_rgba, fb_dm, fb_dmf;
  int fb_time = 0;
  bool fb_valid = false;
  dms_odometry* odom = nullptr;
  dms_odometry* odom_m2m = nullptr;  // Context::modelToModel() (local loop closure)
  dms_predict_out pred_old;          // IndexMap old* textures: the INACTIVE view
  // ORB-triggered global loop closure (hybrid_loops): the two poses of the next frame (armed by dms_fusion_set_orb_loop), their
  // pose blocks in HBM, the constraint rows (device: 256-byte header with the count, then rows of 8 floats; pinned mirror)
  bool orb_armed = false, gloop_ran = false, in_global_loop = false;
  float orb_old[16], orb_new[16];
  dms_pose_block* orb_pose = nullptr;
  char* gloop = nullptr;
  char* h_gloop = nullptr;
  size_t gloop_bytes = 0;
  LoopState* loop = nullptr;         // device: LoopState + constraint rows
  char* h_loop = nullptr;            // pinned, four slots (frame % 4)
  size_t loop_bytes = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  // images of the incoming frame: two sets, so that frame t+1 can be ingested and filtered on the
  // prep stream while frame t is still being tracked and fused; the unsuffixed names are the set
  // of the frame the main stream is working on
  struct LiveImages {
    dms_image2d rgba, depth_raw, depth_filtered, depth_metric, depth_metric_filtered;
  } live[2];
  dms_image2d rgba, depth_raw, depth_filtered, depth_metric, depth_metric_filtered;
  hipStream_t s_prep = nullptr;
  hipEvent_t ev_prep_done[2] = {nullptr, nullptr}, ev_main_done[4] = {nullptr, nullptr, nullptr, nullptr}, ev_inputs = nullptr;
  bool fused_live = true;  // the live half as three fused launches (DMS_FUSED_LIVE=0: the operator chain, fifteen)
  int prep_blocks = 38;  // fat blocks of the bilateral filter on the prep stream, set at creation (DMS_PREP_BLOCKS; 0 = one tile per block)
  // "Late frame" (round 6): the frame's first kernel needs the live half of the SAME frame, which runs on the prep stream.  A
  // hipStreamWaitEvent for it is a barrier packet on the frame's queue - about 10 us of idle queue per frame although the event fired
  // long before.  When the host has that much slack it can wait for the event ITSELF and enqueue the frame behind it: the runtime then
  // adds no packet (the event is complete).  2 412 - 2 418 -> 2 446 - 2 454 frames/s (driver's form), 2 678 - 2 681 -> 2 711 - 2 715
  // (300 steps).  It costs the host its run-ahead, so it is (i) only for a context that is driven alone (one live context in the
  // process, or its owner says so; dms_session says no for its cameras) on a map of its own, and (ii) watched: the host's wait at the
  // start of a frame (for frame t - 2) is its remaining slack; three frames in a row under 15 us switch the mode off for a while
  // (64 frames, doubling).  Forced on, the populated full step fell from 1 690 to 1 082 frames/s and two cameras from 3 150 to 2 815 -
  // both are cases the two rules exclude.  DMS_LATE_MAIN=0 / 1 forces it.
  bool late_main = false;       // this frame
  int late_forced = -1;         // DMS_LATE_MAIN
  int late_owner = -1;          // dms_fusion_allow_late_frame: the owner's word (-1: none - one live context decides)
  int late_low = 0, late_cool = 0, late_backoff = 64;
  double host_wait_prep_ms = 0.0;
  int host_lag = 2;  // the host enqueues frame t once frame t - host_lag has completed (DMS_HOST_LAG = 2 | 3; 3 measured -2.4 %)
  bool inputs_armed = false;  // ev_inputs was recorded by dms_fusion_inputs_ready for the next frame
  int last_prep = -1;         // image set whose ev_prep_done marks the end of the last enqueued ingest
  long frames = 0;  // frames enqueued so far
  // NID key-framing (fuseFrame): candidate key frame of the current prediction, per pyramid level
  dms_image2d kf_img[DMS_NUM_PYRS], kf_dmap[DMS_NUM_PYRS], kf_old_img[DMS_NUM_PYRS], kf_old_dmap[DMS_NUM_PYRS];
  void* nid_ws = nullptr;
  size_t nid_ws_bytes = 0;
  float* nid_host = nullptr;  // pinned [2]
  int frames_since_fusion = 0;
  float last_nid = 0.f;
  // between dms_fusion_process_frame_begin and _end
  bool in_frame = false, cur_bootstrap = false, cur_fuse_now = true;
  int live_set = 0, lastnext_set = 0;  // ring sets of the previous frame's live pyramids and of its lastNextImage
  bool prev_handed_on = false;         // the previous frame's tracker call ran with so3 (or was the first frame)
  // --rl bookkeeping (ElasticFusion.cpp:204-244)
  bool tracking_ok = true, lost = false;
  int tracking_count = 0;
  int cur_k2 = 0;
  dms_image2d depth_synth;  // IndexMap::synthesizeDepth target (deformation frames only)
  dms_indexmap_out imap;
  dms_predict_out pred, fill;
  void* rgba_tmp = nullptr;
  void* untr = nullptr;  // W*H*16 scratch for row-major copies handed out by dms_fusion_get_image
  unsigned long long* zbuf = nullptr;
  // second z-buffer: filled by the final prediction's project pass with the NEXT frame's tracking prediction (same map,
  // same pose unless the caller brings a prior), resolved at that frame's begin instead of projecting the map again
  unsigned long long* zbuf2 = nullptr;
  unsigned* tickets = nullptr;  // 16 counters, 64 bytes apart (fused fill-in: non-black subsampled pixels of the tracking prediction); subsample masks at word 512
  bool fold_track_init = true;     // the tracker call's set-up as a block group of the model pyramid kernel (DMS_FOLD_TRACK_INIT=0: its own launch)
  bool dense_by_counters = false;  // this frame's denseEnough decision is taken from them by the model pyramid kernel
  double host_wait_ms = 0.0;     // host time spent blocked on the bounded run-ahead ("host_wait" of dms_fusion_get_kernel_time)
  bool pre_valid = false;        // zbuf2 holds a projection
  int pre_tick = 0;              // ... rendered for this tick
  unsigned long pre_version = 0; // ... of this version of the map
  FrameState* state = nullptr;
  FrameState* h_state_dev = nullptr;  // device view of h_state
  FrameState* h_state = nullptr;  // pinned, four slots, frame % 4 (the host may read a slot once that frame's event has completed)
  int last_slot = 0;
  int timeouts_reported = 0;  // value of FrameState::track_timeouts the caller has been told about
  int range_failures_reported = 0;  // the same for FrameState::track_range_failures
  void* h_track = nullptr;
  int tick = 1;
  bool map_initialised = false;
  int fused_last = 0;
  bool profiling = false;
  std::map<std::string, FKernelTime> times;
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<hipEvent_t> pool;
};

namespace {

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Carve {
  char* base = nullptr;
  size_t off = 0;
  void* take(size_t bytes) {
    off = up256(off);
    void* p = base ? base + off : nullptr;
    off += bytes + 16;
    return p;
  }
};

dms_image2d mk_img(void* p, int rows, int cols, size_t elem) {
  dms_image2d i;
  i.data = p;
  i.pitch = (size_t)cols * elem;
  i.rows = rows;
  i.cols = cols;
  return i;
}

void layout(dms_fusion* f, Carve& c) {
  const int W = f->p.width, H = f->p.height;
  const size_t N = (size_t)W * H;
  for (int k = 0; k < 2; ++k) {
    f->live[k].rgba = mk_img(c.take(N * 4), H, W, 4);
    f->live[k].depth_raw = mk_img(c.take(N * 2), H, W, 2);
    f->live[k].depth_filtered = mk_img(c.take(N * 2), H, W, 2);
    f->live[k].depth_metric = mk_img(c.take(N * 4), H, W, 4);
    f->live[k].depth_metric_filtered = mk_img(c.take(N * 4), H, W, 4);
  }
  f->rgba = f->live[0].rgba;
  f->depth_raw = f->live[0].depth_raw;
  f->depth_filtered = f->live[0].depth_filtered;
  f->depth_metric = f->live[0].depth_metric;
  f->depth_metric_filtered = f->live[0].depth_metric_filtered;
  f->fb_rgba = mk_img(c.take(N * 4), H, W, 4);
  f->fb_dm = mk_img(c.take(N * 4), H, W, 4);
  f->fb_dmf = mk_img(c.take(N * 4), H, W, 4);
  f->imap.index = mk_img(c.take(N * 4), H, W, 4);
  f->imap.vertConf = mk_img(c.take(N * 16), H, W, 16);
  f->imap.colorTime = mk_img(c.take(N * 16), H, W, 16);
  f->imap.normRad = mk_img(c.take(N * 16), H, W, 16);
  f->pred.image = mk_img(c.take(N * 4), H, W, 4);
  f->pred.vertex = mk_img(c.take(N * 16), H, W, 16);
  f->pred.normal = mk_img(c.take(N * 16), H, W, 16);
  f->pred.time = mk_img(c.take(N * 2), H, W, 2);
  f->fill.image = mk_img(c.take(N * 4), H, W, 4);
  f->fill.vertex = mk_img(c.take(N * 16), H, W, 16);
  f->fill.normal = mk_img(c.take(N * 16), H, W, 16);
  f->fill.time = f->pred.time;
  for (int l = 0; l < DMS_NUM_PYRS; ++l) {
    const int h = H >> l, w = W >> l;
    f->kf_img[l] = mk_img(c.take((size_t)h * w), h, w, 1);
    f->kf_dmap[l] = mk_img(c.take((size_t)h * w * 4), h, w, 4);
    f->kf_old_img[l] = mk_img(c.take((size_t)h * w), h, w, 1);
    f->kf_old_dmap[l] = mk_img(c.take((size_t)h * w * 4), h, w, 4);
  }
  {
    const int nb = f->p.nid_bins_depth > f->p.nid_bins_img ? f->p.nid_bins_depth : f->p.nid_bins_img;
    f->nid_ws_bytes = nid_workspace_bytes(nb > 0 ? nb : 1);
    f->nid_ws = c.take(f->nid_ws_bytes);
  }
  memset(&f->pred_old, 0, sizeof(f->pred_old));
  if (f->p.local_loop_closure || f->p.hybrid_loops) {
    f->pred_old.image = mk_img(c.take(N * 4), H, W, 4);
    f->pred_old.vertex = mk_img(c.take(N * 16), H, W, 16);
    f->pred_old.normal = mk_img(c.take(N * 16), H, W, 16);
    f->pred_old.time = mk_img(c.take(N * 2), H, W, 2);
  }
  if (f->p.hybrid_loops) {
    f->gloop_bytes = 256 + (size_t)(W / 20) * (H / 20) * 8 * sizeof(float);
    f->gloop = (char*)c.take(f->gloop_bytes);
    f->orb_pose = (dms_pose_block*)c.take(2 * sizeof(dms_pose_block));
  }
  if (f->p.local_loop_closure) {
    f->loop_bytes = up256(sizeof(LoopState)) + (size_t)(W / 20) * (H / 20) * 8 * sizeof(float);
    f->loop = (LoopState*)c.take(f->loop_bytes);
  }
  f->depth_synth = mk_img(c.take(N * 4), H, W, 4);
  f->rgba_tmp = c.take(N * 4);
  f->untr = c.take(N * 16);
  f->zbuf = (unsigned long long*)c.take(N * 8);
  f->zbuf2 = (unsigned long long*)c.take(N * 8);
  f->tickets = (unsigned*)c.take(2048 + 512 + 1024);  // + the subsample masks (128 words at word 512) + the thumbnail masks (256 words at word 640)
  f->state = (FrameState*)c.take(sizeof(FrameState));
}

struct FTimer {
  dms_fusion* f;
  hipStream_t s;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  FTimer(dms_fusion* f_, hipStream_t s_, const char* n) : f(f_), s(s_), name(n) {
    if (!f->profiling) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!f->pool.empty()) {
        e = f->pool.back();
        f->pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, s);
  }
  ~FTimer() {
    if (!f->profiling) return;
    (void)hipEventRecord(b, s);
    f->pending.push_back({name, {a, b}});
  }
};

void drain(dms_fusion* f) {
  // DMS_TIMELINE=1: print every stage's start / end relative to the first stage of the batch
  // (both streams share the clock) — shows whether the prep stream really overlaps the main one
  static const bool timeline = getenv("DMS_TIMELINE") != nullptr;
  if (timeline && !f->pending.empty()) {
    hipEvent_t base = f->pending.front().second.first;
    for (auto& p : f->pending) {
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, base, p.second.first);
      (void)hipEventElapsedTime(&b, base, p.second.second);
      fprintf(stderr, "[timeline] %-14s %9.3f -> %9.3f ms\n", p.first.c_str(), a, b);
    }
  }
  for (auto& p : f->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
      FKernelTime& t = f->times[p.first];
      t.ms += ms;
      t.launches += 1;
    }
    f->pool.push_back(p.second.first);
    f->pool.push_back(p.second.second);
  }
  f->pending.clear();
}

// ElasticFusion::predict (ElasticFusion.cpp:688-746): ACTIVE splat + fill-in
// mode 0: plain.  mode 1 (the frame's final prediction): the project pass also fills zbuf2 for the next frame's tracking
// prediction (confidence 0.7, next tick).  mode 2 (that next frame's begin): resolve zbuf2 if it is still what this
// prediction would render — nothing changed the map, the pose comes from the previous frame — else project as usual.
int predict(dms_fusion* f, float confidence, hipStream_t s, void* state_mirror = nullptr, bool dense_test = false, int mode = 0,
            bool have_prior = false, const TrackInitArgs* track_init = nullptr, bool* track_init_taken = nullptr) {
  int rc;
  FillArgs fa;
  const FillArgs* fused = nullptr;
  // passthrough = lost (geometry), lost || frameToFrameRGB (image) (ElasticFusion.cpp:704-712)
  // (the frame's last fill-in also copies the result block into its pinned host slot)
  if ((rc = fill_args(&f->pred, &f->depth_filtered, &f->rgba, &f->cam, f->lost ? 1 : 0, (f->lost || f->p.frameToFrameRGB) ? 1 : 0, &f->fill,
                      state_mirror ? f->state : nullptr, state_mirror, (int)sizeof(FrameState), dense_test ? &f->state->fill_in : nullptr, &fa)))
    return rc;
  // the denseEnough decision of a fused pass is taken by the tracker's model pyramid kernel from the counters filled here
  // (fill.hpp): only when that kernel follows, i.e. with hybrid tracking
  f->dense_by_counters = false;
  if (f->p.fused_fill_in && f->p.width <= 2048 && f->p.height <= 2048 && (!dense_test || f->p.hybrid_tracking)) {
    fused = &fa;
    if (dense_test) {
      fa.dense_cnt = f->tickets;
      fa.sample_mask = f->tickets + 512;
      f->dense_by_counters = true;
    }
    if (mode == 1 && f->armed_now.block && f->thumb_masks_ok) {  // the frame's last kernel also writes the armed frame block
      fa.thumb_block = f->armed_now.block;
      fa.thumb_mask = f->tickets + 640;
      fa.thumb_w = f->p.width / 8;
      fa.thumb_h = f->p.height / 8;
      fa.thumb_pose_src = f->state->cur.pose;
      fa.thumb_pose_dst = f->armed_now.pose_dst;
      fa.thumb_tick_dst = f->armed_now.tick_dst;
      fa.thumb_tick = f->armed_now.tick;
      f->armed_written = true;
    }
  }
  {
    FTimer t(f, s, "predict");
    const int W = f->p.width, H = f->p.height;
    bool done = false;
    if (mode == 2 && f->pre_valid) {
      if (!have_prior && f->pre_tick == f->tick && f->pre_version == f->model->version) {
        if ((rc = splat_predict(f->model, &f->state->cur, &f->cam, f->p.maxDepthProcessed, confidence, f->tick, f->p.timeIdx, f->tick,
                                f->p.timeDelta, 1, f->zbuf2, &f->pred, nullptr, 1, s, nullptr, nullptr, 1, fused, fused ? track_init : nullptr)))
          return rc;
        done = true;
      } else if ((rc = clear_zbuf(f->zbuf2, W * H, s))) {  // stale: the resolve that would have cleaned it never runs
        return rc;
      }
      f->pre_valid = false;
    }
    if (!done) {
      const int next_tick = f->lost ? f->tick : f->tick + 1;  // (the tick advances at the end of the frame unless the camera is lost)
      const float second[3] = {0.7f, (float)next_tick, (float)next_tick};
      // (a projection rendered ahead for this camera's next frame is stale by then when other cameras fuse into the same map: not rendered)
      const bool dual = mode == 1 && f->p.share_projection && f->p.hybrid_tracking && f->model->sharers == 1;
      if (dual && f->pre_valid && (rc = clear_zbuf(f->zbuf2, W * H, s))) return rc;  // (never consumed)
      if ((rc = splat_predict(f->model, &f->state->cur, &f->cam, f->p.maxDepthProcessed, confidence, f->tick, f->p.timeIdx, f->tick,
                              f->p.timeDelta, 1, f->zbuf, &f->pred, nullptr, 1, s, dual ? second : nullptr, dual ? f->zbuf2 : nullptr, 0,
                              fused, fused ? track_init : nullptr)))
        return rc;
      if (dual) {
        f->pre_valid = true;
        f->pre_tick = next_tick;
        f->pre_version = f->model->version;
      }
    }
  }
  if (track_init_taken) *track_init_taken = fused != nullptr && track_init && track_init->blocks > 0;
  if (!fused) {
    FTimer t(f, s, "fill_in");
    if ((rc = fill_in(&f->pred, &f->depth_filtered, &f->rgba, &f->cam, f->lost ? 1 : 0, (f->lost || f->p.frameToFrameRGB) ? 1 : 0, &f->fill,
                      s, state_mirror ? f->state : nullptr, state_mirror, (int)sizeof(FrameState), dense_test ? &f->state->fill_in : nullptr)))
      return rc;
  }
  return DMS_OK;
}

// Device half of the ORB-triggered global loop closure.  `active_new` = 0: the form inside processFrame (ElasticFusion.cpp:293-326):
// ACTIVE prediction at orbTcwOld, INACTIVE at orbTcwNew, every sample with 0 < z < maxDepthProcessed (the time test is commented
// out there).  1: ElasticFusion::applyGlobalLoop (:1158-1200): ACTIVE at orbTcwNew, INACTIVE at orbTcwOld, time > 0 required.
// Both: constraint = (orbTcwOld * p, orbTcwNew * p, INACTIVE time).  The rows reach the pinned mirror on the stream.
int global_loop_device(dms_fusion* f, const float* orbTcwOld, const float* orbTcwNew, int active_new, hipStream_t s) {
  int rc;
  Pose16 a, b;
  memcpy(a.v, orbTcwOld, sizeof(a.v));
  memcpy(b.v, orbTcwNew, sizeof(b.v));
  hipLaunchKernelGGL(k_pose_blocks2, dim3(1), dim3(64), 0, s, f->orb_pose, a, b);
  DMS_CHECK_LAUNCH();
  FTimer t(f, s, "global_loop");
  const int act = active_new ? 1 : 0;
  // predict(context, rf) with currPose = the ACTIVE pose (:294-296 / :1158-1160): only its vertex map is consumed
  if ((rc = splat_predict(f->model, f->orb_pose + act, &f->cam, f->p.maxDepthProcessed, f->p.confidence, f->tick, f->p.timeIdx, f->tick,
                          f->p.timeDelta, 1, f->zbuf, &f->pred, nullptr, 1, s)))
    return rc;
  // combinedPredict(<other pose>, ..., confidenceThreshold, 0, id, tick - timeDelta, timeDelta, INACTIVE) (:299-302 / :1163-1166)
  if ((rc = splat_predict(f->model, f->orb_pose + (1 - act), &f->cam, f->p.maxDepthProcessed, f->p.confidence, 0, f->p.timeIdx,
                          f->tick - f->p.timeDelta, f->p.timeDelta, 0, f->zbuf, &f->pred_old, nullptr, 1, s)))
    return rc;
  hipLaunchKernelGGL(k_global_loop_constraints, dim3(1), dim3(256), 0, s, (const float4*)f->pred.vertex.data,
                     (const unsigned short*)f->pred_old.time.data, f->p.width, f->p.height, f->p.maxDepthProcessed, f->orb_pose, active_new ? 1 : 0,
                     (int*)f->gloop, (float*)(f->gloop + 256));
  DMS_CHECK_LAUNCH();
  DMS_HIP(hipMemcpyAsync(f->h_gloop, f->gloop, f->gloop_bytes, hipMemcpyDeviceToHost, s));
  f->gloop_ran = true;
  return DMS_OK;
}

}  // namespace

extern "C" {

// ---- operator-layer wrappers ----------------------------------------------------------------
int dms_depth_bilateral(const dms_image2d* d, dms_image2d* o, float maxD, dms_stream s) { return depth_bilateral(d, o, maxD, (hipStream_t)s); }
int dms_depth_metric(const dms_image2d* d, dms_image2d* o, float maxD, dms_stream s) { return depth_metric(d, o, maxD, (hipStream_t)s); }
int dms_model_initialise(dms_model* m, const dms_image2d* rgba, const dms_image2d* dm, const dms_image2d* dmf, const dms_camera* cam,
                         int time, int timeIdx, float maxDepth, dms_stream s) {
  return model_initialise(m, rgba, dm, dmf, cam, time, timeIdx, maxDepth, (hipStream_t)s);
}
int dms_index_map(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, int time, int timeIdx, float maxDepth, int timeDelta,
                  unsigned long long* zbuf, dms_indexmap_out* out, dms_stream s) {
  return index_map(m, pose, cam, time, timeIdx, maxDepth, timeDelta, zbuf, out, 0, 0, (hipStream_t)s);
}
int dms_splat_predict(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, float confThreshold, int time,
                      int timeIdx, int maxTime, int timeDelta, int active, unsigned long long* zbuf, dms_predict_out* out, dms_stream s) {
  DMS_REQUIRE(out, "null output");
  return splat_predict(m, pose, cam, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, active, zbuf, out, nullptr, 0, (hipStream_t)s);
}
int dms_splat_depth(dms_model* m, const dms_pose_block* pose, const dms_camera* cam, float maxDepth, float confThreshold, int time,
                    int timeIdx, int maxTime, int timeDelta, unsigned long long* zbuf, dms_image2d* depth, dms_stream s) {
  DMS_REQUIRE(depth, "null output");
  // synthesizeDepth never sets the `actv` uniform (IndexMap.cpp:370-452): it stays false
  return splat_predict(m, pose, cam, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, 0, zbuf, nullptr, depth, 0, (hipStream_t)s);
}
int dms_model_sample_graph(dms_model* m, int sampleRate, float* rows4_host, int max_rows, int* n, dms_stream s) {
  return model_sample_graph(m, sampleRate, rows4_host, max_rows, n, (hipStream_t)s);
}
int dms_model_fuse(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_image2d* rgba, const dms_image2d* dr,
                   const dms_image2d* drf, const dms_indexmap_out* im, const dms_camera* cam, float depthCutoff, float weighting,
                   const float* weighting_dev, dms_stream s) {
  return model_fuse(m, pose, time, timeIdx, rgba, dr, drf, im, cam, depthCutoff, weighting, weighting_dev, 0, (hipStream_t)s);
}
int dms_model_clean(dms_model* m, const dms_pose_block* pose, int time, int timeIdx, const dms_indexmap_out* im,
                    const dms_image2d* depth_synth, const dms_camera* cam, float confThreshold, const float* graph_host, int graph_nodes,
                    int timeDelta, float maxDepth, int isFern, dms_stream s) {
  return model_clean(m, pose, time, timeIdx, im, depth_synth, cam, confThreshold, graph_host, graph_nodes, timeDelta, maxDepth, isFern, 0,
                     nullptr, (hipStream_t)s);
}
int dms_fill_in(const dms_predict_out* ex, const dms_image2d* d, const dms_image2d* rgba, const dms_camera* cam, int pg, int pr,
                dms_predict_out* out, dms_stream s) {
  return fill_in(ex, d, rgba, cam, pg, pr, out, (hipStream_t)s);
}
int dms_resize_nn(const dms_image2d* src, dms_image2d* dst, int elem, dms_stream s) { return resize_nn(src, dst, elem, (hipStream_t)s); }

// ---- frame object ---------------------------------------------------------------------------
void dms_fusion_default_params(dms_fusion_params* p, int width, int height, float fx, float fy, float cx, float cy) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->width = width;
  p->height = height;
  p->fx = fx;
  p->fy = fy;
  p->cx = cx;
  p->cy = cy;
  p->timeDelta = 200;       // Options.h:90
  p->confidence = 10.0f;    // Options.h:93
  p->depthCut = 3.0f;       // Options.h:93
  p->icpWeight = 10.0f;     // Options.h:93
  p->fastOdom = 0;
  p->so3 = 1;
  p->frameToFrameRGB = 0;
  p->pyramid = 1;           // ElasticFusion.cpp:59
  p->hybrid_tracking = 1;
  p->rgbOnly = 0;
  p->timeIdx = 0;
  p->maxDepthProcessed = 25.0f;  // ElasticFusion.cpp:56
  p->model_capacity = 0;
  p->pipeline_ingest = 1;
  p->global_predict = 0;
  p->nid_keyframing = 0;
  p->nid_threshold = 0.80f;    // ElasticFusion.h:73-74
  p->nid_depth_lambda = 0.7f;
  p->nid_bins_img = 64;
  p->nid_bins_depth = 500;
  p->nid_pyramid_level = 0;
  p->local_loop_closure = 0;
  p->reloc = 0;
  p->num_sensors = 3;       // NUM_CAMERAS (Shaders/size.glsl:2)
  p->share_projection = 1;
  p->fused_fill_in = 1;
  p->hybrid_loops = 0;
}

// far depth cut-offs raise the static exponents of the trackers' first reductions (canon.hpp): set at creation and whenever the
// cut-off changes, in a field of its own beside the "exp_bias" test hook
static int set_depth_bias(dms_fusion* f) {
  const int b = canon::depth_exp_bias(f->p.depthCut);
  int rc = dms_odometry_debug_set(f->odom, "depth_exp_bias", b);
  if (!rc && f->odom_m2m) rc = dms_odometry_debug_set(f->odom_m2m, "depth_exp_bias", b);
  return rc;
}

static std::atomic<int> g_live_contexts{0};
int dms_fusion_allow_late_frame(dms_fusion* f, int allow) {
  DMS_REQUIRE(f && allow >= -1 && allow <= 1, "bad argument");
  f->late_owner = allow;
  return DMS_OK;
}

int dms_fusion_create(dms_fusion** out, const dms_fusion_params* p) {
  DMS_REQUIRE(out && p, "null argument");
  DMS_REQUIRE(p->width >= 40 && p->height >= 40, "resolution too small");
  DMS_REQUIRE(p->timeIdx >= 0 && p->timeIdx < DMS_MAX_SENSORS, "timeIdx out of range");
  DMS_REQUIRE(p->num_sensors >= 0 && p->num_sensors <= DMS_MAX_SENSORS, "num_sensors out of range");
  // (0 = a zero-initialised params block of an older caller = the reference's 3: the default is applied BEFORE the slot is checked)
  DMS_REQUIRE(p->timeIdx < (p->num_sensors == 0 ? 3 : p->num_sensors), "timeIdx must be one of the num_sensors time slots");
  DMS_REQUIRE(!p->nid_keyframing || (p->nid_bins_img >= 1 && p->nid_bins_img <= 256 && p->nid_bins_depth >= 1 && p->nid_bins_depth <= 4096 &&
                                     p->nid_pyramid_level >= 0 && p->nid_pyramid_level < DMS_NUM_PYRS),
              "bad NID key-framing parameters");
  dms_fusion* f = new dms_fusion();
  f->p = *p;
  f->cam.fx = p->fx;
  f->cam.fy = p->fy;
  f->cam.cx = p->cx;
  f->cam.cy = p->cy;
  if (f->p.num_sensors == 0) f->p.num_sensors = 3;  // (a zero-initialised params block of an older caller)
  int rc = dms_model_create(&f->model, p->model_capacity, p->width, p->height);
  if (rc) {
    delete f;
    return rc;
  }
  (void)dms_model_set_num_sensors(f->model, f->p.num_sensors);
  f->own_model = f->model;
  f->clusters[0] = f->model;
  rc = dms_odometry_create(&f->odom, p->width, p->height, p->cx, p->cy, p->fx, p->fy, 0.f, 0.f);
  if (rc) {
    dms_model_destroy(f->own_model ? f->own_model : f->model);
    delete f;
    return rc;
  }
  rc = odometry_enable_ring(f->odom);
  if (!rc && p->local_loop_closure) rc = dms_odometry_create(&f->odom_m2m, p->width, p->height, p->cx, p->cy, p->fx, p->fy, 0.f, 0.f);
  if (!rc) rc = set_depth_bias(f);
  if (!rc && f->odom_m2m) odometry_set_early_exit(f->odom_m2m, 1);  // its INACTIVE side is empty on most frames
  if (rc) {
    dms_odometry_destroy(f->odom);
    dms_model_destroy(f->own_model ? f->own_model : f->model);
    delete f;
    return rc;
  }
  Carve sz;
  layout(f, sz);
  f->arena_bytes = up256(sz.off);
  hipError_t e = hipMalloc((void**)&f->arena, f->arena_bytes);
  if (e == hipSuccess) {
    // The live half's stream at the LOWEST priority - not for the priority (its kernels fit beside the frame's either way: 2 430 / 2 413
    // against 2 436 / 2 418 frames/s, driver's form) but for the hardware queue: the runtime spreads the streams of one priority over
    // GPU_MAX_HW_QUEUES (4) queues in creation order, and a process with a fifth stream (a session: caller's, the maps', every camera's
    // live half, the default stream, RCCL's) can find a camera's live half on the queue of the frame it feeds - the pre-processing then
    // runs IN FRONT of the frame instead of beside the previous one (measured: the one-rank RCCL session loop at 0.59 of the bare frame
    // rate with GPU_MAX_HW_QUEUES=6, 0.93 with this; two / three cameras on one GPU 2 632 / 2 267 against 2 422 / 2 055 frames/s, four
    // 2 529 against 2 628).  Streams of another priority have queues of their own.  DMS_PREP_PRIORITY=0: the default priority again.
    int prio = 1;
    if (const char* pp = getenv("DMS_PREP_PRIORITY")) prio = atoi(pp);  // 1: lowest, -1: highest, 0: default
    int least = 0, greatest = 0;
    if (prio != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      e = hipStreamCreateWithPriority(&f->s_prep, hipStreamNonBlocking, prio > 0 ? least : greatest);
    else
      e = hipStreamCreateWithFlags(&f->s_prep, hipStreamNonBlocking);
  }
  for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&f->ev_prep_done[k], hipEventDisableTiming);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&f->ev_main_done[k], hipEventDisableTiming);
  {  // Fat blocks of the depth pre-filter beside the tracker: at most 40, and never more than the resident tracker kernels leave
    // free — a fat block holds a whole compute unit for the length of the filter, and a resident grid that finds fewer free units
    // than it has blocks starts incomplete, its blocks spinning until the filter's retire.  Measured at 640x480 (round 4, the
    // table form of the filter: 112 us on 56 units), driver's form: 16 / 24 / 32 / 40 / 43 / 48 / 50 / 56 / 60 blocks ->
    // 2300 / 2326 / 2375 / 2386 / 2387 / 2388 / 2334 / 2305 / 2290 frames/s; the leg with three predictions per frame: 2412 at
    // 34 - 40 blocks, 2292 at 43, 2370 at 56.  Below the cap the count is the smallest that keeps the number of rounds over the
    // image's 64 x 16 tiles (300 tiles, 8 rounds: 38 blocks).
    const int tiles = ((p->width + 63) / 64) * ((p->height + 15) / 16);
    int cap = 40;
    const int free_cus = odometry_free_cus(f->odom);
    if (free_cus >= 16 && free_cus < cap) cap = free_cus;
    const int rounds = (tiles + cap - 1) / cap;
    f->prep_blocks = (tiles + rounds - 1) / rounds;
  }
  if (const char* pb = getenv("DMS_PREP_BLOCKS")) f->prep_blocks = atoi(pb);
  if (const char* fl = getenv("DMS_FUSED_LIVE")) f->fused_live = atoi(fl) != 0;
  if (const char* ft = getenv("DMS_FOLD_TRACK_INIT")) f->fold_track_init = atoi(ft) != 0;
  if (const char* hl = getenv("DMS_HOST_LAG")) f->host_lag = atoi(hl) == 3 ? 3 : 2;
  if (const char* lm = getenv("DMS_LATE_MAIN")) f->late_forced = atoi(lm) != 0 ? 1 : 0;
  if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_inputs, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMemset(f->arena, 0, f->arena_bytes);
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->h_state, 4 * sizeof(FrameState), hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&f->h_state_dev, f->h_state, 0);
  if (e == hipSuccess) e = hipHostMalloc((void**)&f->nid_host, 2 * sizeof(float), hipHostMallocDefault);
  if (e == hipSuccess && f->loop_bytes) {
    e = hipHostMalloc((void**)&f->h_loop, 4 * f->loop_bytes, hipHostMallocDefault);
    if (e == hipSuccess) memset(f->h_loop, 0, 4 * f->loop_bytes);
  }
  if (e == hipSuccess && f->gloop_bytes) {
    e = hipHostMalloc((void**)&f->h_gloop, f->gloop_bytes, hipHostMallocDefault);
    if (e == hipSuccess) memset(f->h_gloop, 0, f->gloop_bytes);
  }
  if (e != hipSuccess) {
    if (f->arena) (void)hipFree(f->arena);
    if (f->odom_m2m) dms_odometry_destroy(f->odom_m2m);
    dms_odometry_destroy(f->odom);
    dms_model_destroy(f->own_model ? f->own_model : f->model);
    delete f;
    return hip_fail(e, "dms_fusion_create allocation", __FILE__, __LINE__);
  }
  Carve c;
  c.base = f->arena;
  layout(f, c);
  {
    unsigned masks[128];
    fill_sample_masks(p->width < 2048 ? p->width : 2048, p->height < 2048 ? p->height : 2048, masks);
    (void)hipMemcpy(f->tickets + 512, masks, sizeof(masks), hipMemcpyHostToDevice);
    unsigned tm[256];
    f->thumb_masks_ok = thumb_sample_masks(p->width, p->height, tm);
    (void)hipMemcpy(f->tickets + 640, tm, sizeof(tm), hipMemcpyHostToDevice);
  }
  Pose16 I;
  for (int i = 0; i < 16; ++i) I.v[i] = (i % 5 == 0) ? 1.f : 0.f;
  hipLaunchKernelGGL(k_pose_set, dim3(1), dim3(64), 0, 0, f->state, I);
  (void)clear_zbuf(f->zbuf, p->width * p->height, 0);
  (void)clear_zbuf(f->zbuf2, p->width * p->height, 0);
  for (int l = 0; l < DMS_NUM_PYRS; ++l)  // the INACTIVE ("old") prediction is never rendered with loop closure off: no depth anywhere
    (void)hipMemsetD32((hipDeviceptr_t)f->kf_old_dmap[l].data, 0x7fffffff, (size_t)f->kf_old_dmap[l].rows * f->kf_old_dmap[l].cols);  // kept empty from here on: every resolve pass clears what it reads
  (void)hipDeviceSynchronize();
  memset(f->h_state, 0, 4 * sizeof(FrameState));
  g_live_contexts += 1;
  *out = f;
  return DMS_OK;
}

int dms_fusion_destroy(dms_fusion* f) {
  if (!f) return DMS_OK;
  g_live_contexts -= 1;
  (void)hipDeviceSynchronize();
  drain(f);
  for (auto e : f->pool) (void)hipEventDestroy(e);
  for (int k = 0; k < 2; ++k)
    if (f->ev_prep_done[k]) (void)hipEventDestroy(f->ev_prep_done[k]);
  for (int k = 0; k < 4; ++k)
    if (f->ev_main_done[k]) (void)hipEventDestroy(f->ev_main_done[k]);
  if (f->ev_inputs) (void)hipEventDestroy(f->ev_inputs);
  if (f->s_prep) (void)hipStreamDestroy(f->s_prep);
  if (f->arena) (void)hipFree(f->arena);
  if (f->h_state) (void)hipHostFree(f->h_state);
  if (f->nid_host) (void)hipHostFree(f->nid_host);
  if (f->h_loop) (void)hipHostFree(f->h_loop);
  if (f->h_gloop) (void)hipHostFree(f->h_gloop);
  if (f->odom_m2m) dms_odometry_destroy(f->odom_m2m);
  dms_odometry_destroy(f->odom);
  dms_model* const own = f->own_model ? f->own_model : f->model;
  if (f->model && f->model != own) f->model->sharers -= 1;  // (a joined camera: the map's owner outlives it, dmslam_fusion.h)
  for (auto& kv : f->clusters)
    if (kv.second && kv.second != own) dms_model_destroy(kv.second);
  dms_model_destroy(own);
  delete f;
  return DMS_OK;
}

// Context::computeFeedbackBuffers (Context.h:211-223): keeps the current frame's colour and metric depths and the tick
static int snapshot_feedback(dms_fusion* f, hipStream_t s) {
  const size_t N = (size_t)f->p.width * f->p.height;
  DMS_HIP(hipMemcpyAsync(f->fb_rgba.data, f->rgba.data, N * 4, hipMemcpyDeviceToDevice, s));
  DMS_HIP(hipMemcpyAsync(f->fb_dm.data, f->depth_metric.data, N * 4, hipMemcpyDeviceToDevice, s));
  DMS_HIP(hipMemcpyAsync(f->fb_dmf.data, f->depth_metric_filtered.data, N * 4, hipMemcpyDeviceToDevice, s));
  f->fb_time = f->tick;
  f->fb_valid = true;
  return DMS_OK;
}

int dms_fusion_compute_feedback(dms_fusion* f, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(!f->in_frame && f->frames > 0, "computeFeedbackBuffers: between frames, after the first");
  return snapshot_feedback(f, (hipStream_t)st);
}

int dms_fusion_set_cluster(dms_fusion* f, int cluster) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(!f->in_frame, "between frames only");
  f->req_cluster = cluster;
  return DMS_OK;
}

int dms_fusion_clusters(dms_fusion* f, int* ids, int max_ids, int* n_ids, int* current) {
  DMS_REQUIRE(f && n_ids && (ids || max_ids == 0), "null argument");
  int n = 0;
  for (auto& kv : f->clusters) {
    if (n < max_ids) ids[n] = kv.first;
    ++n;
  }
  *n_ids = n;
  if (current) *current = f->cur_cluster;
  return DMS_OK;
}

dms_model* dms_fusion_cluster_model(dms_fusion* f, int cluster) {
  if (!f) return nullptr;
  auto it = f->clusters.find(cluster);
  return it == f->clusters.end() ? nullptr : it->second;
}

// GlobalModel::initialise for an id the map does not know (GlobalModel.cpp:266-398), from the fusion block of a later frame
static int cluster_initialise(dms_fusion* f, int cluster, hipStream_t s) {
  DMS_REQUIRE(f->model == f->own_model && f->model->sharers == 1, "a new cluster on a map that several cameras share is not supported");
  DMS_REQUIRE(f->fb_valid, "no feedback buffers");
  dms_model* nm = nullptr;
  int rc = dms_model_create(&nm, f->p.model_capacity, f->p.width, f->p.height);
  if (rc) return rc;
  (void)dms_model_set_num_sensors(nm, f->p.num_sensors);
  rc = model_initialise(nm, &f->fb_rgba, &f->fb_dm, &f->fb_dmf, &f->cam, f->fb_time, f->p.timeIdx, (float)(int)f->p.maxDepthProcessed, s);
  if (rc) {
    dms_model_destroy(nm);
    return rc;
  }
  nm->count_hold = 4;  // the result slots still in flight count the cluster that was current when their frames ran
  nm->last_writer = f;
  f->clusters[cluster] = nm;
  f->cur_cluster = cluster;
  f->model = f->own_model = nm;
  f->pre_tick = -1;  // (a prediction rendered ahead of time was of the other cluster)
  return DMS_OK;
}

int dms_fusion_inputs_ready(dms_fusion* f, dms_stream producer) {
  DMS_REQUIRE(f, "null argument");
  if (!f->p.pipeline_ingest) return DMS_OK;  // everything runs on the caller's stream: plain stream order applies
  DMS_HIP(hipEventRecord(f->ev_inputs, (hipStream_t)producer));
  f->inputs_armed = true;
  return DMS_OK;
}

int dms_fusion_inputs_consumed(dms_fusion* f, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  if (f->p.pipeline_ingest) {
    if (f->last_prep >= 0) DMS_HIP(hipEventSynchronize(f->ev_prep_done[f->last_prep]));
  } else {
    DMS_HIP(hipStreamSynchronize((hipStream_t)st));
  }
  return DMS_OK;
}

dms_model* dms_fusion_model(dms_fusion* f) { return f ? f->model : nullptr; }
const float* dms_fusion_pose_device(dms_fusion* f) { return f ? f->state->cur.pose : nullptr; }
dms_odometry* dms_fusion_odometry(dms_fusion* f) { return f ? f->odom : nullptr; }

int dms_fusion_process_frame_begin(dms_fusion* f, const void* rgb_dev, int rgb_channels, const unsigned short* depth_dev,
                                   const float* inPose16, float weightMultiplier, dms_stream st) {
  DMS_REQUIRE(f && rgb_dev && depth_dev, "null argument");
  DMS_REQUIRE(rgb_channels == 3 || rgb_channels == 4, "rgb_channels must be 3 or 4");
  DMS_REQUIRE(!f->in_frame, "dms_fusion_process_frame_end has not been called for the previous frame");
  f->armed_now = f->armed;  // (the arming holds for this frame only, whatever path it takes)
  f->armed = dms_fusion::ArmedBlock();
  f->armed_written = false;
  hipStream_t s = (hipStream_t)st;
  const int W = f->p.width, H = f->p.height, N = W * H;
  int rc;
  // the ORB loop poses armed by dms_fusion_set_orb_loop belong to THIS call only, as the reference's hybrid_loops block uses only the
  // pointers of the processFrame call it runs in (ElasticFusion.cpp:292-350, inside the not-first-frame branch): a bootstrap
  // frame or a begin that fails further down must not leave them armed for a later frame
  const bool orb_now = f->orb_armed;
  f->orb_armed = false;
  f->gloop_ran = false;
  // a previous frame that failed between its fuse and the index map that applies the fuse's update left the map with a pending
  // pass: apply it now instead of failing every later frame
  if ((rc = model_flush_pending(f->model, s))) return rc;

  // ---- live half: everything that depends on the incoming frame only -------------------------
  // Runs on the prep stream into image set frames%2 and odometry ring set frames%3, so it
  // overlaps the previous frame's tracking / fusion on the caller's stream.  Hazards: these sets
  // were last read by frame-2 (image set; ring set as lastNextImage), hence the wait on its
  // completion event.  With pipeline_ingest = 0 the same work is issued on the caller's stream.
  // Which ring set holds lastNextImage: the reference hands this frame's intensity pyramid on to the next call only when the
  // tracker ran with so3 (`if (so3) swap(lastNextImage[i], nextImage[i])`, RGBDOdometry.cpp:594-600; the first frame's comes
  // from initFirstRGB, ElasticFusion.cpp:151) — with the GUI's so3 switch off for a while (dms_fusion_set_option) the SO3
  // pre-alignment of the frame that turns it on again compares against the last pyramid that WAS handed on.  In flight are the
  // previous frame's live set and its lastNextImage set; this frame takes the third.  With so3 always on: sets t % 3.
  const int k2 = (int)(f->frames % 2);
  int k3 = 0, k3prev = 0;
  if (f->frames > 0) {
    k3prev = f->prev_handed_on ? f->live_set : f->lastnext_set;
    k3 = 0;
    while (k3 == f->live_set || k3 == f->lastnext_set) ++k3;
  }
  f->live_set = k3;
  f->lastnext_set = k3prev;
  f->prev_handed_on = false;  // set where the tracker is enqueued / the first frame is initialised
  hipStream_t sp = f->p.pipeline_ingest ? f->s_prep : s;
  if (f->p.pipeline_ingest) {
    // The ingest below reads rgb_dev / depth_dev on the prep stream.  Inputs produced asynchronously (an async upload, a
    // decode kernel) are declared with dms_fusion_inputs_ready(f, producer_stream): the prep stream then waits for that
    // point of the producer stream.  (Waiting on the caller's stream `s` unconditionally would put this frame's ingest
    // behind the previous frame's tracking and fusion, which is exactly the overlap the prep stream exists for.)
    if (f->inputs_armed) {
      DMS_HIP(hipStreamWaitEvent(sp, f->ev_inputs, 0));
      f->inputs_armed = false;
    }
    // Bounded run-ahead: the host blocks here until frame t - host_lag has finished.  This keeps the HIP command queues
    // short (with the host many frames ahead the runtime's queue-full handling was measured to cost ~10 % throughput,
    // DESIGN.md §6).  The buffers themselves are protected on the device: this frame's live half reuses the image set and
    // the tracker ring set that frame t-2 read, so the prep stream waits for THAT frame's completion event.  (host_lag = 3
    // has the live half enqueued before frame t-2 ends, so that it starts at once instead of ~60 us into frame t-1:
    // measured 2.4 % slower — the deeper queues cost more than the earlier start returns; default 2.)
    {
      const auto t0 = std::chrono::steady_clock::now();
      DMS_HIP(hipEventSynchronize(f->ev_main_done[(f->frames + 4 - f->host_lag) % 4]));  // (never recorded: returns at once)
      const double waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      f->host_wait_ms += waited_ms;
      // the late frame's two rules (see `late_main`)
      const bool alone = f->late_owner >= 0 ? f->late_owner != 0 : g_live_contexts.load() == 1;
      const bool may = alone && f->model->sharers == 1 && f->map_initialised && f->frames >= 2;
      if (f->late_forced >= 0) {
        f->late_main = f->late_forced != 0;
      } else if (!may) {
        f->late_main = false;
        f->late_low = 0;
      } else if (f->late_cool > 0) {
        f->late_cool -= 1;
        f->late_main = false;
      } else {
        if (f->late_main && waited_ms < 0.015) {
          if (++f->late_low >= 3) {  // the host has no slack left: back to the barrier for a while
            f->late_main = false;
            f->late_low = 0;
            f->late_cool = f->late_backoff;
            if (f->late_backoff < 4096) f->late_backoff *= 2;
          }
        } else {
          f->late_low = 0;
          f->late_main = true;
        }
      }
    }
    DMS_HIP(hipStreamWaitEvent(sp, f->ev_main_done[(f->frames + 2) % 4], 0));  // frame t-2
  }
  // A completed frame's result block (pinned ring slot) tightens the host-side bound of the surfel count that launch grids
  // are sized from; every clean since then adds at most `slots` surfels.  Without this the bound grows by `slots` per
  // frame until it reaches the capacity.
  f->model->last_writer = f;
  if (f->model->count_hold > 0) {
    f->model->count_hold -= 1;
  } else if (f->map_initialised && f->model->sharers == 1) {  // (a shared map: another camera's cleans are not counted by this context's frames)
    // newest frame whose result slot is certainly readable: t - host_lag with the pipeline (just waited for), else t - 2 if done
    const int lag = f->p.pipeline_ingest ? f->host_lag : 2;
    const int slot = (int)((f->frames + 4 - lag) % 4);
    if (f->frames >= lag && (f->p.pipeline_ingest || hipEventQuery(f->ev_main_done[slot]) == hipSuccess)) {
      const size_t known = (size_t)f->h_state[slot].surfels + (size_t)(lag - 1) * (size_t)f->model->slots;  // one clean per later frame
      if (known < f->model->count_upper) f->model->count_upper = known;
    }
  }
  f->rgba = f->live[k2].rgba;
  f->depth_raw = f->live[k2].depth_raw;
  f->depth_filtered = f->live[k2].depth_filtered;
  f->depth_metric = f->live[k2].depth_metric;
  f->depth_metric_filtered = f->live[k2].depth_metric_filtered;
  odometry_bind_live(f->odom, k3);
  odometry_bind_lastnext(f->odom, k3prev);
  const bool want_live = f->p.hybrid_tracking || f->frames == 0;  // the tracker's live pyramids (at the first frame: what initFirstRGB computes, :151)
  if (f->fused_live) {
    // The live half in three launches (prep.hip "Fused live half"): ingest (+ raw metric depth + level-0 intensity), the depth
    // filter (+ filtered metric depth + level-0 depth and vertex map in its epilogue), and one kernel for everything else of the
    // three pyramid levels.  Same bits as the operator chain below (GPU test).
    dms_image2d ld[3], lv[3], ln[3], li[3], ldx[3], ldy[3], lg[3];
    float minScale[3];
    odometry_live_views(f->odom, ld, lv, ln, li, ldx, ldy, lg, minScale, want_live);
    {
      FTimer t(f, sp, "ingest");
      hipLaunchKernelGGL(k_live_ingest, dim3(min((N + 255) / 256, 2048)), dim3(256), 0, sp, (const unsigned char*)rgb_dev, rgb_channels,
                         depth_dev, (uchar4*)f->rgba.data, (unsigned short*)f->depth_raw.data, (float*)f->depth_metric.data,
                         want_live ? (unsigned char*)li[0].data : (unsigned char*)nullptr, N, f->p.depthCut);
      DMS_CHECK_LAUNCH();
    }
    {  // filterDepth + metriciseDepth (ElasticFusion.cpp:118-119)
      FTimer t(f, sp, "preprocess");
      bool beside_tracker = f->p.pipeline_ingest != 0;
      if (beside_tracker && hipStreamQuery(s) == hipSuccess) beside_tracker = false;  // (see the operator chain below)
      (void)hipGetLastError();
      if ((rc = depth_bilateral(&f->depth_raw, &f->depth_filtered, f->p.depthCut, sp, beside_tracker ? f->prep_blocks : 0, &f->depth_metric_filtered,
                                want_live ? &ld[0] : nullptr, want_live ? &lv[0] : nullptr, &f->cam, f->p.maxDepthProcessed)))
        return rc;
    }
    if (want_live) {
      FTimer t(f, sp, "live_pyramids");
      if ((rc = liveLevelsFused(ld, lv, ln, li, ldx, ldy, lg, &f->cam, f->p.maxDepthProcessed, minScale, sp))) return rc;
    }
  } else {
    // "upload": the frame is already in HBM; bring it into the context's textures (ElasticFusion.cpp:111-114)
    {
      FTimer t(f, sp, "ingest");
      if (rgb_channels == 3)
        hipLaunchKernelGGL(k_rgb_to_rgba, dim3(min((N + 255) / 256, 2048)), dim3(256), 0, sp, (const unsigned char*)rgb_dev,
                           (uchar4*)f->rgba.data, N);
      else
        DMS_HIP(hipMemcpyAsync(f->rgba.data, rgb_dev, (size_t)N * 4, hipMemcpyDeviceToDevice, sp));
      DMS_CHECK_LAUNCH();
      DMS_HIP(hipMemcpyAsync(f->depth_raw.data, depth_dev, (size_t)N * 2, hipMemcpyDeviceToDevice, sp));
    }
    {  // filterDepth + metriciseDepth (ElasticFusion.cpp:118-119)
      FTimer t(f, sp, "preprocess");
      // (fat blocks only while the caller's stream is busy: they exist to leave the previous frame's tracker its compute
      // units; when nothing is running there — the first frame after a pause — the whole-chip form is 2.5x shorter and this
      // frame's tracker is waiting for it)
      bool beside_tracker = f->p.pipeline_ingest != 0;
      if (beside_tracker && hipStreamQuery(s) == hipSuccess) beside_tracker = false;
      (void)hipGetLastError();  // (hipErrorNotReady is the expected answer)
      if ((rc = depth_bilateral(&f->depth_raw, &f->depth_filtered, f->p.depthCut, sp, beside_tracker ? f->prep_blocks : 0))) return rc;
      if ((rc = depth_metric(&f->depth_raw, &f->depth_metric, f->p.depthCut, sp))) return rc;
      if ((rc = depth_metric(&f->depth_filtered, &f->depth_metric_filtered, f->p.depthCut, sp))) return rc;
    }
    {  // live half of frameToModel.initICP / initRGB (ElasticFusion.cpp:190-194): depth pyramid, vertex / normal
       // maps, intensity pyramid and its derivatives.  At the first frame the intensity pyramid is
       // what initFirstRGB computes (ElasticFusion.cpp:151); it becomes lastNextImage of frame 1.
      FTimer t(f, sp, "live_pyramids");
      if (f->p.hybrid_tracking || f->frames == 0) {
        if ((rc = dms_odometry_initICP_depth(f->odom, &f->depth_filtered, f->p.maxDepthProcessed, sp))) return rc;
        if ((rc = odometry_initRGB_image(f->odom, &f->rgba, sp))) return rc;
      }
    }
  }
  if (f->p.pipeline_ingest) {
    DMS_HIP(hipEventRecord(f->ev_prep_done[k2], sp));
    if (f->late_main) {
      // the HOST waits for the live half and enqueues the frame behind it: no barrier packet on the frame's queue
      const auto t0 = std::chrono::steady_clock::now();
      DMS_HIP(hipEventSynchronize(f->ev_prep_done[k2]));
      f->host_wait_prep_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else {
      DMS_HIP(hipStreamWaitEvent(s, f->ev_prep_done[k2], 0));
    }
    f->last_prep = k2;
  }

  Pose16 prior;
  memset(&prior, 0, sizeof(prior));
  if (inPose16) memcpy(prior.v, inPose16, sizeof(prior.v));

  f->cur_k2 = k2;
  f->cur_bootstrap = !f->map_initialised;
  f->cur_fuse_now = true;
  f->tracking_ok = true;
  if (!f->map_initialised) {
    // first run (ElasticFusion.cpp:132-152): surfels from this frame, pose = inPose or identity
    if (!inPose16)
      for (int i = 0; i < 16; ++i) prior.v[i] = (i % 5 == 0) ? 1.f : 0.f;
    hipLaunchKernelGGL(k_pose_set, dim3(1), dim3(64), 0, s, f->state, prior);
    DMS_CHECK_LAUNCH();
    if (!f->adopting) {  // (an imported camera joins a map that exists: this frame only seeds its live state)
      FTimer t(f, s, "initialise");
      // computeFeedbackBuffers takes `const int& maxDepthProcessed` (Context.h:211): 25.0f -> 25
      if ((rc = model_initialise(f->model, &f->rgba, &f->depth_metric, &f->depth_metric_filtered, &f->cam, f->tick, f->p.timeIdx,
                                 (float)(int)f->p.maxDepthProcessed, s)))
        return rc;
      if ((rc = snapshot_feedback(f, s))) return rc;
      if (f->req_cluster != f->cur_cluster && !f->clusters.count(f->req_cluster) && f->model == f->own_model) {
        // initialise(..., cluster, pose) of the first frame under another id: the surfels go into buffers of their own and the
        // constructor's cluster stays behind, empty
        f->clusters[f->cur_cluster] = nullptr;
        f->clusters[f->req_cluster] = f->model;
        f->cur_cluster = f->req_cluster;
      }
    }
    // initFirstRGB (ElasticFusion.cpp:151): the intensity pyramid of this frame already sits in ring set 0
    f->map_initialised = true;
    f->prev_handed_on = true;
  } else {
    if (inPose16) {  // without a prior the pose block is already consistent: the previous frame left lastPose = pose and its inverse
      hipLaunchKernelGGL(k_frame_begin, dim3(1), dim3(64), 0, s, f->state, prior, 1);
      DMS_CHECK_LAUNCH();
    }
    // ElasticFusion.cpp:165-167: an extra block of this predict's fill-in launch takes the denseEnough decision
    // (round 6) the set-up of this frame's tracker call rides on the prediction's resolve pass - a kernel boundary before the model
    // pyramid launch, which can then run the tracker's SO3 stage beside the pyramid (track.hip: k_so3_model)
    const TrackFold fold = {f->state->cur.pose, f->p.pyramid, f->p.fastOdom, f->p.so3, 0};
    TrackInitArgs early;
    memset(&early, 0, sizeof(early));
    bool early_taken = false;
    const bool want_early = f->p.hybrid_tracking && f->fold_track_init && !f->lost && odometry_early_init_args(f->odom, &fold, &early) != 0;
    if ((rc = predict(f, 0.7f, s, nullptr, true, 2, inPose16 != nullptr, want_early ? &early : nullptr, &early_taken))) return rc;
    if (want_early && !early_taken) odometry_early_init_args(f->odom, nullptr, &early);  // (no fused resolve pass this frame: the set-up folds into the pyramid kernel as before)
    if (f->p.hybrid_tracking) {
      {
        FTimer t(f, s, "odom_init");
        // WARNING (reference): initICP* must be called before initRGB* (ElasticFusion.cpp:172)
        // (the set-up of the tracker call below rides on the pyramid kernel: no launch of its own, DMS_FOLD_TRACK_INIT=0 to compare)
        if ((rc = odometry_initModel_fused(f->odom, f->pred.vertex.data, f->pred.normal.data, f->pred.image.data, f->fill.vertex.data,
                                           f->fill.normal.data, f->fill.image.data, &f->state->fill_in, f->p.frameToFrameRGB ? 1 : 0,
                                           f->state->cur.pose, s, 1,  // (last pyramid step: inside the tracker's first kernel, below)
                                           f->dense_by_counters ? f->tickets : nullptr, (f->p.width / 20) * (f->p.height / 20),
                                           f->fold_track_init ? &fold : nullptr)))
          return rc;
        // initICP / initRGB: the live half ran on the prep stream; nextDepth = lastDepth (same source)
        odometry_alias_next_depth(f->odom);
      }
      {
        FTimer t(f, s, "track");
        if ((rc = odometry_track_enqueue(f->odom, nullptr, nullptr, f->state->cur.pose, f->p.rgbOnly, f->p.icpWeight, f->p.pyramid,
                                         f->p.fastOdom, f->p.so3, 0, s, f->state, weightMultiplier)))
          return rc;
        f->prev_handed_on = f->p.so3 != 0;
        // (the tracker's finalize kernel writes the new pose straight back into f->state->cur.pose)
      }
      if (f->p.reloc) {  // ElasticFusion.cpp:204-244; lastFrameRecovery is only ever set by the compiled-out fern block
        dms_track_result tr;
        if ((rc = dms_odometry_fetch_result(f->odom, &tr, s))) return rc;  // synchronises
        f->tracking_ok = (double)tr.lastICPError < 1e-04;
        if (!f->lost) {
          double cov[36];
          if ((rc = dms_odometry_getCovariance(f->odom, cov))) return rc;
          for (int i = 0; i < 6; ++i)
            if (cov[i * 7] > 1e-04) {
              f->tracking_ok = false;
              break;
            }
          if (!f->tracking_ok) {
            f->tracking_count += 1;
            if (f->tracking_count > 10) f->lost = true;
          } else {
            f->tracking_count = 0;
          }
        }
      }
    }
    if (!f->p.hybrid_tracking) {  // with tracking on, the tracker's last kernel does this
      hipLaunchKernelGGL(k_frame_after_track, dim3(1), dim3(64), 0, s, f->state, weightMultiplier);
      DMS_CHECK_LAUNCH();
    }
    // "GlobalPredict" (ElasticFusion.cpp:273): its consumers are the NID key-framing gate below and the
    // fern / loop-closure blocks, which this reference compiles out with `if (false)` (ElasticFusion.cpp:279,
    // :593); the final predict overwrites every image it writes before the frame returns.
    bool fuse_now = true;
    if (f->p.global_predict || f->p.nid_keyframing || f->p.local_loop_closure)
      if ((rc = predict(f, f->p.confidence, s))) return rc;
    if (orb_now) {
      // hybrid_loops && orbTcwOld && orbTcwNew (ElasticFusion.cpp:292-350): the constraints of the ORB loop closure; the caller's
      // Deformation::constrain (:337) decides between dms_fusion_fetch_loop and _end.  The block ends with predict(context, rf)
      // (:349), which restores the current view for whoever reads it next in this frame
      if ((rc = global_loop_device(f, f->orb_old, f->orb_new, 0, s))) return rc;
      if ((rc = predict(f, f->p.confidence, s))) return rc;  // unconditional, as :349
    }
    if (f->p.local_loop_closure && !f->lost) {
      // closeLoops without a fern match (ElasticFusion.cpp:399-497): the camera is never lost and
      // rawGraph is empty (nothing deforms the map inside this library)
      {
        FTimer t(f, s, "predict_old");  // combinedPredict(..., 0, id, tick - timeDelta, timeDelta, INACTIVE) (:403-406)
        if ((rc = splat_predict(f->model, &f->state->cur, &f->cam, f->p.maxDepthProcessed, f->p.confidence, 0, f->p.timeIdx,
                                f->tick - f->p.timeDelta, f->p.timeDelta, 0, f->zbuf, &f->pred_old, nullptr, 1, s)))
          return rc;
      }
      {
        FTimer t(f, s, "loop_init");  // modelToModel().initICPModel / initRGBModel / initICP / initRGB (:409-418)
        if ((rc = odometry_initModel_fused(f->odom_m2m, f->pred_old.vertex.data, f->pred_old.normal.data, f->pred_old.image.data,
                                           f->pred_old.vertex.data, f->pred_old.normal.data, f->pred_old.image.data, &f->state->fill_in, 0,
                                           f->state->cur.pose, s)))
          return rc;
        if ((rc = odometry_initLive_fused(f->odom_m2m, f->pred.vertex.data, f->pred.normal.data, f->pred.image.data, &f->state->fill_in, s)))
          return rc;
      }
      {
        FTimer t(f, s, "loop_track");  // getIncrementalTransformation(trans, rot, false, 10, pyramid, fastOdom, false) (:424-425)
        if ((rc = odometry_track_enqueue(f->odom_m2m, nullptr, nullptr, f->state->cur.pose, 0, 10.f, f->p.pyramid, f->p.fastOdom, 0, 0, s,
                                         nullptr, 1.f)))
          return rc;
      }
      // covariance + acceptance test + constraint sampling (:427-474), all on device
      if ((rc = odometry_loop_candidate(f->odom_m2m, f->state, &f->pred.vertex, &f->pred_old.time, f->p.maxDepthProcessed, f->loop,
                                        (float*)((char*)f->loop + up256(sizeof(LoopState))), s)))
        return rc;
    }
    if (f->p.nid_keyframing) {
      // ElasticFusion::fuseFrame (ElasticFusion.cpp:639-677): candidate key frame = this prediction
      // (KeyFrame.h:83-172: intensity of the image, verticesToDepth of the vertex map), reduced
      // nid_pyramid_level times (MutualInformation.cpp:169-174), scored against the live pyramids
      FTimer t(f, s, "nid");
      const int L = f->p.nid_pyramid_level;
      if (f->p.local_loop_closure) {
        // with a rendered INACTIVE view the key frame's "old" half is real (KeyFrame.h:139-166; the
        // reference copies the old vertex texture with the byte count of a released array, :147-150,
        // so its old depth map is undefined — the evident intent is implemented)
        if ((rc = imageToIntensity(&f->pred_old.image, &f->kf_old_img[0], s))) return rc;
        if ((rc = verticesToDepth((const float*)f->pred_old.vertex.data, &f->kf_old_dmap[0], f->p.maxDepthProcessed, s))) return rc;
        for (int l = 1; l <= L; ++l) {
          if ((rc = pyrDownUcharGauss(&f->kf_old_img[l - 1], &f->kf_old_img[l], s))) return rc;
          if ((rc = pyrDownGaussF(&f->kf_old_dmap[l - 1], &f->kf_old_dmap[l], s))) return rc;
        }
      }
      if ((rc = imageToIntensity(&f->pred.image, &f->kf_img[0], s))) return rc;
      if ((rc = verticesToDepth((const float*)f->pred.vertex.data, &f->kf_dmap[0], f->p.maxDepthProcessed, s))) return rc;
      for (int l = 1; l <= L; ++l) {
        if ((rc = pyrDownUcharGauss(&f->kf_img[l - 1], &f->kf_img[l], s))) return rc;
        if ((rc = pyrDownGaussF(&f->kf_dmap[l - 1], &f->kf_dmap[l], s))) return rc;
      }
      dms_image2d nextImg, nextD;
      if ((rc = odometry_next_buffers(f->odom, L, &nextImg, &nextD))) return rc;
      if ((rc = computeNIDImg(&f->kf_img[L], &f->kf_old_img[L], &f->kf_dmap[L], &f->kf_old_dmap[L], &nextImg, f->p.nid_bins_img, f->nid_ws,
                              f->nid_ws_bytes, f->nid_host, nullptr, s)))
        return rc;
      if ((rc = computeNIDDepth(&f->kf_dmap[L], &f->kf_old_dmap[L], &nextD, f->p.nid_bins_depth, f->p.maxDepthProcessed * 1000.0f, f->nid_ws,
                                f->nid_ws_bytes, f->nid_host + 1, nullptr, s)))
        return rc;  // (both calls synchronise: the decision is taken on the host, as in the reference)
      f->last_nid = (f->p.nid_depth_lambda * f->nid_host[1]) + ((1.0f - f->p.nid_depth_lambda) * f->nid_host[0]);
      fuse_now = f->last_nid > f->p.nid_threshold;
    } else {
      f->last_nid = 0.f;
    }
    f->cur_fuse_now = fuse_now;
    if (f->p.local_loop_closure && !f->lost) {
      // the candidate is readable (dms_fusion_fetch_loop) before the second half is enqueued
      const int k4 = (int)(f->frames % 4);  // this frame's result slot (the final prediction mirrors into the same one)
      DMS_HIP(hipMemcpyAsync(f->h_state + k4, f->state, sizeof(FrameState), hipMemcpyDeviceToHost, s));
      DMS_HIP(hipMemcpyAsync(f->h_loop + (size_t)k4 * f->loop_bytes, f->loop, f->loop_bytes, hipMemcpyDeviceToHost, s));
      f->last_slot = k4;
    }
  }
  f->in_frame = true;
  return DMS_OK;
}

int dms_fusion_process_frame_end(dms_fusion* f, const float* graph_host, int graph_nodes, const float* newPose16, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(f->in_frame, "dms_fusion_process_frame_begin has not been called");
  DMS_REQUIRE(graph_nodes == 0 || graph_host, "null graph");
  hipStream_t s = (hipStream_t)st;
  int rc;
  int fused = 0;
  bool surfels_written = false;
  f->in_frame = false;
  if (f->cur_bootstrap) {
    fused = f->adopting ? 0 : 1;
    f->adopting = false;
  } else {
    if (newPose16) {
      Pose16 np;
      memcpy(np.v, newPose16, sizeof(np.v));
      hipLaunchKernelGGL(k_pose_override, dim3(1), dim3(64), 0, s, f->state, np);
      DMS_CHECK_LAUNCH();
    }
    // fuseFrame(context, rawGraph.size() > 0): a deforming frame always fuses (ElasticFusion.cpp:641-644)
    const bool fuse_now = f->cur_fuse_now || graph_nodes > 0;
    if (graph_nodes > 0) f->last_nid = 0.f;
    const int timeDeltaEff = f->p.timeDelta + f->frames_since_fusion;  // ElasticFusion.cpp:518,541,563

    if (!f->p.rgbOnly && f->tracking_ok && !f->lost && fuse_now) {  // fusion (ElasticFusion.cpp:506-564)
      if (!f->clusters.count(f->req_cluster) && (rc = cluster_initialise(f, f->req_cluster, s))) return rc;  // :508-515
      {
        FTimer t(f, s, "index_map");
        if ((rc = index_map(f->model, &f->state->cur, &f->cam, f->tick, f->p.timeIdx, f->p.maxDepthProcessed, timeDeltaEff, f->zbuf,
                            &f->imap, 1, 1, s)))
          return rc;
      }
      {
        FTimer t(f, s, "fuse");
        if ((rc = model_fuse(f->model, &f->state->cur, f->tick, f->p.timeIdx, &f->rgba, &f->depth_metric, &f->depth_metric_filtered,
                             &f->imap, &f->cam, f->p.maxDepthProcessed, 1.f, &f->state->weighting, 1, s, 1)))  // (update pass: inside the next index map)
          return rc;
      }
      {
        FTimer t(f, s, "index_map");
        if ((rc = index_map(f->model, &f->state->cur, &f->cam, f->tick, f->p.timeIdx, f->p.maxDepthProcessed, timeDeltaEff, f->zbuf,
                            &f->imap, 1, 1, s)))
          return rc;
      }
      if (graph_nodes > 0) {
        // a deformation is a second pose update this frame: predict the depth again to decide whose
        // time stamps to refresh (ElasticFusion.cpp:541-553; synthesizeDepth leaves `actv` false)
        FTimer t(f, s, "synth_depth");
        if ((rc = splat_predict(f->model, &f->state->cur, &f->cam, f->p.maxDepthProcessed, f->p.confidence, f->tick, f->p.timeIdx,
                                f->tick - timeDeltaEff, 65535, 0, f->zbuf, nullptr, &f->depth_synth, 1, s)))
          return rc;
      }
      {
        FTimer t(f, s, "clean");
        if ((rc = model_clean(f->model, &f->state->cur, f->tick, f->p.timeIdx, &f->imap, graph_nodes > 0 ? &f->depth_synth : nullptr, &f->cam,
                              f->p.confidence, graph_host, graph_nodes, timeDeltaEff, f->p.maxDepthProcessed, 0, 1, &f->state->surfels, s)))
          return rc;
        surfels_written = true;  // the clean's scan also stores the new count into the result block
      }
      fused = 1;
    }
    f->frames_since_fusion = fuse_now ? 0 : f->frames_since_fusion + 1;  // ElasticFusion.cpp:567-568
  }
  if (!surfels_written) {
    hipLaunchKernelGGL(k_frame_end, dim3(1), dim3(64), 0, s, f->state, f->model->d_count);
    DMS_CHECK_LAUNCH();
  }
  // finalPredict (ElasticFusion.cpp:586); its fill-in kernel mirrors the result block to the host slot
  const int k4 = (int)(f->frames % 4);
  if ((rc = predict(f, f->p.confidence, s, f->h_state_dev + k4, false, 1))) return rc;
  f->last_slot = k4;
  DMS_HIP(hipEventRecord(f->ev_main_done[k4], s));  // (also without the pipeline: it gates the reading of this frame's result slot)
  f->fused_last = fused;
  if (!f->lost) f->tick += 1;  // ElasticFusion.cpp:588-591
  f->frames += 1;
  return DMS_OK;
}

int dms_fusion_process_frame(dms_fusion* f, const void* rgb_dev, int rgb_channels, const unsigned short* depth_dev, const float* inPose16,
                             float weightMultiplier, dms_stream st) {
  int rc = dms_fusion_process_frame_begin(f, rgb_dev, rgb_channels, depth_dev, inPose16, weightMultiplier, st);
  if (rc) return rc;
  return dms_fusion_process_frame_end(f, nullptr, 0, nullptr, st);
}

// Sticky tracker-timeout report: FrameState::track_timeouts counts the frames whose resident tracker kernels gave up at
// a grid barrier (those frames kept their prior pose and fused nothing).  Any fetch that sees the counter move says so,
// whichever frame it happened in.
static int report_timeouts(dms_fusion* f, const FrameState* hs) {
  if (hs->track_range_failures != f->range_failures_reported) {
    // not a residency problem: the maps held values no fixed-point range fits (non-finite input); the execution mode stays
    const int n = hs->track_range_failures - f->range_failures_reported;
    f->range_failures_reported = hs->track_range_failures;
    if (hs->track_timeouts == f->timeouts_reported) {
      set_error("dms_fusion: %d frame(s) since the last fetch found no fixed-point range for a cross-pixel sum of the tracker (non-finite "
                "input maps?); those frames kept their prior pose and were not fused", n);
      return DMS_ERR_TIMEOUT;
    }
  }
  if (hs->track_timeouts == f->timeouts_reported) return DMS_OK;
  const int n = hs->track_timeouts - f->timeouts_reported;
  f->timeouts_reported = hs->track_timeouts;
  // resident kernels that cannot be co-resident would time out on every frame: the trackers of this camera run
  // launch-per-phase from here on (same bits, see track.hip)
  (void)dms_odometry_fall_back_to_launches(f->odom);
  if (f->odom_m2m) (void)dms_odometry_fall_back_to_launches(f->odom_m2m);
  set_error("dms_fusion: %d frame(s) since the last fetch had a resident tracker kernel time out at a grid-wide wait (its blocks were not "
            "all on the device: another process on this GPU?); those frames kept their prior pose and were not fused.  "
            "This camera's trackers have switched to launch-per-phase kernels (DMS_TRACK_MODE=launches)",
            n);
  return DMS_ERR_TIMEOUT;
}

static void fill_loop(const dms_fusion* f, dms_frame_result* r) {
  r->tracking_ok = f->tracking_ok ? 1 : 0;
  r->lost = f->lost ? 1 : 0;
  if (!f->h_loop || f->lost) return;
  const LoopState* L = (const LoopState*)(f->h_loop + (size_t)f->last_slot * f->loop_bytes);
  r->loop_ok = L->ok;
  r->loop_constraints = L->n_constraints;
  r->loop_icp_error = L->icp_error;
  r->loop_icp_count = L->icp_count;
  memcpy(r->loop_pose, L->est_pose, sizeof(r->loop_pose));
  memcpy(r->loop_cov_diag, L->cov_diag, sizeof(r->loop_cov_diag));
}

int dms_fusion_fetch_loop(dms_fusion* f, dms_frame_result* r, dms_stream st) {
  DMS_REQUIRE(f && r, "null argument");
  DMS_REQUIRE(f->in_frame && (f->p.local_loop_closure || f->gloop_ran),
              "only between process_frame_begin and _end, with local_loop_closure or after dms_fusion_set_orb_loop");
  memset(r, 0, sizeof(*r));
  DMS_HIP(hipStreamSynchronize((hipStream_t)st));
  if (f->cur_bootstrap) return DMS_OK;
  if (!(f->p.local_loop_closure && !f->lost)) {  // (that branch has mirrored the frame state already)
    const int k4 = (int)(f->frames % 4);
    DMS_HIP(hipMemcpyAsync(f->h_state + k4, f->state, sizeof(FrameState), hipMemcpyDeviceToHost, (hipStream_t)st));
    DMS_HIP(hipStreamSynchronize((hipStream_t)st));
    f->last_slot = k4;
  }
  const FrameState* hs = f->h_state + f->last_slot;
  memcpy(r->pose, hs->cur.pose, sizeof(r->pose));
  r->tick = f->tick;
  r->fill_in = hs->fill_in;
  r->weighting = hs->weighting;
  r->nid_score = f->last_nid;
  fill_loop(f, r);
  return report_timeouts(f, hs);
}

int dms_fusion_fetch(dms_fusion* f, dms_frame_result* r, dms_stream st) {
  DMS_REQUIRE(f && r, "null argument");
  hipStream_t s = (hipStream_t)st;
  memset(r, 0, sizeof(*r));
  int rc = DMS_OK;
  if (f->p.hybrid_tracking && f->tick > 2) {
    rc = dms_odometry_fetch_result(f->odom, &r->track, s);  // syncs
    if (rc && rc != DMS_ERR_TIMEOUT) return rc;  // (a timeout of the last frame is reported below with every other one)
  } else {
    DMS_HIP(hipStreamSynchronize(s));
  }
  drain(f);
  const FrameState* hs = f->h_state + f->last_slot;
  memcpy(r->pose, hs->cur.pose, sizeof(r->pose));
  r->surfels = hs->surfels;
  if (f->model->sharers == 1 || f->model->last_writer == f) f->model->count_upper = r->surfels;
  r->tick = f->tick;
  r->fused = f->fused_last;
  r->fill_in = hs->fill_in;
  r->weighting = hs->weighting;
  r->nid_score = f->last_nid;
  fill_loop(f, r);
  if ((size_t)r->surfels >= f->model->cap) {
    // the reference asserts on this (GlobalModel.cpp:703); here the kernels stop appending at the
    // capacity, nothing is overwritten, and the caller is told (the result above is still valid)
    set_error("dms_fusion_fetch: the map has reached its capacity of %zu surfels; new surfels are being dropped", f->model->cap);
    (void)report_timeouts(f, hs);
    return DMS_ERR_CAPACITY;
  }
  return report_timeouts(f, hs);
}

size_t dms_thumb_block_bytes(int width, int height) { return dms::thumb_block_size((size_t)(width / 8) * (size_t)(height / 8)); }
void dms_thumb_block_offsets(int width, int height, size_t* vertex_offset, size_t* normal_offset) {
  const size_t n = (size_t)(width / 8) * (size_t)(height / 8);
  if (vertex_offset) *vertex_offset = dms::thumb_vertex_off(n);
  if (normal_offset) *normal_offset = dms::thumb_normal_off(n);
}

int dms_fusion_thumbnails(dms_fusion* f, void* block_dev, dms_stream st) {
  DMS_REQUIRE(f && block_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  const int tw = f->p.width / 8, th = f->p.height / 8;
  hipLaunchKernelGGL(k_thumbnails, dim3((tw * th + 255) / 256), dim3(256), 0, (hipStream_t)st, (const uchar4*)f->fill.image.data,
                     (const float4*)f->fill.vertex.data, (const float4*)f->fill.normal.data, f->p.width, f->p.height, tw, th,
                     (unsigned char*)block_dev, (const float*)nullptr, (float*)nullptr, (int*)nullptr, 0);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int dms_fusion_frame_block(dms_fusion* f, void* block_dev, float* pose16_dst_dev, int* tick_dst_dev, int tick, dms_stream st) {
  DMS_REQUIRE(f && block_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  const int tw = f->p.width / 8, th = f->p.height / 8;
  hipLaunchKernelGGL(k_thumbnails, dim3((tw * th + 255) / 256), dim3(256), 0, (hipStream_t)st, (const uchar4*)f->fill.image.data,
                     (const float4*)f->fill.vertex.data, (const float4*)f->fill.normal.data, f->p.width, f->p.height, tw, th,
                     (unsigned char*)block_dev, (const float*)f->state->cur.pose, pose16_dst_dev, tick_dst_dev, tick);
  DMS_CHECK_LAUNCH();
  return DMS_OK;
}

int dms_fusion_arm_frame_block(dms_fusion* f, void* block_dev, float* pose16_dst_dev, int* tick_dst_dev, int tick) {
  DMS_REQUIRE(f && block_dev, "null argument");
  DMS_REQUIRE(((uintptr_t)block_dev & 15) == 0, "thumbnail block must be 16-byte aligned");
  DMS_REQUIRE(!f->in_frame, "between process_frame_begin and _end");
  f->armed.block = (unsigned char*)block_dev;
  f->armed.pose_dst = pose16_dst_dev;
  f->armed.tick_dst = tick_dst_dev;
  f->armed.tick = tick;
  f->armed_written = false;
  return DMS_OK;
}
int dms_fusion_frame_block_written(dms_fusion* f) { return f && f->armed_written ? 1 : 0; }
int dms_fusion_wait_frame_done(dms_fusion* f, dms_stream waiter) {
  DMS_REQUIRE(f && f->last_slot >= 0 && f->frames > 0, "no frame has been enqueued");
  DMS_HIP(hipStreamWaitEvent((hipStream_t)waiter, f->ev_main_done[f->last_slot], 0));
  return DMS_OK;
}

int dms_fusion_get_loop_constraints(dms_fusion* f, float* rows7_host, int max_rows, int* n) {
  DMS_REQUIRE(f && n && (rows7_host || max_rows == 0), "null argument");
  *n = 0;
  if (!f->h_loop) return DMS_OK;
  const char* slot = f->h_loop + (size_t)f->last_slot * f->loop_bytes;
  const LoopState* L = (const LoopState*)slot;
  const float* c = (const float*)(slot + up256(sizeof(LoopState)));
  *n = L->n_constraints;
  for (int i = 0; i < L->n_constraints && i < max_rows; ++i) memcpy(rows7_host + (size_t)i * 7, c + (size_t)i * 8, 7 * sizeof(float));
  return DMS_OK;
}

int dms_fusion_set_orb_loop(dms_fusion* f, const float* orbTcwOld16, const float* orbTcwNew16) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(f->p.hybrid_loops, "the context was created without hybrid_loops");
  DMS_REQUIRE(!f->in_frame, "between process_frame_begin and _end");
  f->orb_armed = orbTcwOld16 && orbTcwNew16;
  if (f->orb_armed) {
    memcpy(f->orb_old, orbTcwOld16, sizeof(f->orb_old));
    memcpy(f->orb_new, orbTcwNew16, sizeof(f->orb_new));
  }
  return DMS_OK;
}

int dms_fusion_get_global_loop_constraints(dms_fusion* f, float* rows7_host, int max_rows, int* n, dms_stream st) {
  DMS_REQUIRE(f && n && (rows7_host || max_rows == 0), "null argument");
  *n = 0;
  if (!f->h_gloop || !f->gloop_ran) return DMS_OK;
  DMS_HIP(hipStreamSynchronize((hipStream_t)st));
  const int cnt = *(const int*)f->h_gloop;
  const float* c = (const float*)(f->h_gloop + 256);
  *n = cnt;
  for (int i = 0; i < cnt && i < max_rows; ++i) memcpy(rows7_host + (size_t)i * 7, c + (size_t)i * 8, 7 * sizeof(float));
  return DMS_OK;
}

// ElasticFusion::applyGlobalLoop (ElasticFusion.cpp:1148-1240) in two halves around the caller's Deformation::constrain
int dms_fusion_apply_global_loop_begin(dms_fusion* f, const float* orbTcwOld16, const float* orbTcwNew16, dms_stream st) {
  DMS_REQUIRE(f && orbTcwOld16 && orbTcwNew16, "null argument");
  DMS_REQUIRE(f->p.hybrid_loops, "the context was created without hybrid_loops");
  DMS_REQUIRE(!f->in_frame && !f->in_global_loop && f->map_initialised, "needs an initialised map, outside a frame");
  int rc = global_loop_device(f, orbTcwOld16, orbTcwNew16, 1, (hipStream_t)st);
  if (rc) return rc;
  f->in_global_loop = true;
  return DMS_OK;
}

int dms_fusion_apply_global_loop_end(dms_fusion* f, const float* graph_host, int graph_nodes, int accepted, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(f->in_global_loop, "dms_fusion_apply_global_loop_begin has not been called");
  DMS_REQUIRE(graph_nodes == 0 || graph_host, "null graph");
  hipStream_t s = (hipStream_t)st;
  f->in_global_loop = false;
  int rc;
  // predict(context, rf); predictIndices(currPose, tick, id, model, maxDepthProcessed, timeDelta + framesSinceLastFusion);
  // clean(..., rawGraph, timeDelta + framesSinceLastFusion, maxDepthProcessed, orbLoopClosureAccepted) (:1222-1239)
  // (the clean changes the map's version: a projection the previous frame left for the next tracking prediction is stale and
  // that prediction projects afresh, see predict() mode 2)
  // in the reference's order: the prediction images a caller reads afterwards show the map BEFORE this clean
  if ((rc = predict(f, f->p.confidence, s))) return rc;
  const int timeDeltaEff = f->p.timeDelta + f->frames_since_fusion;
  if ((rc = index_map(f->model, &f->state->cur, &f->cam, f->tick, f->p.timeIdx, f->p.maxDepthProcessed, timeDeltaEff, f->zbuf, &f->imap, 1, 1, s)))
    return rc;
  return model_clean(f->model, &f->state->cur, f->tick, f->p.timeIdx, &f->imap, nullptr, &f->cam, f->p.confidence, graph_host, graph_nodes,
                     timeDeltaEff, f->p.maxDepthProcessed, accepted ? 1 : 0, 1, &f->state->surfels, s);
}

int dms_relative_transform(const float* recoveryPose16, const float* currPose16, float* out16) {
  DMS_REQUIRE(recoveryPose16 && currPose16 && out16, "null argument");
  float inv[16];
  sm::inv4t<float>(currPose16, inv);
  sm::mul44_host(recoveryPose16, inv, out16);
  return DMS_OK;
}
int dms_pose_compose(const float* a16, const float* b16, float* out16) {
  DMS_REQUIRE(a16 && b16 && out16, "null argument");
  float tmp[16];
  sm::mul44_host(a16, b16, tmp);
  memcpy(out16, tmp, sizeof(tmp));
  return DMS_OK;
}

// ---- after a map merge: several cameras, one map --------------------------------------------------------------------------------
static int share_model(dms_fusion* f, dms_fusion* owner) {
  DMS_REQUIRE(f && owner && f != owner, "bad argument");
  DMS_REQUIRE(!f->in_frame && !owner->in_frame && !f->in_global_loop && !owner->in_global_loop, "inside a frame");
  DMS_REQUIRE(f->p.width == owner->p.width && f->p.height == owner->p.height, "cameras of different resolution (the reference shares one Resolution singleton)");
  DMS_REQUIRE(f->model != owner->model, "this camera already tracks against the consuming map");
  DMS_REQUIRE(owner->map_initialised, "the consuming map is empty");
  DMS_REQUIRE(f->p.timeIdx != owner->p.timeIdx, "both cameras use the same time slot (timeIdx = Context::id())");
  DMS_REQUIRE(f->p.timeIdx < owner->model->num_sensors && f->p.timeIdx < DMS_MAX_SENSORS, "the consuming map has no time slot for this camera (num_sensors)");
  return DMS_OK;
}

// ReferenceFrame::consumeReferenceFrame moves EVERY camera of the consumed frame (ReferenceFrame.h:127-145).  The camera whose own
// map is the consumed frame's (f->model == f->own_model: the founder) carries the surfels over; a camera that had joined that map
// earlier, or was imported into it (f->model != f->own_model: a chained merge), only moves: pose re-based, sharer counts, map
// pointer.  The order in which a frame's cameras are handed over does not matter.
int dms_fusion_join_map(dms_fusion* f, dms_fusion* owner, const float* relativeTransform16, dms_stream st) {
  int rc = share_model(f, owner);
  if (rc) return rc;
  const bool founder = f->model == f->own_model;
  DMS_REQUIRE(relativeTransform16, "join_map: the transform into the consuming map");
  DMS_REQUIRE(!founder || f->map_initialised, "join_map: a camera with a map of its own");
  DMS_REQUIRE(!f->adopting, "join_map: an imported camera's seeding frame is still pending");
  hipStream_t s = (hipStream_t)st;
  DMS_HIP(hipStreamSynchronize(s));
  if (f->p.pipeline_ingest) DMS_HIP(hipStreamSynchronize(f->s_prep));
  if (founder) {
    // m_localModel.consume(other.globalModel().model(), relativeTransform) (ReferenceFrame.h:124)
    if ((rc = model_flush_pending(f->model, s))) return rc;
    if ((rc = dms_model_consume(owner->model, f->model, relativeTransform16, st))) return rc;
  }
  // kv.second->currPose() = relativeTransform * kv.second->currPose() (:131); the next frame's lastPose is read from it (:158)
  float cur[16];
  Pose16 np;
  DMS_HIP(hipMemcpy(cur, f->state->cur.pose, sizeof(cur), hipMemcpyDeviceToHost));
  sm::mul44_host(relativeTransform16, cur, np.v);
  hipLaunchKernelGGL(k_pose_override, dim3(1), dim3(64), 0, s, f->state, np);
  DMS_CHECK_LAUNCH();
  // a projection rendered ahead for the next frame (share_projection) is of the map this camera leaves: its keys are surfel
  // indices of that map.  predict() only clears zbuf2 while pre_valid is set, so clear it here before dropping the flag.
  if (f->pre_valid && (rc = clear_zbuf(f->zbuf2, f->p.width * f->p.height, s))) return rc;
  f->pre_valid = false;
  if (!founder) f->model->sharers -= 1;  // leaves the map it had joined (that map's founder carries the surfels)
  f->model = owner->model;
  f->model->sharers += 1;
  f->model->count_hold = 3;
  DMS_HIP(hipStreamSynchronize(s));
  return DMS_OK;
}

int dms_fusion_import_camera(dms_fusion* f, dms_fusion* owner, const float* pose16, int tick, const void* last_rgb_dev, int rgb_channels,
                             const unsigned short* last_depth_dev, dms_stream st) {
  int rc = share_model(f, owner);
  if (rc) return rc;
  DMS_REQUIRE(pose16 && last_rgb_dev && last_depth_dev && tick >= 2, "null argument / a camera that has not processed a frame");
  DMS_REQUIRE(!f->map_initialised && f->frames == 0, "import_camera: a context that has not processed a frame");
  f->model = owner->model;
  f->model->sharers += 1;
  f->model->count_hold = 3;
  f->adopting = true;
  f->tick = tick - 1;  // the seeding frame below ends with tick += 1 like any other
  // the camera's last frame again, at its (already transformed) pose: its live pyramids are what the next frame's SO3
  // pre-alignment compares against (lastNextImage), exactly what the camera's own context held after that frame
  rc = dms_fusion_process_frame(f, last_rgb_dev, rgb_channels, last_depth_dev, pose16, 1.f, st);
  if (rc) {  // leave the context as it was
    f->adopting = false;
    f->model->sharers -= 1;
    f->model = f->own_model;
    f->tick = 1;
  }
  return rc;
}

int dms_fusion_get_image(dms_fusion* f, int which, dms_image2d* view) {
  DMS_REQUIRE(f && view, "null argument");
  const dms_image2d* t[] = {&f->rgba,          &f->depth_raw,   &f->depth_filtered, &f->depth_metric, &f->depth_metric_filtered,
                            &f->imap.index,    &f->imap.vertConf, &f->imap.colorTime, &f->imap.normRad, &f->pred.image,
                            &f->pred.vertex,   &f->pred.normal, &f->pred.time,      &f->fill.image,   &f->fill.vertex,
                            &f->fill.normal,   &f->pred_old.image, &f->pred_old.vertex, &f->pred_old.normal, &f->pred_old.time};
  DMS_REQUIRE(which >= 0 && which < 20, "bad image id");
  DMS_REQUIRE(which < 16 || f->p.local_loop_closure || f->p.hybrid_loops, "the INACTIVE view exists only with local_loop_closure or hybrid_loops");
  *view = *t[which];
  if (which >= 5 && which <= 8) {
    // the frame step keeps the index-map images column-major; hand out a row-major copy
    const int elem = which == 5 ? 4 : 16;
    void* dst = f->untr;
    int rc = untranspose(t[which]->data, dst, f->p.width, f->p.height, elem, 0);
    if (rc) return rc;
    DMS_HIP(hipDeviceSynchronize());
    view->data = dst;
  }
  return DMS_OK;
}

int dms_fusion_set_profiling(dms_fusion* f, int enabled) {
  DMS_REQUIRE(f, "null argument");
  f->profiling = enabled != 0;
  if (enabled) f->times.clear();
  return DMS_OK;
}

// ElasticFusion.cpp:1023-1043 (the reference's setters write members that processFrame reads on the next frame)
int dms_fusion_set_tracker_budget(dms_fusion* f, int max_blocks, int unchained) {
  DMS_REQUIRE(f && max_blocks >= 0, "bad argument");
  if (f->in_frame || f->in_global_loop) {
    ::dms::set_error("dms_fusion_set_tracker_budget: inside a frame");
    return DMS_ERR_STATE;
  }
  int rc = dms_odometry_set_resident_budget(f->odom, max_blocks, unchained);
  if (!rc && f->odom_m2m) rc = dms_odometry_set_resident_budget(f->odom_m2m, max_blocks, unchained);
  return rc;
}

int dms_fusion_set_option(dms_fusion* f, int option, double value) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(option >= 0 && option < DMS_OPT_COUNT, "unknown option");
  if (f->in_frame || f->in_global_loop) {
    ::dms::set_error("dms_fusion_set_option: inside a frame");
    return DMS_ERR_STATE;
  }
  const int on = value != 0.0 ? 1 : 0;
  switch (option) {
    case DMS_OPT_RGB_ONLY: f->p.rgbOnly = on; break;
    case DMS_OPT_ICP_WEIGHT: f->p.icpWeight = (float)value; break;
    case DMS_OPT_PYRAMID: f->p.pyramid = on; break;
    case DMS_OPT_FAST_ODOM: f->p.fastOdom = on; break;
    case DMS_OPT_SO3: f->p.so3 = on; break;
    case DMS_OPT_FRAME_TO_FRAME_RGB: f->p.frameToFrameRGB = on; break;
    case DMS_OPT_CONFIDENCE:
      DMS_REQUIRE(value == value, "confidence threshold is NaN");
      f->p.confidence = (float)value;
      break;
    case DMS_OPT_DEPTH_CUTOFF:
      DMS_REQUIRE(value > 0.0, "depth cut-off must be positive");
      f->p.depthCut = (float)value;
      return set_depth_bias(f);
    case DMS_OPT_NID_THRESHOLD: f->p.nid_threshold = (float)value; break;
    case DMS_OPT_NID_DEPTH_LAMBDA: f->p.nid_depth_lambda = (float)value; break;
    case DMS_OPT_NID_BINS_IMG:
    case DMS_OPT_NID_BINS_DEPTH: {
      const int nb = (int)value;
      DMS_REQUIRE(nb >= 1 && nb <= (option == DMS_OPT_NID_BINS_IMG ? 256 : 4096), "bin count out of range");
      DMS_REQUIRE(!f->nid_ws || nid_workspace_bytes(nb) <= f->nid_ws_bytes,
                  "more bins than the NID workspace of this context holds (it is sized from the creation-time bin counts)");
      (option == DMS_OPT_NID_BINS_IMG ? f->p.nid_bins_img : f->p.nid_bins_depth) = nb;
      break;
    }
    case DMS_OPT_NID_PYRAMID_LEVEL:
      DMS_REQUIRE((int)value >= 0 && (int)value < DMS_NUM_PYRS, "pyramid level out of range");
      f->p.nid_pyramid_level = (int)value;
      break;
  }
  return DMS_OK;
}

int dms_fusion_get_option(dms_fusion* f, int option, double* value) {
  DMS_REQUIRE(f && value, "null argument");
  DMS_REQUIRE(option >= 0 && option < DMS_OPT_COUNT, "unknown option");
  switch (option) {
    case DMS_OPT_RGB_ONLY: *value = f->p.rgbOnly; break;
    case DMS_OPT_ICP_WEIGHT: *value = f->p.icpWeight; break;
    case DMS_OPT_PYRAMID: *value = f->p.pyramid; break;
    case DMS_OPT_FAST_ODOM: *value = f->p.fastOdom; break;
    case DMS_OPT_SO3: *value = f->p.so3; break;
    case DMS_OPT_FRAME_TO_FRAME_RGB: *value = f->p.frameToFrameRGB; break;
    case DMS_OPT_CONFIDENCE: *value = f->p.confidence; break;
    case DMS_OPT_DEPTH_CUTOFF: *value = f->p.depthCut; break;
    case DMS_OPT_NID_THRESHOLD: *value = f->p.nid_threshold; break;
    case DMS_OPT_NID_DEPTH_LAMBDA: *value = f->p.nid_depth_lambda; break;
    case DMS_OPT_NID_BINS_IMG: *value = f->p.nid_bins_img; break;
    case DMS_OPT_NID_BINS_DEPTH: *value = f->p.nid_bins_depth; break;
    case DMS_OPT_NID_PYRAMID_LEVEL: *value = f->p.nid_pyramid_level; break;
  }
  return DMS_OK;
}

int dms_fusion_predict(dms_fusion* f, float confidence, dms_stream st) {
  DMS_REQUIRE(f, "null argument");
  DMS_REQUIRE(!f->in_frame && !f->in_global_loop, "inside a frame");
  DMS_REQUIRE(f->map_initialised, "no map yet");
  return predict(f, confidence < 0.f ? f->p.confidence : confidence, (hipStream_t)st);
}

int dms_fusion_get_kernel_time(dms_fusion* f, const char* name, double* total_ms, int* launches) {
  DMS_REQUIRE(f && name && total_ms && launches, "null argument");
  if (strcmp(name, "host_wait") == 0) {  // always on: how long the host has waited for frame t-2 so far (0 = the host is the bottleneck)
    *total_ms = f->host_wait_ms;
    *launches = (int)f->frames;
    return DMS_OK;
  }
  auto it = f->times.find(name);
  *total_ms = it == f->times.end() ? 0.0 : it->second.ms;
  *launches = it == f->times.end() ? 0 : it->second.launches;
  return DMS_OK;
}

}  // extern "C"
