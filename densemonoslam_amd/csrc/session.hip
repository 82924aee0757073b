// The collaborative session behind the C ABI (include/dmslam_session.h): MainController.cpp:262-400's camera loop with the inter-map
// block of ElasticFusion.cpp:595-632 for one rank of a node, composed from the library's own entry points — dms_fusion_* (the cameras),
// dms_ferns_* (the reference frames' key-frame databases), dms_refframe_* (ReferenceFrame's m_index + m_rgbd), dms_model_* / dms_ferns_*
// records (a merge across ranks) — and a dms_transport.  Host code: no kernel of its own.  The protocol is the one of
// densemonoslam_amd/session.py, step for step (same order of queries, same decision rule, same float operations for the re-basing:
// dms_pose_compose / dms_relative_transform), so a session run through this file ends in the same bits as the one-process oracle
// session (tests/test_session_gpu.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../include/dmslam_session.h"
#include "common.hpp"

namespace {

constexpr int kMetaBytes = 80;  // per published block: camera id i32 | tick i32 | 2 x pad | pose 16 x f32
constexpr int kRow = 18;        // table entry: valid, closest, recoveryPose
// The pipelined step's block: thumbnails | tail, the tail = fern codes 512 | good codes i32 | tick i32 | camera id i32 | pad | pose
// 16 x f32 | hit rows 64 x {candidate, dissimilarity bits, codes valid in both, of those equal}.  The hit rows of a rank's i-th slot
// are those of its i-th hosted key-frame database (ascending frame id) against every gathered block, written by the search of the tick
// BEFORE (a rank hosts at most as many databases as cameras, so there is a slot for each).
constexpr int kTailCodes = 0, kTailGood = 512, kTailTick = 516, kTailCam = 520, kTailPose = 528, kTailHits = 592, kMaxQueries = 64;
constexpr int kTailBytes = kTailHits + kMaxQueries * 16;  // 1616
constexpr int kTailHostBytes = kTailBytes - kTailGood;    // what the host mirrors of every gathered block (everything behind the codes)
constexpr int kWakeLatency = 3;  // a hit of the search enqueued at tick j runs the full block at tick j + 3 (dmslam_session.h)

struct Merge {
  int k, fb, fa;
  float T[16];
};
struct Refinement {
  int k, a, fb, accepted;
};
struct Camera {
  dms_fusion* f = nullptr;
  void* last_rgb = nullptr;              // receive buffers of a camera whose frames are forwarded from another rank (RGB8 | depth u16)
  unsigned short* last_depth = nullptr;
  const void* frame_rgb = nullptr;       // where the frame of the current tick lies - the caller's buffer or the receive buffer: what a
  const unsigned short* frame_depth = nullptr;  // migration ships (a merge happens inside the step, while the caller's buffers are valid)
  dms_frame_result last{};
  int tick = 1;
  float pose[16];
  bool has_frame = false;
  std::vector<int> pg_tick;              // Context::poseGraph()
  std::vector<float> pg_pose;            // 16 floats each
  std::vector<float> rel_cons;           // Context::relativeCons(): 6 floats each {src xyz, target xyz}, what the caller's deformation solver produced
};

}  // namespace

struct dms_session {
  dms_session_params p{};
  dms_transport t{};
  bool local_only = true;
  int rank = 0, world = 1, n = 0, W = 0, H = 0;
  std::vector<int> frame_of;             // camera -> reference frame
  std::map<int, int> host_of_frame;      // reference frame -> rank
  std::map<int, Camera> cams;            // hosted here
  std::map<int, dms_ferns*> ferns;       // hosted frames
  std::map<int, dms_refframe*> refiners;
  std::vector<Merge> merges;
  std::vector<Refinement> refinements;
  size_t thumb_bytes = 0, block_bytes = 0;
  // device scratch
  unsigned char* d_local = nullptr;      // n x block_bytes
  unsigned char* d_gathered = nullptr;   // world x n x block_bytes
  float* d_table = nullptr;              // (1 + world) x (n + 1) x n x kRow
  float* d_pose = nullptr;               // 16 floats (a pose for addFrame)
  float* d_small = nullptr;              // (1 + world) x 32 floats: a refinement's status + result of this rank | of every rank
  void* d_frame_rgb = nullptr;           // a forwarded frame
  unsigned short* d_frame_depth = nullptr;
  unsigned char* d_tex = nullptr;        // a remote camera's fill-in textures (36 B per pixel)
  // the pipelined step (dms_session_step_async)
  size_t tail_off = 0, ablock_bytes = 0;
  unsigned char* d_alocal[2] = {nullptr, nullptr};     // n x ablock_bytes each: this rank's blocks of even / odd ticks
  unsigned char* d_agathered[2] = {nullptr, nullptr};  // world x n x ablock_bytes each
  bool can_pipeline = false;
  int ids_tick = -1;                                   // the camera ids in d_alocal's tails are current (placement as of this many merges)
  struct Entry {
    int tick = -1;                 // the tick whose gathered tails this mirror holds (-1: none)
    bool consumed = true;
    int slots = 0;
    unsigned char* host = nullptr; // pinned: world x slots x kTailHostBytes
    hipEvent_t done = nullptr;
    hipEvent_t ag0 = nullptr, ag1 = nullptr;  // params.time_exchange: around the tick's all-gather
    bool timed = false;
  } ring[2];
  double ag_ms = 0.0;
  int ag_n = 0;
  // Pipelined ticks run the cameras of every hosted MAP on a stream of that map's own (cameras that share a map stay serial, as the
  // reference's loop has them; independent maps overlap, and the exchange on the caller's stream runs beside the next frames).
  // DMS_SESSION_MAP_STREAMS=n: a pool of n streams (default 2), 0: everything on the caller's stream.
  struct MapStream {
    hipStream_t s = nullptr;
    hipEvent_t blocks = nullptr;  // this tick's frame blocks of the map's cameras are packed
  };
  std::vector<MapStream> map_streams;  // a small pool: the hosted maps take its streams in turn (more streams than hardware queues cost more than they overlap)
  bool fused_block = true;  // the frame block written by the frame's last kernel (DMS_SESSION_FUSED_BLOCK=0: a launch of its own, the A/B switch)
  bool share_device = true;  // DMS_SESSION_SHARE_DEVICE=0: the cameras' trackers keep their full grids and the chain (round 5)
  int tracker_cap = 0;       // the cap the hosted cameras' trackers carry now (0: none)
  int late_told = -1;        // what the hosted cameras were told about the late frame (dms_fusion_allow_late_frame)
  bool join_by_frame_event = true;  // DMS_SESSION_JOIN_BY_FRAME_EVENT=0: a marker of the session's own behind the frames of every map stream (round 5)
  int n_map_streams = 2;  // (measured on one MI355X, 2 - 8 cameras: two beat one by 5 - 35 %, three and four are no better, four lose with 8 cameras)
  int valid_from = 0;              // searches enqueued before this tick ran on a layout that a merge has changed since
  std::set<int> wake_ticks;        // ticks that run the full inter-map block: a search three ticks earlier hit
  int wakes = 0, async_ticks = 0;
  dms_stream last_stream = nullptr;
};

namespace {

using ::dms::set_error;

int host_of_camera(const dms_session* s, int c) { return s->host_of_frame.at(s->frame_of[c]); }

// DMS_SESSION_TRACE=1: host-side wall-clock marks of the inter-map block (query, refinement, merge) on stderr - where a woken tick's time goes
struct Trace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  Trace() : on(getenv("DMS_SESSION_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what, int a = -1, int b = -1) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[dms_session] %-28s %2d %2d  %8.3f ms\n", what, a, b, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

std::vector<int> hosted(const dms_session* s) {
  std::vector<int> v;
  for (auto& kv : s->cams) v.push_back(kv.first);
  return v;  // (std::map: ascending)
}

int sync(dms_stream st) { return dms_stream_sync(st); }

// ---- transport helpers: the session's own one-rank transport, sized messages ----------------------------------------------------
int t_allgather(dms_session* s, const void* send, void* recv, size_t bytes, dms_stream st) {
  if (s->local_only) return dms_memcpy_d2d_async(recv, send, bytes, st);
  return s->t.allgather(s->t.ctx, send, recv, bytes, st);
}
int t_send(dms_session* s, const void* src, size_t bytes, int peer, dms_stream st) { return bytes ? s->t.send(s->t.ctx, src, bytes, peer, st) : 0; }
int t_recv(dms_session* s, void* dst, size_t bytes, int peer, dms_stream st) { return bytes ? s->t.recv(s->t.ctx, dst, bytes, peer, st) : 0; }
// a host array through the device scratch (headers of a migration: counts, poses)
int send_host(dms_session* s, const void* host, size_t bytes, int peer, dms_stream st) {
  void* d = nullptr;
  int rc = dms_device_alloc(&d, bytes ? bytes : 4);
  if (rc) return rc;
  if (bytes) rc = dms_memcpy_h2d(d, host, bytes, st);
  if (!rc) rc = t_send(s, d, bytes, peer, st);
  if (!rc) rc = sync(st);
  dms_device_free(d);
  return rc;
}
int recv_host(dms_session* s, void* host, size_t bytes, int peer, dms_stream st) {
  void* d = nullptr;
  int rc = dms_device_alloc(&d, bytes ? bytes : 4);
  if (rc) return rc;
  rc = t_recv(s, d, bytes, peer, st);
  if (!rc) rc = sync(st);
  if (!rc && bytes) rc = dms_memcpy_d2h(host, d, bytes, st);
  dms_device_free(d);
  return rc;
}

int make_camera(dms_session* s, int c, Camera& cam) {
  dms_fusion_params fp = s->p.camera;
  fp.timeIdx = c;
  int rc = dms_fusion_create(&cam.f, &fp);
  if (rc) return rc;
  const size_t N = (size_t)s->W * s->H;
  if ((rc = dms_device_alloc(&cam.last_rgb, N * 3))) return rc;
  void* d = nullptr;
  if ((rc = dms_device_alloc(&d, N * 2))) return rc;
  cam.last_depth = (unsigned short*)d;
  for (int i = 0; i < 16; ++i) cam.pose[i] = (i % 5 == 0) ? 1.f : 0.f;
  return DMS_OK;
}
void free_camera(Camera& cam) {
  if (cam.f) dms_fusion_destroy(cam.f);
  if (cam.last_rgb) dms_device_free(cam.last_rgb);
  if (cam.last_depth) dms_device_free(cam.last_depth);
  cam = Camera();
}
int make_ferns(dms_session* s, dms_ferns** out) {
  const dms_fusion_params& c = s->p.camera;
  return dms_ferns_create(out, s->p.fern_num, s->p.fern_max_depth_mm, s->p.fern_photo_thresh, s->W, s->H, c.cx, c.cy, c.fx, c.fy, s->p.fern_seed,
                          s->p.fern_capacity);
}

void rebase(Camera& cam, const float* T) {  // kv.second->poseGraph()[i].second = relativeTransform * ... (ReferenceFrame.h:138-141)
  for (size_t i = 0; i < cam.pg_tick.size(); ++i) dms_pose_compose(T, &cam.pg_pose[i * 16], &cam.pg_pose[i * 16]);
  // relativeCons()[i].src = r * src + t, .target likewise (:133-136): ((T0 x + T1 y) + T2 z) + T3, every operation rounded
  for (size_t i = 0; i + 2 < cam.rel_cons.size(); i += 3) {
    float* p = &cam.rel_cons[i];
    const float x = p[0], y = p[1], z = p[2];
    for (int r = 0; r < 3; ++r) {
      float v = T[r * 4 + 0] * x;
      v = v + T[r * 4 + 1] * y;
      v = v + T[r * 4 + 2] * z;
      p[r] = v + T[r * 4 + 3];
    }
  }
}

Camera* owner_of(dms_session* s, int fb) {
  for (auto& kv : s->cams)
    if (s->frame_of[kv.first] == fb) return &kv.second;
  return nullptr;
}

// ReferenceFrame::resolveRelativeTransformationFern's second half for camera a against frame fb, on fb's rank; every rank ends with the
// same {accepted, relativeTransform} - or, when ANY rank failed on the way, with an error on every rank: each rank's status rides the
// all-gather that carries the owner's result, the owner posts its receives before anything that can fail, and the sender sends its three
// messages whatever happened before them (from the scratch textures if the camera's cannot be read), so nobody is left waiting in a
// collective or a point-to-point call.
int refine(dms_session* s, int a, int fb, const float* rec16, const float* curr16, int tick, int* accepted, float* T16, dms_stream st) {
  const int ha = host_of_camera(s, a), hb = s->host_of_frame.at(fb);
  const size_t N = (size_t)s->W * s->H;
  constexpr int kRes = 18;  // status (0 / -1) | accepted | relativeTransform
  float res[kRes];
  memset(res, 0, sizeof(res));
  int my_rc = DMS_OK, rc = DMS_OK;
  if (hb == s->rank) {
    const float *vtx = nullptr, *nrm = nullptr;
    const void* img = nullptr;
    if (ha != s->rank) {  // the receives first
      if ((rc = t_recv(s, s->d_tex, N * 4, ha, st)) || (rc = t_recv(s, s->d_tex + N * 4, N * 16, ha, st)) ||
          (rc = t_recv(s, s->d_tex + N * 20, N * 16, ha, st)) || (rc = sync(st)))
        my_rc = rc;
      img = s->d_tex;
      vtx = (const float*)(s->d_tex + N * 4);
      nrm = (const float*)(s->d_tex + N * 20);
    }
    auto owner_side = [&]() -> int {
      Camera* owner = owner_of(s, fb);
      DMS_REQUIRE(owner, "the matched frame has no camera here");
      int r0;
      if (!s->refiners.count(fb)) {
        const dms_fusion_params& c = s->p.camera;
        dms_refframe* r = nullptr;
        if ((r0 = dms_refframe_create(&r, s->W, s->H, c.cx, c.cy, c.fx, c.fy))) return r0;
        s->refiners[fb] = r;
      }
      if (ha == s->rank) {
        dms_image2d vi, vv, vn;
        dms_fusion* q = s->cams.at(a).f;
        if ((r0 = dms_fusion_get_image(q, 13, &vi)) || (r0 = dms_fusion_get_image(q, 14, &vv)) || (r0 = dms_fusion_get_image(q, 15, &vn))) return r0;
        img = vi.data;
        vtx = (const float*)vv.data;
        nrm = (const float*)vn.data;
      }
      double conf = 0.0;
      if ((r0 = dms_fusion_get_option(owner->f, DMS_OPT_CONFIDENCE, &conf))) return r0;
      dms_intermap_result r;
      if ((r0 = dms_refframe_refine(s->refiners[fb], dms_fusion_model(owner->f), rec16, curr16, vtx, nrm, img, (int)s->p.camera.maxDepthProcessed,
                                    (float)conf, a, s->p.camera.timeDelta, tick, s->p.cov_thresh, s->p.icp_err_thresh, s->p.icp_count_thresh, &r, st)))
        return r0;
      res[1] = r.accepted ? 1.f : 0.f;
      memcpy(res + 2, r.relativeTransform, 64);
      return DMS_OK;
    };
    if (!my_rc) my_rc = owner_side();
  } else if (ha == s->rank) {
    dms_image2d vi, vv, vn;
    dms_fusion* q = s->cams.at(a).f;
    const void *pi = s->d_tex, *pv = s->d_tex + N * 4, *pn = s->d_tex + N * 20;
    if ((rc = dms_fusion_get_image(q, 13, &vi)) || (rc = dms_fusion_get_image(q, 14, &vv)) || (rc = dms_fusion_get_image(q, 15, &vn))) {
      my_rc = rc;  // (the owner has posted three receives: they are matched, with the scratch textures)
    } else {
      pi = vi.data;
      pv = vv.data;
      pn = vn.data;
    }
    if ((rc = t_send(s, pi, N * 4, hb, st)) || (rc = t_send(s, pv, N * 16, hb, st)) || (rc = t_send(s, pn, N * 16, hb, st)) || (rc = sync(st)))
      if (!my_rc) my_rc = rc;
  }
  if (s->local_only) {
    if (my_rc) return my_rc;
  } else {
    // (the message of a failure here belongs to this rank's own error; keep it across the calls below)
    res[0] = my_rc ? -1.f : 0.f;
    std::vector<float> all((size_t)kRes * s->world, 0.f);
    if ((rc = dms_memcpy_h2d(s->d_small, res, sizeof(res), st)) || (rc = t_allgather(s, s->d_small, s->d_small + kRes, sizeof(res), st)) || (rc = sync(st)) ||
        (rc = dms_memcpy_d2h(all.data(), s->d_small + kRes, sizeof(res) * s->world, st)))
      return my_rc ? my_rc : rc;
    for (int r = 0; r < s->world; ++r)
      if (all[(size_t)r * kRes] < 0.f) {
        if (!my_rc) set_error("dms_session_step: the inter-map refinement of camera %d against frame %d failed on rank %d", a, fb, r);
        return my_rc ? my_rc : DMS_ERR_STATE;
      }
    memcpy(res, &all[(size_t)hb * kRes], sizeof(res));
  }
  *accepted = res[1] == 1.f ? 1 : 0;
  memcpy(T16, res + 2, 64);
  return DMS_OK;
}

struct MigrationHeader {  // per moving camera, consumed side -> consuming side
  int tick, n_pg, has_frame, n_rc;
  float pose[16];
};

// reference frame fb consumes fa (ReferenceFrame::consumeReferenceFrame)
int merge(dms_session* s, int k, int fb, int fa, const float* T, dms_stream st) {
  const int hb = s->host_of_frame.at(fb), ha = s->host_of_frame.at(fa);
  std::vector<int> moving;
  for (int c = 0; c < s->n; ++c)
    if (s->frame_of[c] == fa) moving.push_back(c);
  const size_t N = (size_t)s->W * s->H;
  int rc = DMS_OK;
  if (hb == ha) {
    if (hb == s->rank) {
      Camera* owner = owner_of(s, fb);
      DMS_REQUIRE(owner, "the consuming frame has no camera here");
      // the camera that owns fa's map carries it over, the rest only move (dms_fusion_join_map)
      std::vector<int> order = moving;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (x != fa) < (y != fa) || ((x != fa) == (y != fa) && x < y); });
      for (int c : order) {
        Camera& cam = s->cams.at(c);
        if ((rc = dms_fusion_join_map(cam.f, owner->f, T, st))) return rc;
        dms_pose_compose(T, cam.pose, cam.pose);
        rebase(cam, T);
      }
      int added = 0;
      if ((rc = dms_ferns_consume(s->ferns.at(fb), s->ferns.at(fa), T, s->p.fern_threshold, &added, st))) return rc;
      dms_ferns_destroy(s->ferns.at(fa));
      s->ferns.erase(fa);
    }
  } else if (ha == s->rank) {  // the consumed side: ship everything, free the local copies
    Camera& founder = s->cams.at(fa);
    dms_model* m = dms_fusion_model(founder.f);
    unsigned cnt = 0;
    if ((rc = dms_model_count(m, &cnt, st))) return rc;
    long long hdr[2] = {(long long)cnt, (long long)dms_ferns_num_frames(s->ferns.at(fa))};
    if ((rc = send_host(s, hdr, sizeof(hdr), hb, st))) return rc;
    if (cnt) {
      void* rec = nullptr;
      if ((rc = dms_device_alloc(&rec, (size_t)cnt * 80))) return rc;
      unsigned got = 0;
      rc = dms_model_export_records(m, (float*)rec, cnt, &got, st);
      if (!rc) rc = sync(st);
      if (!rc && got != cnt) {
        set_error("dms_session: the map changed while it was exported");
        rc = DMS_ERR_STATE;
      }
      if (!rc) rc = t_send(s, rec, (size_t)cnt * 80, hb, st);
      if (!rc) rc = sync(st);
      dms_device_free(rec);
      if (rc) return rc;
    }
    if (hdr[1]) {
      const size_t rb = dms_ferns_record_bytes(s->ferns.at(fa));
      void* rec = nullptr;
      if ((rc = dms_device_alloc(&rec, (size_t)hdr[1] * rb))) return rc;
      int got = 0;
      rc = dms_ferns_export_records(s->ferns.at(fa), rec, (int)hdr[1], &got, st);
      if (!rc) rc = sync(st);
      if (!rc && got != (int)hdr[1]) {
        set_error("dms_session: the key-frame database changed while it was exported");
        rc = DMS_ERR_STATE;
      }
      if (!rc) rc = t_send(s, rec, (size_t)hdr[1] * rb, hb, st);
      if (!rc) rc = sync(st);
      dms_device_free(rec);
      if (rc) return rc;
    }
    for (int c : moving) {
      Camera& cam = s->cams.at(c);
      MigrationHeader h;
      memset(&h, 0, sizeof(h));
      h.tick = cam.tick;
      h.n_pg = (int)cam.pg_tick.size();
      h.has_frame = cam.has_frame ? 1 : 0;
      h.n_rc = (int)(cam.rel_cons.size() / 6);
      memcpy(h.pose, cam.pose, 64);
      if ((rc = send_host(s, &h, sizeof(h), hb, st))) return rc;
      DMS_REQUIRE(cam.frame_rgb && cam.frame_depth, "a camera migrates before its first frame");
      if ((rc = t_send(s, cam.frame_rgb, N * 3, hb, st)) || (rc = t_send(s, cam.frame_depth, N * 2, hb, st)) || (rc = sync(st))) return rc;
      if ((rc = send_host(s, cam.pg_tick.data(), cam.pg_tick.size() * 4, hb, st)) || (rc = send_host(s, cam.pg_pose.data(), cam.pg_pose.size() * 4, hb, st)) ||
          (rc = send_host(s, cam.rel_cons.data(), cam.rel_cons.size() * 4, hb, st)))
        return rc;
    }
    // joined / imported cameras first: the map's owner outlives them
    for (int c : moving)
      if (c != fa) {
        free_camera(s->cams.at(c));
        s->cams.erase(c);
      }
    free_camera(s->cams.at(fa));
    s->cams.erase(fa);
    dms_ferns_destroy(s->ferns.at(fa));
    s->ferns.erase(fa);
  } else if (hb == s->rank) {  // the consuming side
    Camera* owner = owner_of(s, fb);
    DMS_REQUIRE(owner, "the consuming frame has no camera here");
    long long hdr[2] = {0, 0};
    if ((rc = recv_host(s, hdr, sizeof(hdr), ha, st))) return rc;
    if (hdr[0]) {
      void* rec = nullptr;
      if ((rc = dms_device_alloc(&rec, (size_t)hdr[0] * 80))) return rc;
      rc = t_recv(s, rec, (size_t)hdr[0] * 80, ha, st);
      if (!rc) rc = sync(st);
      if (!rc) rc = dms_model_consume_records(dms_fusion_model(owner->f), (const float*)rec, (unsigned)hdr[0], T, st);
      if (!rc) rc = sync(st);
      dms_device_free(rec);
      if (rc) return rc;
    }
    if (hdr[1]) {
      const size_t rb = dms_ferns_record_bytes(s->ferns.at(fb));
      void* rec = nullptr;
      if ((rc = dms_device_alloc(&rec, (size_t)hdr[1] * rb))) return rc;
      int added = 0;
      rc = t_recv(s, rec, (size_t)hdr[1] * rb, ha, st);
      if (!rc) rc = sync(st);
      if (!rc) rc = dms_ferns_consume_records(s->ferns.at(fb), rec, (int)hdr[1], T, s->p.fern_threshold, &added, st);
      dms_device_free(rec);
      if (rc) return rc;
    }
    for (int c : moving) {
      MigrationHeader h;
      if ((rc = recv_host(s, &h, sizeof(h), ha, st))) return rc;
      Camera& cam = s->cams[c];
      if ((rc = make_camera(s, c, cam))) return rc;
      if ((rc = t_recv(s, cam.last_rgb, N * 3, ha, st)) || (rc = t_recv(s, cam.last_depth, N * 2, ha, st)) || (rc = sync(st))) return rc;
      cam.pg_tick.resize(h.n_pg);
      cam.pg_pose.resize((size_t)h.n_pg * 16);
      cam.rel_cons.resize((size_t)h.n_rc * 6);
      if ((rc = recv_host(s, cam.pg_tick.data(), cam.pg_tick.size() * 4, ha, st)) || (rc = recv_host(s, cam.pg_pose.data(), cam.pg_pose.size() * 4, ha, st)) ||
          (rc = recv_host(s, cam.rel_cons.data(), cam.rel_cons.size() * 4, ha, st)))
        return rc;
      float moved[16];
      dms_pose_compose(T, h.pose, moved);
      if ((rc = dms_fusion_import_camera(cam.f, owner->f, moved, h.tick, cam.last_rgb, 3, cam.last_depth, st))) return rc;
      cam.tick = h.tick;
      cam.frame_rgb = cam.last_rgb;
      cam.frame_depth = cam.last_depth;
      memcpy(cam.pose, moved, 64);
      cam.has_frame = h.has_frame != 0;
      rebase(cam, T);
    }
  }
  for (int c : moving) s->frame_of[c] = fb;
  if (s->refiners.count(fa)) {  // (the consumed reference frame is erased, ElasticFusion.cpp:610-616)
    dms_refframe_destroy(s->refiners.at(fa));
    s->refiners.erase(fa);
  }
  s->host_of_frame.erase(fa);
  Merge mg;
  mg.k = k;
  mg.fb = fb;
  mg.fa = fa;
  memcpy(mg.T, T, 64);
  s->merges.push_back(mg);
  return DMS_OK;
}

// `only`: bit a * 8 + fb set = the pair (camera a, frame fb) is due (the pipelined mode's candidates); ~0 = every pair
int query(dms_session* s, int k, unsigned long long only, std::map<int, std::vector<float>>& poses, std::map<int, int>& ticks,
          std::map<int, const unsigned char*>& blocks, float* table, dms_stream st) {
  // owner computes: every hosted reference frame against every camera of another frame
  for (auto& kv : s->ferns) {
    const int fb = kv.first;
    for (int a = 0; a < s->n; ++a) {
      if (s->frame_of[a] == fb || k < s->p.query_from || !((only >> (a * 8 + fb)) & 1ull)) continue;
      dms_fern_match m;
      int rc = dms_ferns_find_frame_thumbs(kv.second, blocks.at(a), poses.at(a).data(), ticks.at(a), 0, s->p.inter_map, &m, nullptr, st);
      if (rc) return rc;
      float* e = table + ((size_t)a * s->n + fb) * kRow;
      e[0] = 1.f;
      e[1] = (float)m.closest;
      memcpy(e + 2, m.estPose, 64);
    }
  }
  return DMS_OK;
}

// ---- the pipelined step's host side ------------------------------------------------------------------------------------------
std::vector<int> cams_of_rank(const dms_session* s, int r) {
  std::vector<int> v;
  for (int c = 0; c < s->n; ++c)
    if (host_of_camera(s, c) == r) v.push_back(c);
  return v;
}
std::vector<int> frames_of_rank(const dms_session* s, int r) {
  std::vector<int> v;
  for (auto& kv : s->host_of_frame)
    if (kv.second == r) v.push_back(kv.first);
  return v;
}
int slots_now(const dms_session* s) {
  std::vector<int> host_counts(s->world, 0);
  for (int c = 0; c < s->n; ++c) host_counts[host_of_camera(s, c)] += 1;
  return *std::max_element(host_counts.begin(), host_counts.end());
}
const unsigned char* entry_tail(const dms_session::Entry& e, int r, int i) { return e.host + ((size_t)r * e.slots + i) * kTailHostBytes; }

// Takes a mirror of gathered tails that the device has finished: the pose graph / pose of the cameras hosted here (unless the tick
// was fetched), and the hit rows - an eligible hit of the search enqueued at tick j schedules the full block for tick j + 3.  The
// placement has not changed since the entry was enqueued: a merge consumes every outstanding entry first.
int consume_entry(dms_session* s, dms_session::Entry& e, bool bookkeep) {
  if (e.tick < 0 || e.consumed) return DMS_OK;
  if (hipEventSynchronize(e.done) != hipSuccess) {
    set_error("dms_session: waiting for the gathered blocks of tick %d failed", e.tick);
    return DMS_ERR_HIP;
  }
  e.consumed = true;
  if (e.timed) {  // (the tick is complete: so is its all-gather)
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.ag0, e.ag1) == hipSuccess) {
      s->ag_ms += ms;
      s->ag_n += 1;
    }
    e.timed = false;
  }
  if (bookkeep) {
    const std::vector<int> mine = cams_of_rank(s, s->rank);
    for (size_t i = 0; i < mine.size(); ++i) {
      const unsigned char* t = entry_tail(e, s->rank, (int)i);
      int tick, cam_id;
      memcpy(&tick, t + (kTailTick - kTailGood), 4);
      memcpy(&cam_id, t + (kTailCam - kTailGood), 4);
      DMS_REQUIRE(cam_id == mine[i], "a gathered block is not the camera the placement names");
      Camera& cam = s->cams.at(mine[i]);
      memcpy(cam.pose, t + (kTailPose - kTailGood), 64);
      cam.pg_tick.push_back(tick - 1);  // (never lost in this mode: the tick before the frame)
      cam.pg_pose.insert(cam.pg_pose.end(), cam.pose, cam.pose + 16);
      cam.has_frame = true;
    }
  }
  // the hit rows were written by the search of tick e.tick - 1
  const int searched = e.tick - 1;
  if (searched < s->valid_from || searched < s->p.query_from) return DMS_OK;
  for (int r = 0; r < s->world; ++r) {
    const std::vector<int> dbs = frames_of_rank(s, r);
    for (size_t d = 0; d < dbs.size(); ++d) {
      const unsigned char* rows = entry_tail(e, r, (int)d) + (kTailHits - kTailGood);
      for (int r2 = 0; r2 < s->world; ++r2) {
        const std::vector<int> qs = cams_of_rank(s, r2);
        for (size_t i2 = 0; i2 < qs.size(); ++i2) {
          if (s->frame_of[qs[i2]] == dbs[d]) continue;
          int row[4];
          memcpy(row, rows + ((size_t)r2 * e.slots + i2) * 16, 16);
          // Ferns.cpp:346: the FLOAT ratio against the DOUBLE literal 0.3 (150 / 500 rounds to 0.3f, which is above 0.3)
          if (row[0] >= 0 && (double)((float)row[3] / (float)row[2]) > 0.3) s->wake_ticks.insert(searched + kWakeLatency);
        }
      }
    }
  }
  return DMS_OK;
}

// a merge (or a synchronous step) at tick k: the searches enqueued up to it ran on the placement before it
void invalidate_searches(dms_session* s, int k) {
  s->valid_from = k + 1;
  s->wake_ticks.erase(s->wake_ticks.begin(), s->wake_ticks.lower_bound(s->valid_from + kWakeLatency));
}

// everything the device still owes the host of earlier pipelined ticks (pose graphs), in tick order
int drain_entries(dms_session* s) {
  int order[2] = {0, 1};
  if (s->ring[0].tick > s->ring[1].tick) std::swap(order[0], order[1]);
  for (int b : order) {
    int rc = consume_entry(s, s->ring[b], true);
    if (rc) return rc;
  }
  return DMS_OK;
}

// Trackers that run at the same time share the device.  In a pipelined tick the cameras of different hosted maps run on different streams;
// by default resident tracker launches of different streams wait for each other (each needs all its blocks on the device at once:
// level 0 takes 200 of the 256 units).  With `concurrent` streams the session caps every hosted camera's grids at (units - 16) /
// concurrent blocks and lifts the chain for them (dms_odometry_set_resident_budget: 120 blocks each for two streams - level 0 then runs
// five pixels per thread); one stream, a synchronous tick, or a frame too large for the capped grid: full grids, chained.  Same bits
// either way.  Measured, two cameras before their maps merge: 3 155 against 2 566 frames/s; one stream with the cap: 2 258 against 2 437 -
// hence per tick, by what the tick will run.  The inter-map trackers (queries, refinement) run in woken ticks, after every frame of
// the tick has been fetched: alone.
int apply_tracker_policy(dms_session* s, int concurrent) {
  int cap = 0;
  if (s->share_device && concurrent >= 2) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 64) {
      cap = (cus - 16) / concurrent;
      const long long px = (long long)s->W * s->H;
      if (px > (long long)cap * 512 * 5) cap = 0;  // (level 0 at five pixels per thread of 512-thread blocks would not fit: a capped handle falls back to a launch per phase)
    }
  }
  // (the "late frame" of dmslam_fusion.h is not for a session's cameras: the host has the exchange to enqueue and the earlier ticks'
  // tails to wait for between two frames, and its slack does not show where the context looks for it - measured with one hosted camera:
  // 0.945 and 0.878 of the bare frame rate against 0.955 - 0.957)
  const int late = 0;
  if (late != s->late_told) {
    for (auto& kv : s->cams)
      if (int rc = dms_fusion_allow_late_frame(kv.second.f, late)) return rc;
    s->late_told = late;
  }
  if (cap == s->tracker_cap) return DMS_OK;
  for (auto& kv : s->cams)
    if (int rc = dms_fusion_set_tracker_budget(kv.second.f, cap, cap > 0 ? 1 : 0)) return rc;
  s->tracker_cap = cap;
  return DMS_OK;
}

// the stream a hosted map's cameras run on in a pipelined tick: its own, unless one of its cameras is read on another rank (the
// frame arrives through the transport, whose calls stay on the caller's stream)
int stream_of_map(dms_session* s, int frame, hipStream_t caller, hipStream_t* out, dms_session::MapStream** ms_out) {
  *out = caller;
  *ms_out = nullptr;
  // One camera in a one-rank session: nothing to overlap (measured 2 % slower across two streams).  One camera per rank of several:
  // the all-gather is a real collective whose latency and whose wait for the slowest rank belong beside the next frame, not in front of it.
  // (a one-rank session WITH a transport - the one-GPU rehearsal of the multi-rank loop - takes the multi-rank arrangement)
  if (s->n_map_streams <= 0 || (s->cams.size() < 2 && s->world == 1 && s->local_only)) return DMS_OK;
  for (int c = 0; c < s->n; ++c)
    if (s->frame_of[c] == frame && c % s->world != s->rank) return DMS_OK;
  if (s->map_streams.empty()) s->map_streams.resize(s->n_map_streams);
  int order = 0;  // this map's place among the maps hosted here (ascending frame id)
  for (auto& kv : s->host_of_frame)
    if (kv.second == s->rank && kv.first < frame) ++order;
  dms_session::MapStream& ms = s->map_streams[order % s->n_map_streams];
  if (!ms.s) {
    if (hipStreamCreateWithFlags(&ms.s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ms.blocks, hipEventDisableTiming) != hipSuccess) {
      set_error("dms_session: could not create a map's stream");
      return DMS_ERR_HIP;
    }
  }
  *out = ms.s;
  *ms_out = &ms;
  return DMS_OK;
}

// Phases 4 - 6 of a tick on the gathered blocks (camera -> thumbnails in HBM, pose, tick): queries, the common decision walk with the
// full-resolution refinement, merges.  `only` as in query().
int decide_and_merge(dms_session* s, int k, unsigned long long only, std::map<int, std::vector<float>>& poses, std::map<int, int>& ticks,
                     std::map<int, const unsigned char*>& blocks, dms_stream st) {
  int rc = DMS_OK;
  // 4. queries.  A failure on one rank must not leave the others waiting in the next collective: it travels in the table's last
  // row and every rank returns together.
  const size_t tab_floats = (size_t)(s->n + 1) * s->n * kRow;
  std::vector<float> table(tab_floats, 0.f), tables(tab_floats * s->world, 0.f);
  Trace tr;
  int qrc = query(s, k, only, poses, ticks, blocks, table.data(), st);
  tr.mark("queries (findFrame)", k);
  if (qrc) table[(size_t)s->n * s->n * kRow] = 1.f;
  if ((rc = dms_memcpy_h2d(s->d_table, table.data(), tab_floats * 4, st))) return rc;
  if ((rc = t_allgather(s, s->d_table, s->d_table + tab_floats, tab_floats * 4, st)) || (rc = sync(st))) return rc;
  if ((rc = dms_memcpy_d2h(tables.data(), s->d_table + tab_floats, tab_floats * 4 * s->world, st))) return rc;
  for (int r = 0; r < s->world; ++r)
    if (tables[(size_t)r * tab_floats + (size_t)s->n * s->n * kRow] != 0.f) {
      if (!qrc) set_error("dms_session_step: the inter-map query failed on rank %d at tick %d", r, k);
      return qrc ? qrc : DMS_ERR_STATE;
    }
  // 5. the same walk on every rank (cameras in id order, each over the other cameras' frames in camera-id order; first accepted candidate
  // wins, one merge per frame and tick)
  struct Decided {
    int fb, fa;
    float T[16];
  };
  std::vector<Decided> decided;
  std::set<int> busy;
  for (int a = 0; a < s->n; ++a) {
    const int fa = s->frame_of[a];
    if (busy.count(fa) || k < s->p.query_from) continue;
    // ElasticFusion.cpp:598-599 walks m_contextToReferenceFrameMap - CONTEXT ids in ascending order, each mapped to its frame - so a frame
    // that holds several cameras is visited once per camera (a revisit repeats the query - same database, same block: the table's row -
    // and the refinement, whose m_rgbd has moved on by one call)
    for (int c = 0; c < s->n; ++c) {
      const int fb = s->frame_of[c];
      if (fb == fa || busy.count(fb) || !((only >> (a * 8 + fb)) & 1ull)) continue;
      const float* e = &tables[(size_t)s->host_of_frame.at(fb) * tab_floats + ((size_t)a * s->n + fb) * kRow];
      DMS_REQUIRE(e[0] == 1.f, "no verification result for a camera / frame pair");
      if (e[1] < 0.f) continue;
      Decided d;
      d.fb = fb;
      d.fa = fa;
      if (s->p.full_refine) {
        int accepted = 0;
        tr.mark("table + walk", a, fb);
        if ((rc = refine(s, a, fb, e + 2, poses.at(a).data(), ticks.at(a), &accepted, d.T, st))) return rc;
        tr.mark("refine", a, fb);
        s->refinements.push_back(Refinement{k, a, fb, accepted});
        if (!accepted) continue;
      } else {
        dms_relative_transform(e + 2, poses.at(a).data(), d.T);
      }
      decided.push_back(d);
      busy.insert(fa);
      busy.insert(fb);
      break;
    }
  }
  // 6. merges
  for (auto& d : decided) {
    if ((rc = merge(s, k, d.fb, d.fa, d.T, st))) return rc;
    if ((rc = sync(st))) return rc;
    tr.mark("merge", d.fb, d.fa);
  }
  return DMS_OK;
}

// dms_session_step with params.query_inside_frame: the reference's order.  ElasticFusion::processFrame of camera c ends with the inter-map
// block (ElasticFusion.cpp:595-632): camera c has been offered to its own map's database (processFerns, :588-591), queries the other
// cameras' frames in camera-id order, and a successful query merges at once - all before camera c + 1's processFrame of the same tick
// (MainController.cpp:262-400 serves the cameras in turn).  Per camera: forward / frame / fetch on its host, ONE block published and
// gathered (the other ranks contribute an empty one), owner-computes table, walk of that one camera, merge.
int step_inside(dms_session* s, int k, const void* const* rgb_dev, const unsigned short* const* depth_dev, dms_stream st) {
  const size_t N = (size_t)s->W * s->H;
  int rc = DMS_OK;
  std::map<int, int> read_index;
  {
    int i = 0;
    for (int c = 0; c < s->n; ++c)
      if (c % s->world == s->rank) read_index[c] = i++;
  }
  for (int c = 0; c < s->n; ++c) {
    const int src = c % s->world, host = host_of_camera(s, c);
    if (src != host && src == s->rank) {
      const int i = read_index.at(c);
      if ((rc = t_send(s, rgb_dev[i], N * 3, host, st)) || (rc = t_send(s, depth_dev[i], N * 2, host, st)) || (rc = sync(st))) return rc;
    }
    if ((rc = dms_memset(s->d_local, 0, s->block_bytes, st))) return rc;
    if (host == s->rank) {
      Camera& cam = s->cams.at(c);
      if (src == s->rank) {
        const int i = read_index.at(c);
        if ((rc = dms_fusion_process_frame(cam.f, rgb_dev[i], 3, depth_dev[i], nullptr, 1.f, st))) return rc;
        cam.frame_rgb = rgb_dev[i];
        cam.frame_depth = depth_dev[i];
      } else {
        cam.frame_rgb = cam.last_rgb;
        cam.frame_depth = cam.last_depth;
        if ((rc = t_recv(s, cam.last_rgb, N * 3, src, st)) || (rc = t_recv(s, cam.last_depth, N * 2, src, st))) return rc;
        if ((rc = dms_fusion_inputs_ready(cam.f, st))) return rc;
        if ((rc = dms_fusion_process_frame(cam.f, cam.last_rgb, 3, cam.last_depth, nullptr, 1.f, st))) return rc;
      }
      const int tick_before = cam.tick;
      rc = dms_fusion_fetch(cam.f, &cam.last, st);
      if (rc && rc != DMS_ERR_CAPACITY) return rc;
      cam.tick = cam.last.tick;
      memcpy(cam.pose, cam.last.pose, 64);
      cam.pg_tick.push_back(tick_before);
      cam.pg_pose.insert(cam.pg_pose.end(), cam.pose, cam.pose + 16);
      cam.has_frame = true;
      unsigned char* blk = s->d_local;
      if ((rc = dms_fusion_thumbnails(cam.f, blk, st))) return rc;
      float meta[kMetaBytes / 4];
      memset(meta, 0, sizeof(meta));
      const int ids[2] = {c, cam.tick};
      memcpy(meta, ids, 8);
      memcpy(meta + 4, cam.pose, 64);
      if ((rc = dms_memcpy_h2d(blk + s->thumb_bytes, meta, kMetaBytes, st))) return rc;
      if (!cam.last.lost) {  // processFerns (ElasticFusion.cpp:588-591)
        if ((rc = dms_ferns_add_frame_async(s->ferns.at(s->frame_of[c]), nullptr, nullptr, nullptr, blk, nullptr, (const float*)(blk + s->thumb_bytes + 16),
                                            cam.tick, s->p.fern_threshold, st)))
          return rc;
      }
    }
    if (k < s->p.query_from) continue;  // (uniform: nobody enters the collectives below)
    bool others = false;
    for (int c2 = 0; c2 < s->n; ++c2) others = others || s->frame_of[c2] != s->frame_of[c];
    if (!others) continue;
    if ((rc = t_allgather(s, s->d_local, s->d_gathered, s->block_bytes, st)) || (rc = sync(st))) return rc;
    const unsigned char* raw = s->d_gathered + (size_t)host * s->block_bytes;
    float meta[kMetaBytes / 4];
    if ((rc = dms_memcpy_d2h(meta, raw + s->thumb_bytes, kMetaBytes, st))) return rc;
    int ids[2];
    memcpy(ids, meta, 8);
    DMS_REQUIRE(ids[0] == c, "the gathered block is not the camera whose turn it is");
    std::map<int, std::vector<float>> poses;
    std::map<int, int> ticks;
    std::map<int, const unsigned char*> blocks;
    blocks[c] = raw;
    ticks[c] = ids[1];
    poses[c] = std::vector<float>(meta + 4, meta + 20);
    unsigned long long only = 0;
    for (int fb = 0; fb < s->n; ++fb) only |= 1ull << (c * 8 + fb);
    if ((rc = decide_and_merge(s, k, only, poses, ticks, blocks, st))) return rc;
  }
  return DMS_OK;
}

}  // namespace

extern "C" {

void dms_session_default_params(dms_session_params* p, int n_cameras, int width, int height, float fx, float fy, float cx, float cy) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->n_cameras = n_cameras;
  dms_fusion_default_params(&p->camera, width, height, fx, fy, cx, cy);
  p->camera.num_sensors = n_cameras > 3 ? n_cameras : 3;
  p->fern_num = 500;             // ReferenceFrame.h:17
  p->fern_max_depth_mm = 3000;   // Options::depth (3 m) * 1000
  p->fern_photo_thresh = 115.f;  // Options::interMapPhotoThresh
  p->fern_seed = 20260929u;
  p->fern_capacity = 1024;
  p->fern_threshold = 0.3095f;   // Options::fernThresh
  p->inter_map = 1;
  p->query_from = 0;
  p->full_refine = 1;
  p->cov_thresh = 1e-05f;        // Options.h:91-94
  p->icp_err_thresh = 2e-05f;
  p->icp_count_thresh = 35000.f;
}

static int rccl_allgather(void* c, const void* s, void* r, size_t b, dms_stream st) { return dms_collab_allgather((dms_collab*)c, s, r, b, st); }
static int rccl_send(void* c, const void* s, size_t b, int p, dms_stream st) { return dms_collab_send((dms_collab*)c, s, b, p, st); }
static int rccl_recv(void* c, void* d, size_t b, int p, dms_stream st) { return dms_collab_recv((dms_collab*)c, d, b, p, st); }
static int rccl_bcast(void* c, void* buf, size_t b, int root, dms_stream st) {
  // a broadcast of a few dozen bytes as an all-gather of every rank's copy: the root's is kept
  dms_collab* cc = (dms_collab*)c;
  const int w = dms_collab_size(cc);
  void* all = nullptr;
  int rc = dms_device_alloc(&all, b * (size_t)w);
  if (rc) return rc;
  rc = dms_collab_allgather(cc, buf, all, b, st);
  if (!rc) rc = dms_memcpy_d2d_async(buf, (const char*)all + (size_t)root * b, b, st);
  if (!rc) rc = dms_stream_sync(st);
  dms_device_free(all);
  return rc;
}
int dms_transport_rccl(dms_collab* c, dms_transport* out) {
  DMS_REQUIRE(c && out, "null argument");
  out->ctx = c;
  out->rank = dms_collab_rank(c);
  out->world = dms_collab_size(c);
  out->allgather = rccl_allgather;
  out->send = rccl_send;
  out->recv = rccl_recv;
  out->broadcast = rccl_bcast;
  return DMS_OK;
}

int dms_session_create(dms_session** out, const dms_session_params* p, const dms_transport* t) {
  DMS_REQUIRE(out && p, "null argument");
  DMS_REQUIRE(p->n_cameras >= 1 && p->n_cameras <= DMS_MAX_SENSORS, "1 <= n_cameras <= DMS_MAX_SENSORS (a merged map holds every camera's time slot)");
  DMS_REQUIRE(p->camera.num_sensors >= p->n_cameras, "num_sensors < n_cameras: a merge would fail at join / import");
  DMS_REQUIRE(!t || (t->allgather && t->send && t->recv && t->world >= 1 && t->rank >= 0 && t->rank < t->world), "incomplete transport");
  dms_session* s = new dms_session();
  s->p = *p;
  if (t) {
    s->t = *t;
    s->rank = t->rank;
    s->world = t->world;
    s->local_only = false;  // (a one-rank communicator still carries the collectives: the RCCL calls are exercised on a one-GPU box)
  }
  if (const char* e = getenv("DMS_SESSION_MAP_STREAMS")) s->n_map_streams = std::max(0, std::min(16, atoi(e)));
  if (const char* e = getenv("DMS_SESSION_FUSED_BLOCK")) s->fused_block = atoi(e) != 0;
  if (const char* e = getenv("DMS_SESSION_JOIN_BY_FRAME_EVENT")) s->join_by_frame_event = atoi(e) != 0;
  if (const char* e = getenv("DMS_SESSION_SHARE_DEVICE")) s->share_device = atoi(e) != 0;
  s->n = p->n_cameras;
  s->W = p->camera.width;
  s->H = p->camera.height;
  s->thumb_bytes = dms_thumb_block_bytes(s->W, s->H);
  s->block_bytes = (s->thumb_bytes + kMetaBytes + 255) & ~(size_t)255;  // (every block 16-byte aligned whatever the image size: 1241 x 376)
  s->frame_of.resize(s->n);
  int rc = DMS_OK;
  for (int c = 0; c < s->n && !rc; ++c) {
    s->frame_of[c] = c;
    s->host_of_frame[c] = c % s->world;
    if (c % s->world == s->rank) {
      rc = make_camera(s, c, s->cams[c]);
      if (!rc) rc = make_ferns(s, &s->ferns[c]);
    }
  }
  const size_t N = (size_t)s->W * s->H;
  void* d = nullptr;
  auto alloc = [&](size_t bytes) -> void* {
    if (rc) return nullptr;
    d = nullptr;
    rc = dms_device_alloc(&d, bytes);
    if (!rc) rc = dms_memset(d, 0, bytes, nullptr);
    return d;
  };
  s->d_local = (unsigned char*)alloc((size_t)s->n * s->block_bytes);
  s->d_gathered = (unsigned char*)alloc((size_t)s->world * s->n * s->block_bytes);
  s->d_table = (float*)alloc((size_t)(1 + s->world) * (s->n + 1) * s->n * kRow * 4);
  s->d_pose = (float*)alloc(64);
  s->d_small = (float*)alloc((size_t)(1 + s->world) * 32 * 4);  // a rank's status + result, and everybody's (refine)
  s->d_frame_rgb = alloc(N * 3);
  s->d_frame_depth = (unsigned short*)alloc(N * 2);
  s->d_tex = (unsigned char*)alloc(N * 36);
  s->tail_off = (s->thumb_bytes + 15) & ~(size_t)15;
  s->ablock_bytes = (s->tail_off + kTailBytes + 255) & ~(size_t)255;
  if (s->world * s->n <= kMaxQueries) {  // (larger sessions run the synchronous step only)
    for (int b = 0; b < 2; ++b) {
      s->d_alocal[b] = (unsigned char*)alloc((size_t)s->n * s->ablock_bytes);
      s->d_agathered[b] = (unsigned char*)alloc((size_t)s->world * s->n * s->ablock_bytes);
      if (!rc && hipHostMalloc((void**)&s->ring[b].host, (size_t)s->world * s->n * kTailHostBytes) != hipSuccess) rc = DMS_ERR_HIP;
      if (!rc && hipEventCreateWithFlags(&s->ring[b].done, hipEventDisableTiming) != hipSuccess) rc = DMS_ERR_HIP;
      if (!rc && p->time_exchange && (hipEventCreate(&s->ring[b].ag0) != hipSuccess || hipEventCreate(&s->ring[b].ag1) != hipSuccess)) rc = DMS_ERR_HIP;
    }
    s->can_pipeline = !rc;
  }
  if (rc) {
    dms_session_destroy(s);
    return rc;
  }
  *out = s;
  return DMS_OK;
}

int dms_session_destroy(dms_session* s) {
  if (!s) return DMS_OK;
  (void)hipDeviceSynchronize();  // (pipelined ticks may still be writing the host mirrors and the block sets freed below)
  // joined cameras first: their owner must outlive them
  for (auto it = s->cams.begin(); it != s->cams.end();) {
    if (s->frame_of[it->first] != it->first) {
      free_camera(it->second);
      it = s->cams.erase(it);
    } else {
      ++it;
    }
  }
  for (auto& kv : s->cams) free_camera(kv.second);
  for (auto& kv : s->ferns) dms_ferns_destroy(kv.second);
  for (auto& kv : s->refiners) dms_refframe_destroy(kv.second);
  for (auto& ms : s->map_streams) {
    if (ms.blocks) (void)hipEventDestroy(ms.blocks);
    if (ms.s) (void)hipStreamDestroy(ms.s);
  }
  for (int b = 0; b < 2; ++b) {
    if (s->ring[b].host) (void)hipHostFree(s->ring[b].host);
    if (s->ring[b].done) (void)hipEventDestroy(s->ring[b].done);
    if (s->ring[b].ag0) (void)hipEventDestroy(s->ring[b].ag0);
    if (s->ring[b].ag1) (void)hipEventDestroy(s->ring[b].ag1);
  }
  void* bufs[] = {s->d_local, s->d_gathered, s->d_table, s->d_pose, s->d_small, s->d_frame_rgb, s->d_frame_depth, s->d_tex,
                  s->d_alocal[0], s->d_alocal[1], s->d_agathered[0], s->d_agathered[1]};
  for (void* b : bufs)
    if (b) dms_device_free(b);
  delete s;
  return DMS_OK;
}

int dms_session_step(dms_session* s, int k, const void* const* rgb_dev, const unsigned short* const* depth_dev, dms_stream st) {
  DMS_REQUIRE(s && rgb_dev && depth_dev, "null argument");
  const size_t N = (size_t)s->W * s->H;
  int rc = DMS_OK;
  if ((rc = drain_entries(s))) return rc;  // (pipelined ticks before this one: their pose-graph rows come first)
  invalidate_searches(s, k);               // (and their searches wake nothing: this tick queries every pair itself)
  s->last_stream = st;
  if ((rc = apply_tracker_policy(s, 1))) return rc;  // (every frame of a synchronous tick runs on the caller's stream)
  if (s->p.query_inside_frame) return step_inside(s, k, rgb_dev, depth_dev, st);
  // 1 + 2. forward the frames of cameras hosted elsewhere; every hosted camera's frame, in id order.  (A frame that arrives from
  // another rank is processed when its camera's turn comes: receive and process are interleaved in camera order on both sides.)
  std::map<int, int> read_index;  // camera read here -> position in rgb_dev / depth_dev
  {
    int i = 0;
    for (int c = 0; c < s->n; ++c)
      if (c % s->world == s->rank) read_index[c] = i++;
  }
  for (int c = 0; c < s->n; ++c) {
    const int src = c % s->world, host = host_of_camera(s, c);
    if (src != host && src == s->rank) {
      const int i = read_index.at(c);
      if ((rc = t_send(s, rgb_dev[i], N * 3, host, st)) || (rc = t_send(s, depth_dev[i], N * 2, host, st)) || (rc = sync(st))) return rc;
    }
    if (host != s->rank) continue;
    // Nothing waits for the host between two cameras' frames: they pipeline on the stream like the frames of one camera (each
    // context's live half on its own prep stream, beside the previous frame).  A frame read here is processed where the caller put
    // it (the step returns after the fetch below, so the buffer outlives its use; a migration inside this step ships it from there)
    // ; a forwarded frame is received into the camera's own buffer.
    Camera& cam = s->cams.at(c);
    if (src == s->rank) {
      const int i = read_index.at(c);
      if ((rc = dms_fusion_process_frame(cam.f, rgb_dev[i], 3, depth_dev[i], nullptr, 1.f, st))) return rc;
      cam.frame_rgb = rgb_dev[i];
      cam.frame_depth = depth_dev[i];
    } else {
      cam.frame_rgb = cam.last_rgb;
      cam.frame_depth = cam.last_depth;
      if ((rc = t_recv(s, cam.last_rgb, N * 3, src, st)) || (rc = t_recv(s, cam.last_depth, N * 2, src, st))) return rc;
      if ((rc = dms_fusion_inputs_ready(cam.f, st))) return rc;  // (a stream-ordered transport: the frame's ingest, on the context's own stream, behind the receive)
      if ((rc = dms_fusion_process_frame(cam.f, cam.last_rgb, 3, cam.last_depth, nullptr, 1.f, st))) return rc;
    }
  }
  for (auto& kv : s->cams) {  // results (the first fetch waits for the stream, the others find it drained)
    Camera& cam = kv.second;
    const int tick_before = cam.tick;
    rc = dms_fusion_fetch(cam.f, &cam.last, st);
    if (rc && rc != DMS_ERR_CAPACITY) return rc;
    cam.tick = cam.last.tick;
    memcpy(cam.pose, cam.last.pose, 64);
    cam.pg_tick.push_back(tick_before);
    cam.pg_pose.insert(cam.pg_pose.end(), cam.pose, cam.pose + 16);
    cam.has_frame = true;
  }
  // 3. publish: own map's database, then the all-gather (slots per rank = the most cameras any rank hosts)
  std::vector<int> host_counts(s->world, 0);
  for (int c = 0; c < s->n; ++c) host_counts[host_of_camera(s, c)] += 1;
  const int slots = *std::max_element(host_counts.begin(), host_counts.end());
  if ((rc = dms_memset(s->d_local, 0, (size_t)slots * s->block_bytes, st))) return rc;
  {
    int i = 0;
    for (auto& kv : s->cams) {
      const int c = kv.first;
      Camera& cam = kv.second;
      unsigned char* blk = s->d_local + (size_t)i * s->block_bytes;
      if ((rc = dms_fusion_thumbnails(cam.f, blk, st))) return rc;
      float meta[kMetaBytes / 4];
      memset(meta, 0, sizeof(meta));
      const int ids[2] = {c, cam.tick};
      memcpy(meta, ids, 8);
      memcpy(meta + 4, cam.pose, 64);
      if ((rc = dms_memcpy_h2d(blk + s->thumb_bytes, meta, kMetaBytes, st))) return rc;
      if (!cam.last.lost) {  // processFerns sits under `if (!lost)` (ElasticFusion.cpp:588-591)
        if ((rc = dms_ferns_add_frame_async(s->ferns.at(s->frame_of[c]), nullptr, nullptr, nullptr, blk, nullptr, (const float*)(blk + s->thumb_bytes + 16),
                                            cam.tick, s->p.fern_threshold, st)))
          return rc;
      }
      ++i;
    }
  }
  const size_t per_rank = (size_t)slots * s->block_bytes;
  if ((rc = t_allgather(s, s->d_local, s->d_gathered, per_rank, st)) || (rc = sync(st))) return rc;
  std::map<int, std::vector<float>> poses;
  std::map<int, int> ticks;
  std::map<int, const unsigned char*> blocks;
  for (int r = 0; r < s->world; ++r)
    for (int i = 0; i < host_counts[r]; ++i) {
      const unsigned char* raw = s->d_gathered + (size_t)r * per_rank + (size_t)i * s->block_bytes;
      float meta[kMetaBytes / 4];
      if ((rc = dms_memcpy_d2h(meta, raw + s->thumb_bytes, kMetaBytes, st))) return rc;
      int ids[2];
      memcpy(ids, meta, 8);
      DMS_REQUIRE(ids[0] >= 0 && ids[0] < s->n, "a gathered block names no camera of this session");
      blocks[ids[0]] = raw;
      ticks[ids[0]] = ids[1];
      poses[ids[0]] = std::vector<float>(meta + 4, meta + 20);
    }
  return decide_and_merge(s, k, ~0ull, poses, ticks, blocks, st);
}


int dms_session_step_async(dms_session* s, int k, const void* const* rgb_dev, const unsigned short* const* depth_dev, dms_stream st) {
  DMS_REQUIRE(s && rgb_dev && depth_dev, "null argument");
  DMS_REQUIRE(s->can_pipeline, "the pipelined step serves world * n_cameras <= 64");
  DMS_REQUIRE(!s->p.camera.reloc, "the pipelined step does not read the tracker's verdict: relocalisation must be off");
  DMS_REQUIRE(!s->p.query_inside_frame, "query_inside_frame serialises the cameras of a tick: dms_session_step only");
  const size_t N = (size_t)s->W * s->H;
  hipStream_t hs = (hipStream_t)st;
  int rc = DMS_OK;
  s->last_stream = st;
  s->async_ticks += 1;
  // 0. what the device finished two ticks ago: pose-graph rows, and whether a descriptor search (of tick k - 3) hit
  {
    dms_session::Entry& e = s->ring[k & 1];
    const dms_session::Entry& e1 = s->ring[(k + 1) & 1];
    DMS_REQUIRE((e.consumed || e.tick == k - 2) && (e1.consumed || e1.tick == k - 1), "dms_session_step_async: ticks must be consecutive");
    if ((rc = consume_entry(s, e, true))) return rc;
  }
  s->wake_ticks.erase(s->wake_ticks.begin(), s->wake_ticks.lower_bound(k));
  const bool wake = s->wake_ticks.count(k) != 0;
  // 1 + 2. frames, as the synchronous step enqueues them - each hosted map's cameras on that map's stream; nothing is fetched.  A
  // camera's frame block (thumbnails, pose, tick: one launch) is packed right behind its frame.
  std::map<int, int> read_index;
  {
    int i = 0;
    for (int c = 0; c < s->n; ++c)
      if (c % s->world == s->rank) read_index[c] = i++;
  }
  const int slots = slots_now(s);
  const std::vector<int> mine = cams_of_rank(s, s->rank);
  unsigned char* local = s->d_alocal[k & 1];
  // (a one-rank session without a transport gathers nothing: the search reads the blocks where they were packed)
  unsigned char* gathered = s->local_only ? local : s->d_agathered[k & 1];
  const size_t T0 = s->tail_off, B = s->ablock_bytes;
  {  // how many streams this tick's frames run on
    std::set<hipStream_t> streams;
    for (auto& kv : s->cams) {
      hipStream_t fs = hs;
      dms_session::MapStream* ms = nullptr;
      if ((rc = stream_of_map(s, s->frame_of[kv.first], hs, &fs, &ms))) return rc;
      streams.insert(fs);
    }
    if ((rc = apply_tracker_policy(s, (int)streams.size()))) return rc;
  }
  std::map<int, hipStream_t> cam_stream;
  std::set<dms_session::MapStream*> used;
  std::map<dms_session::MapStream*, int> last_cam_of;   // the camera whose frame is the stream's last this tick
  std::set<dms_session::MapStream*> own_marker;         // a launch of the session's own follows a frame there: its event must be recorded
  for (int c = 0; c < s->n; ++c) {
    const int src = c % s->world, host = host_of_camera(s, c);
    if (src != host && src == s->rank) {
      const int i = read_index.at(c);
      if ((rc = t_send(s, rgb_dev[i], N * 3, host, st)) || (rc = t_send(s, depth_dev[i], N * 2, host, st))) return rc;
    }
    if (host != s->rank) continue;
    Camera& cam = s->cams.at(c);
    hipStream_t fs = hs;
    dms_session::MapStream* ms = nullptr;
    if ((rc = stream_of_map(s, s->frame_of[c], hs, &fs, &ms))) return rc;
    cam_stream[c] = fs;
    // The camera's frame block (thumbnails, pose, tick) is written by the frame's own last kernel (dms_fusion_arm_frame_block): no
    // launch of its own behind the frame - unless the frame took a path without the fused fill-in.
    unsigned char* blk = local + (size_t)(std::find(mine.begin(), mine.end(), c) - mine.begin()) * B;
    if (!wake && s->fused_block && (rc = dms_fusion_arm_frame_block(cam.f, blk, (float*)(blk + T0 + kTailPose), (int*)(blk + T0 + kTailTick), cam.tick + 1))) return rc;
    if (src == s->rank) {
      const int i = read_index.at(c);
      if ((rc = dms_fusion_process_frame(cam.f, rgb_dev[i], 3, depth_dev[i], nullptr, 1.f, (dms_stream)fs))) return rc;
      cam.frame_rgb = rgb_dev[i];
      cam.frame_depth = depth_dev[i];
    } else {
      cam.frame_rgb = cam.last_rgb;
      cam.frame_depth = cam.last_depth;
      if ((rc = t_recv(s, cam.last_rgb, N * 3, src, st)) || (rc = t_recv(s, cam.last_depth, N * 2, src, st))) return rc;
      if ((rc = dms_fusion_inputs_ready(cam.f, st))) return rc;
      if ((rc = dms_fusion_process_frame(cam.f, cam.last_rgb, 3, cam.last_depth, nullptr, 1.f, st))) return rc;
    }
    if (!wake) {
      cam.tick += 1;  // ElasticFusion.cpp:588-591 (the camera is never lost without relocalisation)
      const bool own_launch = !dms_fusion_frame_block_written(cam.f);
      if (own_launch && (rc = dms_fusion_frame_block(cam.f, blk, (float*)(blk + T0 + kTailPose), (int*)(blk + T0 + kTailTick), cam.tick, (dms_stream)fs)))
        return rc;
      if (ms) {
        used.insert(ms);
        last_cam_of[ms] = c;
        if (own_launch || src != s->rank) own_marker.insert(ms);  // (a forwarded camera's frame runs on the caller's stream: ms is null there anyway)
      }
    }
  }
  for (dms_session::MapStream* ms : used) {  // the exchange below reads the blocks on the caller's stream
    // (when the frames wrote their blocks themselves, the stream's last frame's own completion event is the join: no marker behind it)
    if (!own_marker.count(ms) && s->join_by_frame_event) {
      if ((rc = dms_fusion_wait_frame_done(s->cams.at(last_cam_of.at(ms)).f, st))) return rc;
      continue;
    }
    if (hipEventRecord(ms->blocks, ms->s) != hipSuccess || hipStreamWaitEvent(hs, ms->blocks, 0) != hipSuccess) {
      set_error("dms_session_step_async: joining a map's stream failed");
      return DMS_ERR_HIP;
    }
  }
  if (wake) {
    // a woken tick is a synchronous one: the previous tick's rows first, then this tick's results from the contexts
    if ((rc = consume_entry(s, s->ring[(k + 1) & 1], true))) return rc;
    s->wakes += 1;
    for (auto& kv : s->cams) {
      Camera& cam = kv.second;
      const int tick_before = cam.tick;
      rc = dms_fusion_fetch(cam.f, &cam.last, (dms_stream)cam_stream.at(kv.first));  // (synchronises the stream the frame ran on)
      if (rc && rc != DMS_ERR_CAPACITY) return rc;
      cam.tick = cam.last.tick;
      memcpy(cam.pose, cam.last.pose, 64);
      cam.pg_tick.push_back(tick_before);
      cam.pg_pose.insert(cam.pg_pose.end(), cam.pose, cam.pose + 16);
      cam.has_frame = true;
    }
  }
  // 3. publish: descriptor + key-frame insertion read the block in place
  if (s->ids_tick != (int)s->merges.size()) {  // the camera id of every slot, in both block sets: constant until the placement changes
    for (int b = 0; b < 2; ++b)
      for (int i = 0; i < slots; ++i) {
        const int c = i < (int)mine.size() ? mine[i] : -1;
        if ((rc = dms_memcpy_h2d(s->d_alocal[b] + (size_t)i * B + T0 + kTailCam, &c, 4, st))) return rc;
      }
    s->ids_tick = (int)s->merges.size();
  }
  bool any_search = false;
  for (auto& kv : s->ferns)
    for (int a = 0; a < s->n; ++a) any_search = any_search || s->frame_of[a] != kv.first;
  const bool publish_mirrors = s->local_only && !any_search && (int)mine.size() == slots;
  for (int i = 0; i < slots; ++i) {
    unsigned char* blk = local + (size_t)i * B;
    if (i >= (int)mine.size()) {  // an empty slot: no good codes (the search passes it over); its id and hit rows stay
      if (hipMemsetAsync(blk + T0 + kTailGood, 0, 4, hs) != hipSuccess) return DMS_ERR_HIP;
      continue;
    }
    const int c = mine[i];
    Camera& cam = s->cams.at(c);
    if (wake && (rc = dms_fusion_frame_block(cam.f, blk, (float*)(blk + T0 + kTailPose), (int*)(blk + T0 + kTailTick), cam.tick, st))) return rc;
    // (a one-process session whose cameras all share one map searches nothing: the key-frame insertion's own launch then hands
    // this block's tail to the host mirror - no launch is left behind the frame but this one)
    void* mirror = publish_mirrors ? (void*)(s->ring[k & 1].host + (size_t)i * kTailHostBytes) : nullptr;
    if ((rc = dms_ferns_publish_block_mirror(s->ferns.at(s->frame_of[c]), blk, blk + T0 + kTailCodes, (int*)(blk + T0 + kTailGood),
                                             (const float*)(blk + T0 + kTailPose), cam.tick, s->p.fern_threshold, mirror, T0 + kTailGood,
                                             mirror ? (size_t)kTailHostBytes : 0, st)))
      return rc;
  }
  const size_t per_rank = (size_t)slots * B;
  dms_session::Entry& e = s->ring[k & 1];
  if (!s->local_only) {
    const bool timed = e.ag0 != nullptr;
    if (timed && hipEventRecord(e.ag0, hs) != hipSuccess) return DMS_ERR_HIP;
    if ((rc = t_allgather(s, local, gathered, per_rank, st))) return rc;
    if (timed && hipEventRecord(e.ag1, hs) != hipSuccess) return DMS_ERR_HIP;
    e.timed = timed;
  }
  // 4a. every hosted database against every gathered block; the hit rows ride in the NEXT tick's blocks.  The tails come to the host
  // beside the next tick (read at the start of tick k + 2), written into mapped memory by the first search's second launch (a launch
  // of its own when no database here has anybody left to be queried by).
  {
    int d = 0;
    bool mirrored = false;
    unsigned char* next_local = s->d_alocal[(k + 1) & 1];
    for (auto& kv : s->ferns) {
      // (a database whose map already holds every camera has nobody to be queried by - and never will again: frames only merge)
      bool anyone = false;
      for (int a = 0; a < s->n; ++a) anyone = anyone || s->frame_of[a] != kv.first;
      if (anyone) {
        if ((rc = dms_ferns_search_blocks_hd_mirror(kv.second, gathered, B, s->world * slots, T0 + kTailCodes, T0 + kTailGood, 0, s->p.inter_map ? 1 : 0,
                                                    (int*)(next_local + (size_t)d * B + T0 + kTailHits), mirrored ? nullptr : e.host, T0 + kTailGood,
                                                    kTailHostBytes, st)))
          return rc;
        mirrored = true;
      }
      ++d;
    }
    if (!mirrored && !publish_mirrors && (rc = dms_copy_rows_async(e.host, kTailHostBytes, gathered + T0 + kTailGood, B, kTailHostBytes, (size_t)s->world * slots, st))) return rc;
  }
  if (hipEventRecord(e.done, hs) != hipSuccess) {
    set_error("dms_session_step_async: mirroring the gathered tails failed");
    return DMS_ERR_HIP;
  }
  e.tick = k;
  e.consumed = false;
  e.slots = slots;
  if (!wake) return DMS_OK;
  // 4 - 6. the reference's inter-map block for this tick, on this tick's blocks
  if ((rc = consume_entry(s, e, false))) return rc;
  std::map<int, std::vector<float>> poses;
  std::map<int, int> ticks;
  std::map<int, const unsigned char*> blocks;
  for (int r = 0; r < s->world; ++r) {
    const std::vector<int> cs = cams_of_rank(s, r);
    for (size_t i = 0; i < cs.size(); ++i) {
      const unsigned char* t = entry_tail(e, r, (int)i);
      int tick, cam_id;
      memcpy(&tick, t + (kTailTick - kTailGood), 4);
      memcpy(&cam_id, t + (kTailCam - kTailGood), 4);
      DMS_REQUIRE(cam_id == cs[i], "a gathered block is not the camera the placement names");
      const float* pose = (const float*)(t + (kTailPose - kTailGood));
      blocks[cam_id] = gathered + (size_t)r * per_rank + i * B;
      ticks[cam_id] = tick;
      poses[cam_id] = std::vector<float>(pose, pose + 16);
    }
  }
  const size_t merges_before = s->merges.size();
  if ((rc = decide_and_merge(s, k, ~0ull, poses, ticks, blocks, st))) return rc;
  if (s->merges.size() != merges_before) invalidate_searches(s, k);  // (the searches in flight ran on the placement before the merge)
  return DMS_OK;
}

int dms_session_sync(dms_session* s) {
  DMS_REQUIRE(s, "null argument");
  return drain_entries(s);
}

int dms_session_async_stats(dms_session* s, int* ticks, int* wakes) {
  DMS_REQUIRE(s, "null argument");
  if (ticks) *ticks = s->async_ticks;
  if (wakes) *wakes = s->wakes;
  return DMS_OK;
}

int dms_session_exchange_time(dms_session* s, double* allgather_ms_sum, int* allgathers) {
  DMS_REQUIRE(s, "null argument");
  if (allgather_ms_sum) *allgather_ms_sum = s->ag_ms;
  if (allgathers) *allgathers = s->ag_n;
  return DMS_OK;
}

int dms_session_frame_of(dms_session* s, int* frame_of) {
  DMS_REQUIRE(s && frame_of, "null argument");
  for (int c = 0; c < s->n; ++c) frame_of[c] = s->frame_of[c];
  return DMS_OK;
}
int dms_session_host_of_frame(dms_session* s, int frame) {
  if (!s) return -1;
  auto it = s->host_of_frame.find(frame);
  return it == s->host_of_frame.end() ? -1 : it->second;
}
int dms_session_num_merges(dms_session* s) { return s ? (int)s->merges.size() : 0; }
int dms_session_get_merge(dms_session* s, int i, int* k, int* fb, int* fa, float* T16) {
  DMS_REQUIRE(s && i >= 0 && i < (int)s->merges.size(), "no such merge");
  const Merge& m = s->merges[i];
  if (k) *k = m.k;
  if (fb) *fb = m.fb;
  if (fa) *fa = m.fa;
  if (T16) memcpy(T16, m.T, 64);
  return DMS_OK;
}
int dms_session_num_refinements(dms_session* s) { return s ? (int)s->refinements.size() : 0; }
int dms_session_get_refinement(dms_session* s, int i, int* k, int* camera, int* frame, int* accepted) {
  DMS_REQUIRE(s && i >= 0 && i < (int)s->refinements.size(), "no such refinement");
  const Refinement& r = s->refinements[i];
  if (k) *k = r.k;
  if (camera) *camera = r.a;
  if (frame) *frame = r.fb;
  if (accepted) *accepted = r.accepted;
  return DMS_OK;
}
int dms_session_hosted(dms_session* s, int* cameras, int max, int* n) {
  DMS_REQUIRE(s && n, "null argument");
  int i = 0;
  for (auto& kv : s->cams) {
    if (cameras && i < max) cameras[i] = kv.first;
    ++i;
  }
  *n = i;
  return DMS_OK;
}
dms_fusion* dms_session_camera(dms_session* s, int camera) {
  if (!s) return nullptr;
  auto it = s->cams.find(camera);
  return it == s->cams.end() ? nullptr : it->second.f;
}
dms_ferns* dms_session_ferns(dms_session* s, int frame) {
  if (!s) return nullptr;
  auto it = s->ferns.find(frame);
  return it == s->ferns.end() ? nullptr : it->second;
}
int dms_session_last_result(dms_session* s, int camera, dms_frame_result* r) {
  DMS_REQUIRE(s && r && s->cams.count(camera), "the camera is not hosted here");
  *r = s->cams.at(camera).last;
  return DMS_OK;
}
int dms_session_add_relative_constraint(dms_session* s, int camera, const float* src3, const float* target3) {
  DMS_REQUIRE(s && src3 && target3 && s->cams.count(camera), "the camera is not hosted here");
  Camera& cam = s->cams.at(camera);
  cam.rel_cons.insert(cam.rel_cons.end(), src3, src3 + 3);
  cam.rel_cons.insert(cam.rel_cons.end(), target3, target3 + 3);
  return DMS_OK;
}
int dms_session_relative_constraints(dms_session* s, int camera, float* rows6, int max, int* n) {
  DMS_REQUIRE(s && n && s->cams.count(camera), "the camera is not hosted here");
  const Camera& cam = s->cams.at(camera);
  *n = (int)(cam.rel_cons.size() / 6);
  if (rows6)
    for (int i = 0; i < *n && i < max; ++i) memcpy(rows6 + (size_t)i * 6, &cam.rel_cons[(size_t)i * 6], 24);
  return DMS_OK;
}
int dms_session_pose_graph(dms_session* s, int camera, int* ticks, float* poses16, int max, int* n) {
  DMS_REQUIRE(s && n && s->cams.count(camera), "the camera is not hosted here");
  int rc = drain_entries(s);  // (rows of pipelined ticks the device still owes)
  if (rc) return rc;
  const Camera& cam = s->cams.at(camera);
  *n = (int)cam.pg_tick.size();
  for (int i = 0; i < *n && i < max; ++i) {
    if (ticks) ticks[i] = cam.pg_tick[i];
    if (poses16) memcpy(poses16 + (size_t)i * 16, &cam.pg_pose[(size_t)i * 16], 64);
  }
  return DMS_OK;
}

}  // extern "C"
