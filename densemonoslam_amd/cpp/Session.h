// C++ face of the collaborative session (include/dmslam_session.h) for a front end that spreads its cameras over the GPUs of a node.
//
// The reference drives every camera of a session from one loop in one process (GUI/src/MainController.cpp:262-400:
// `for (reader : logReaders) { reader->getNext(); eFusion->processFrame(reader->name(), ...); }`) and would let a camera's frame close
// a loop against another map inside processFrame (ElasticFusion.cpp:595-632, ReferenceFrame::resolveRelativeTransformationFern /
// consumeReferenceFrame).  dms::Session is that loop for ONE rank of `world` processes, one per GPU: every rank constructs the same
// object and calls step() with the same tick index; cameras migrate between ranks as their maps merge.
//
//   reference                                              -> here
//   MainController::run's camera loop, one iteration        -> Session::step(k, rgb, depth)            (dms_session_step)
//   ... for a live front end that must not wait for the GPU -> Session::stepPipelined(k, rgb, depth)   (dms_session_step_async)
//   ElasticFusion::referenceFrames() / whichReferenceFrame  -> Session::frameOf()
//   Context::poseGraph(), Context::relativeCons()           -> Session::poseGraph(camera), relativeCons(camera)
//
// Header-only over the C ABI; no HIP, torch or Eigen types.
#pragma once
#include <array>
#include <utility>
#include <vector>

#include "../../include/dmslam_session.h"
#include "dmslam.hpp"

namespace dms {

class Session {
 public:
  struct Merge {
    int tick, consumingFrame, consumedFrame;
    std::array<float, 16> relativeTransform;  // row-major: the consumed map's coordinates into the consuming map's
  };
  struct Refinement {
    int tick, camera, frame;
    bool accepted;
  };

  // Options' defaults for a session of n cameras (fernThresh, interMapPhotoThresh, covThresh, icpErrThresh, icpCountThresh ...);
  // edit the returned struct (params.camera is every camera's dms_fusion_params template) before constructing the session
  static dms_session_params defaults(int cameras, int width, int height, float fx, float fy, float cx, float cy) {
    dms_session_params p;
    dms_session_default_params(&p, cameras, width, height, fx, fy, cx, cy);
    return p;
  }

  // transport == nullptr: every camera in this process (the reference's own arrangement)
  explicit Session(const dms_session_params& p, const dms_transport* transport = nullptr) : n_(p.n_cameras) {
    check(dms_session_create(&h_, &p, transport), "dms_session_create");
  }
  // one process per GPU over RCCL: `comm` from dms_collab_create (include/dmslam_collab.h), same world size on every rank
  Session(const dms_session_params& p, dms_collab* comm) : n_(p.n_cameras) {
    dms_transport t;
    check(dms_transport_rccl(comm, &t), "dms_transport_rccl");
    check(dms_session_create(&h_, &p, &t), "dms_session_create");
  }
  ~Session() { dms_session_destroy(h_); }
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;

  // rgb[i] / depth[i]: the frame (RGB8 W x H x 3, depth u16 W x H, in this device's HBM) of the i-th camera READ on this rank - the
  // cameras c with c % world == rank, ascending.  The tick synchronises where the reference's loop does.
  void step(int k, const std::vector<const void*>& rgb, const std::vector<const unsigned short*>& depth, dms_stream s = nullptr) {
    check(dms_session_step(h_, k, rgb.data(), depth.data(), s), "dms_session_step");
  }
  // The same tick without a host synchronisation between descriptor hits (cameras need reloc = 0): the buffers of tick k must stay
  // untouched until the call of tick k + 2 has returned; the full inter-map query runs three ticks after the search that hit.
  void stepPipelined(int k, const std::vector<const void*>& rgb, const std::vector<const unsigned short*>& depth, dms_stream s = nullptr) {
    check(dms_session_step_async(h_, k, rgb.data(), depth.data(), s), "dms_session_step_async");
  }
  void sync() { check(dms_session_sync(h_), "dms_session_sync"); }

  int cameras() const { return n_; }
  std::vector<int> frameOf() const {  // camera -> reference frame (the id of the frame's founding camera); the same on every rank
    std::vector<int> v(n_);
    check(dms_session_frame_of(h_, v.data()), "dms_session_frame_of");
    return v;
  }
  int hostOfFrame(int frame) const { return dms_session_host_of_frame(h_, frame); }
  std::vector<int> hosted() const {  // cameras served on this rank
    std::vector<int> v(n_);
    int n = 0;
    check(dms_session_hosted(h_, v.data(), n_, &n), "dms_session_hosted");
    v.resize(n);
    return v;
  }
  std::vector<Merge> merges() const {
    std::vector<Merge> v(dms_session_num_merges(h_));
    for (size_t i = 0; i < v.size(); ++i)
      check(dms_session_get_merge(h_, (int)i, &v[i].tick, &v[i].consumingFrame, &v[i].consumedFrame, v[i].relativeTransform.data()), "dms_session_get_merge");
    return v;
  }
  std::vector<Refinement> refinements() const {
    std::vector<Refinement> v(dms_session_num_refinements(h_));
    for (size_t i = 0; i < v.size(); ++i) {
      int acc = 0;
      check(dms_session_get_refinement(h_, (int)i, &v[i].tick, &v[i].camera, &v[i].frame, &acc), "dms_session_get_refinement");
      v[i].accepted = acc != 0;
    }
    return v;
  }
  // a hosted camera's context (nullptr when it is served on another rank): dms_fusion_* for everything the frame step exposes
  dms_fusion* camera(int c) const { return dms_session_camera(h_, c); }
  dms_ferns* ferns(int frame) const { return dms_session_ferns(h_, frame); }
  dms_frame_result lastResult(int c) const {  // (the last FETCHED frame: every frame with step(), the last woken one with stepPipelined())
    dms_frame_result r;
    check(dms_session_last_result(h_, c, &r), "dms_session_last_result");
    return r;
  }
  // Context::poseGraph(): (tick before the frame, pose after it) per processed frame of a hosted camera, re-based by every merge
  std::vector<std::pair<int, std::array<float, 16>>> poseGraph(int c) const {
    int n = 0;
    check(dms_session_pose_graph(h_, c, nullptr, nullptr, 0, &n), "dms_session_pose_graph");
    std::vector<int> t(n > 0 ? n : 1);
    std::vector<float> p((size_t)(n > 0 ? n : 1) * 16);
    check(dms_session_pose_graph(h_, c, t.data(), p.data(), n, &n), "dms_session_pose_graph");
    std::vector<std::pair<int, std::array<float, 16>>> v(n);
    for (int i = 0; i < n; ++i) {
      v[i].first = t[i];
      for (int j = 0; j < 16; ++j) v[i].second[j] = p[(size_t)i * 16 + j];
    }
    return v;
  }
  void addRelativeConstraint(int c, const float* src3, const float* target3) {
    check(dms_session_add_relative_constraint(h_, c, src3, target3), "dms_session_add_relative_constraint");
  }
  std::vector<std::array<float, 6>> relativeCons(int c) const {
    int n = 0;
    check(dms_session_relative_constraints(h_, c, nullptr, 0, &n), "dms_session_relative_constraints");
    std::vector<std::array<float, 6>> v(n);
    if (n) check(dms_session_relative_constraints(h_, c, v[0].data(), n, &n), "dms_session_relative_constraints");
    return v;
  }
  void pipelinedStats(int* ticks, int* woken) const { check(dms_session_async_stats(h_, ticks, woken), "dms_session_async_stats"); }
  // with params.time_exchange: device time of the pipelined ticks' all-gathers seen complete so far (sum in ms, how many)
  void exchangeTime(double* allgather_ms_sum, int* allgathers) const { check(dms_session_exchange_time(h_, allgather_ms_sum, allgathers), "dms_session_exchange_time"); }

  dms_session* handle() const { return h_; }

 private:
  dms_session* h_ = nullptr;
  int n_ = 0;
};

}  // namespace dms
