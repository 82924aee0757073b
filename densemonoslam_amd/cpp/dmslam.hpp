// C++ mirror of the reference's elasticfusion/Core classes over the C ABI of libdmslam_hip.so.
//
// Header-only; no Eigen / Pangolin / CUDA needed.  Matrices cross as plain row-major float
// arrays; when <Eigen/Core> is available the overloads at the bottom accept the reference's own
// argument types (Eigen::Vector3f, Eigen::Matrix<float,3,3,RowMajor>, Eigen::Matrix4f) so that
// GUI/src/MainController.cpp, Core/src/ElasticFusion.cpp and GPUTest.cpp call sites compile
// unchanged against these classes.
//
//   reference class / method                                   -> here
//   RGBDOdometry (Core/src/Utils/RGBDOdometry.h:32-153)        -> dms::RGBDOdometry
//   GPUTexture   (Core/src/GPUTexture.h:29-57)                 -> dms::DeviceTexture (pitched HBM image)
//   GlobalModel  (Core/src/GlobalModel.h:43-141)               -> dms::GlobalModel
//   IndexMap     (Core/src/IndexMap.h:33-205)                  -> dms::IndexMap
//   ElasticFusion::processFrame (ElasticFusion.h:92-100)       -> dms::ElasticFusion::processFrame
//   Ferns        (Core/src/Ferns.h:35-266)                     -> dms::Ferns
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dmslam.h"
#include "../../include/dmslam_ferns.h"
#include "../../include/dmslam_fusion.h"

namespace dms {

inline void check(int rc, const char* what) {
  if (rc != DMS_OK) throw std::runtime_error(std::string(what) + ": " + dms_last_error());
}

// Pitched device image; replaces GPUTexture (GL texture + cudaGraphicsResource) in the kept API.
class DeviceTexture {
 public:
  DeviceTexture(int width, int height, size_t elemBytes) : width(width), height(height), elem(elemBytes) {
    check(dms_device_alloc(&ptr, (size_t)width * height * elemBytes), "dms_device_alloc");
    view.data = ptr;
    view.pitch = (size_t)width * elemBytes;
    view.rows = height;
    view.cols = width;
  }
  // a view of an image the library owns (a context's fill-in / prediction textures, dms_fusion_get_image): nothing is allocated or freed
  DeviceTexture(const dms_image2d& existing, size_t elemBytes)
      : ptr(existing.data), view(existing), width(existing.cols), height(existing.rows), elem(elemBytes), owned(false) {}
  ~DeviceTexture() {
    if (owned) dms_device_free(ptr);
  }
  DeviceTexture(const DeviceTexture&) = delete;
  DeviceTexture& operator=(const DeviceTexture&) = delete;
  // GPUTexture::texture->Upload(ptr, format, type) (ElasticFusion.cpp:111-114, GPUTest.cpp:39,59)
  void Upload(const void* host, dms_stream s = nullptr) { check(dms_memcpy_h2d(ptr, host, (size_t)width * height * elem, s), "upload"); }
  void Download(void* host, dms_stream s = nullptr) const { check(dms_memcpy_d2h(host, ptr, (size_t)width * height * elem, s), "download"); }
  const dms_image2d* image() const { return &view; }
  void* ptr = nullptr;
  dms_image2d view;
  const int width, height;
  const size_t elem;
  const bool owned = true;
};

class RGBDOdometry {
 public:
  RGBDOdometry(int width, int height, float cx, float cy, float fx, float fy, float distThresh = 0.10f, float angleThresh = 0.0f) {
    check(dms_odometry_create(&h, width, height, cx, cy, fx, fy, distThresh, angleThresh), "dms_odometry_create");
  }
  virtual ~RGBDOdometry() { dms_odometry_destroy(h); }
  RGBDOdometry(const RGBDOdometry&) = delete;

  void initICP(DeviceTexture* filteredDepth, const float depthCutoff, dms_stream stream = nullptr) {
    check(dms_odometry_initICP_depth(h, filteredDepth->image(), depthCutoff, stream), "initICP");
  }
  void initICP(DeviceTexture* predictedVertices, DeviceTexture* predictedNormals, const float depthCutoff, dms_stream stream = nullptr) {
    check(dms_odometry_initICP_maps(h, (const float*)predictedVertices->ptr, (const float*)predictedNormals->ptr, depthCutoff, stream), "initICP");
  }
  void initICPModel(DeviceTexture* predictedVertices, DeviceTexture* predictedNormals, const float depthCutoff, const float* modelPose16,
                    dms_stream stream = nullptr) {
    check(dms_odometry_initICPModel(h, (const float*)predictedVertices->ptr, (const float*)predictedNormals->ptr, depthCutoff, modelPose16,
                                    stream),
          "initICPModel");
  }
  void initRGB(DeviceTexture* rgb, dms_stream stream = nullptr) { check(dms_odometry_initRGB(h, rgb->image(), stream), "initRGB"); }
  void initRGBModel(DeviceTexture* rgb, dms_stream stream = nullptr) { check(dms_odometry_initRGBModel(h, rgb->image(), stream), "initRGBModel"); }
  void initFirstRGB(DeviceTexture* rgb, dms_stream s = nullptr) { check(dms_odometry_initFirstRGB(h, rgb->image(), s), "initFirstRGB"); }

  // trans[3], rot[9] row-major, in/out — RGBDOdometry::getIncrementalTransformation (RGBDOdometry.cpp:268)
  void getIncrementalTransformation(float* trans, float* rot, const bool& rgbOnly, const float& icpWeight, const bool& pyramid,
                                    const bool& fastOdom, const bool& so3, const bool interMap = false, dms_stream stream = nullptr) {
    dms_track_result r;
    check(dms_odometry_getIncrementalTransformation(h, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap, &r, stream),
          "getIncrementalTransformation");
    lastICPError = r.lastICPError;
    lastICPCount = r.lastICPCount;
    lastRGBError = r.lastRGBError;
    lastRGBCount = r.lastRGBCount;
    lastSO3Error = r.lastSO3Error;
    lastSO3Count = r.lastSO3Count;
    for (int i = 0; i < 36; ++i) lastA[i] = r.lastA[i];
    for (int i = 0; i < 6; ++i) lastb[i] = r.lastb[i];
  }
  void getCovariance(double* cov36) { check(dms_odometry_getCovariance(h, cov36), "getCovariance"); }

  float lastICPError = 0, lastICPCount = 0, lastRGBError = 0, lastRGBCount = 0, lastSO3Error = 0, lastSO3Count = 0;
  double lastA[36] = {0};
  double lastb[6] = {0};
  dms_odometry* h = nullptr;
};

class GlobalModel {
 public:
  static const int TEXTURE_DIMENSION = 5700;  // GlobalModel.cpp:22
  GlobalModel(int width, int height, size_t capacity = 0) : owned(true) { check(dms_model_create(&h, capacity, width, height), "dms_model_create"); }
  // a view of a frame step's map: `ctx` = the context that owns the per-cluster buffers (GlobalModel.h:93-109)
  explicit GlobalModel(dms_model* borrowed, dms_fusion* ctx = nullptr) : h(borrowed), ctx(ctx), owned(false) {}
  virtual ~GlobalModel() {
    if (owned) dms_model_destroy(h);
  }
  unsigned int lastCount() {
    unsigned int n = 0;
    check(dms_model_count(h, &n, nullptr), "lastCount");
    return n;
  }
  // GlobalModel::downloadMap (GlobalModel.cpp:866-896): reference 15-float records
  std::vector<float> downloadMap() {
    unsigned int n = lastCount(), got = 0;
    std::vector<float> v((size_t)n * 15);
    check(dms_model_download_ref(h, v.data(), n, &got, nullptr), "downloadMap");
    v.resize((size_t)got * 15);
    return v;
  }
  // GlobalModel::consume (GlobalModel.cpp:898-993): append `other` moved by the row-major 4x4 relativeTransform
  void consume(GlobalModel& other, const float* relativeTransform16) { check(dms_model_consume(h, other.h, relativeTransform16, nullptr), "consume"); }
  // GlobalModel::clusters / isCluster (GlobalModel.cpp:251-264)
  std::vector<int> clusters() {
    std::vector<int> ids(1, 0);
    if (ctx) {
      int n = 0;
      check(dms_fusion_clusters(ctx, nullptr, 0, &n, nullptr), "clusters");
      ids.assign((size_t)n, 0);
      check(dms_fusion_clusters(ctx, ids.data(), n, &n, nullptr), "clusters");
    }
    return ids;
  }
  bool isCluster(const int cluster) {
    for (int c : clusters())
      if (c == cluster) return true;
    return false;
  }
  dms_model* h = nullptr;
  dms_fusion* ctx = nullptr;

 private:
  bool owned;
};

// IndexMap (IndexMap.h:39-162): the render targets of one camera and the three "draws" over a GlobalModel.
// pose = device dms_pose_block (dms_pose_block_set); K = (fx, fy, cx, cy).
class IndexMap {
 public:
  IndexMap(int width, int height)
      : index(width, height, 4), vertConf(width, height, 16), colorTime(width, height, 16), normRad(width, height, 16),
        image(width, height, 4), vertex(width, height, 16), normal(width, height, 16), time(width, height, 2), depth(width, height, 4) {
    check(dms_device_alloc((void**)&zbuf, (size_t)width * height * 8), "dms_device_alloc");
    check(dms_device_alloc((void**)&pose, sizeof(dms_pose_block)), "dms_device_alloc");
  }
  ~IndexMap() {
    dms_device_free(zbuf);
    dms_device_free(pose);
  }
  IndexMap(const IndexMap&) = delete;
  void setPose(const float* pose16, dms_stream s = nullptr) { check(dms_pose_block_set(pose, pose16, s), "dms_pose_block_set"); }
  // IndexMap::predictIndices (IndexMap.cpp:146-217)
  void predictIndices(const float* pose16, const int& time, const int& timeIdx, GlobalModel& model, const dms_camera& K, const float depthCutoff,
                      const int timeDelta, dms_stream s = nullptr) {
    setPose(pose16, s);
    dms_indexmap_out o = {index.view, vertConf.view, colorTime.view, normRad.view};
    check(dms_index_map(model.h, pose, &K, time, timeIdx, depthCutoff, timeDelta, zbuf, &o, s), "predictIndices");
  }
  // IndexMap::combinedPredict (IndexMap.cpp:253-368); predictionType: 1 = ACTIVE, 0 = INACTIVE
  void combinedPredict(const float* pose16, GlobalModel& model, const dms_camera& K, const float depthCutoff, const float confThreshold,
                       const int time, const int timeIdx, const int maxTime, const int timeDelta, int predictionType, dms_stream s = nullptr) {
    setPose(pose16, s);
    dms_predict_out o = {image.view, vertex.view, normal.view, this->time.view};
    check(dms_splat_predict(model.h, pose, &K, depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta, predictionType, zbuf, &o, s),
          "combinedPredict");
  }
  // IndexMap::synthesizeDepth (IndexMap.cpp:370-452)
  void synthesizeDepth(const float* pose16, GlobalModel& model, const dms_camera& K, const float depthCutoff, const float confThreshold,
                       const int time, const int timeIdx, const int maxTime, const int timeDelta, dms_stream s = nullptr) {
    setPose(pose16, s);
    check(dms_splat_depth(model.h, pose, &K, depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta, zbuf, &depth.view, s),
          "synthesizeDepth");
  }
  DeviceTexture index, vertConf, colorTime, normRad;  // indexTex / vertConfTex / colorTimeTex / normalRadTex
  DeviceTexture image, vertex, normal, time;          // imageTex / vertexTex / normalTex / timeTex
  DeviceTexture depth;                                // depthTex
  unsigned long long* zbuf = nullptr;
  dms_pose_block* pose = nullptr;
};

// Ferns (Ferns.h:35-266): the per-map keyframe database; textures = full-resolution RGBA8 / RGBA32F DeviceTextures
class Ferns {
 public:
  Ferns(int n, int maxDepth, const float photoThresh, int width, int height, float cx, float cy, float fx, float fy, unsigned seed = 0,
        int capacity = 4096) {
    check(dms_ferns_create(&h, n, maxDepth, photoThresh, width, height, cx, cy, fx, fy, seed, capacity), "dms_ferns_create");
  }
  virtual ~Ferns() { dms_ferns_destroy(h); }
  Ferns(const Ferns&) = delete;
  bool addFrame(DeviceTexture* imageTexture, DeviceTexture* vertexTexture, DeviceTexture* normalTexture, const float* pose16, int srcTime,
                const float threshold, dms_stream s = nullptr) {
    int added = 0;
    check(dms_ferns_add_frame(h, imageTexture->image(), vertexTexture->image(), normalTexture->image(), pose16, srcTime, threshold, &added, s),
          "addFrame");
    return added != 0;
  }
  // returns the estimated pose in match.estPose, lastClosest in match.closest; constraints = rows {raw xyz1 | model xyz1}
  dms_fern_match findFrame(std::vector<float>& constraints, const float* currPose16, DeviceTexture* vertexTexture, DeviceTexture* normalTexture,
                           DeviceTexture* imageTexture, const int time, const bool lost, const bool interMap = false, dms_stream s = nullptr) {
    dms_fern_match m;
    constraints.assign(8 * 64, 0.f);
    check(dms_ferns_find_frame(h, vertexTexture->image(), normalTexture->image(), imageTexture->image(), currPose16, time, lost, interMap, &m,
                               constraints.data(), s),
          "findFrame");
    constraints.resize((size_t)m.n_constraints * 8);
    lastClosest = m.closest;
    return m;
  }
  // Ferns::consume (Ferns.cpp:160-168)
  int consume(Ferns& other, const float* relativeTransform16, const float threshold, dms_stream s = nullptr) {
    int added = 0;
    check(dms_ferns_consume(h, other.h, relativeTransform16, threshold, &added, s), "consume");
    return added;
  }
  int frames() { return dms_ferns_num_frames(h); }
  int lastClosest = -1;
  dms_ferns* h = nullptr;
};

class ElasticFusion {
 public:
  // the constructor arguments of the reference that reach the hot path (ElasticFusion.cpp:22-73)
  ElasticFusion(int width, int height, float fx, float fy, float cx, float cy, const int timeDelta = 200, const float confidence = 10,
                const float depthCut = 3, const float icpThresh = 10, const bool fastOdom = false, const bool so3 = true,
                const bool frameToFrameRGB = false) {
    dms_fusion_params p;
    dms_fusion_default_params(&p, width, height, fx, fy, cx, cy);
    p.timeDelta = timeDelta;
    p.confidence = confidence;
    p.depthCut = depthCut;
    p.icpWeight = icpThresh;
    p.fastOdom = fastOdom;
    p.so3 = so3;
    p.frameToFrameRGB = frameToFrameRGB;
    check(dms_fusion_create(&h, &p), "dms_fusion_create");
  }
  // every option of dms_fusion_params (NID key-framing --nid/--ndw/--npl, hybrid tracking, pipelining ...)
  explicit ElasticFusion(const dms_fusion_params& p) { check(dms_fusion_create(&h, &p), "dms_fusion_create"); }
  virtual ~ElasticFusion() { dms_fusion_destroy(h); }
  ElasticFusion(const ElasticFusion&) = delete;

  // rgb (RGB8) and depth (u16 mm) already in HBM; inPose 4×4 row-major or nullptr
  void processFrame(const unsigned char* rgb_dev, const unsigned short* depth_dev, const float* inPose = nullptr,
                    const float weightMultiplier = 1.f, dms_stream s = nullptr) {
    check(dms_fusion_process_frame(h, rgb_dev, 3, depth_dev, inPose, weightMultiplier, s), "processFrame");
  }
  dms_frame_result fetch(dms_stream s = nullptr) {
    dms_frame_result r;
    check(dms_fusion_fetch(h, &r, s), "fetch");
    return r;
  }
  // local loop closure (ElasticFusion.cpp:399-497): the frame step in two halves around Deformation::constrain
  void processFrameBegin(const unsigned char* rgb_dev, const unsigned short* depth_dev, const float* inPose = nullptr,
                         const float weightMultiplier = 1.f, dms_stream s = nullptr) {
    check(dms_fusion_process_frame_begin(h, rgb_dev, 3, depth_dev, inPose, weightMultiplier, s), "processFrameBegin");
  }
  dms_frame_result fetchLoop(dms_stream s = nullptr) {
    dms_frame_result r;
    check(dms_fusion_fetch_loop(h, &r, s), "fetchLoop");
    return r;
  }
  // rows of {worldRawPoint, worldModelPoint, source time}: the arguments of Deformation::addConstraint (:468-470)
  std::vector<float> loopConstraints(int count) {
    std::vector<float> rows((size_t)count * 7);
    int n = 0;
    check(dms_fusion_get_loop_constraints(h, rows.data(), count, &n), "loopConstraints");
    rows.resize((size_t)(n < count ? n : count) * 7);
    return rows;
  }
  void processFrameEnd(const float* rawGraph = nullptr, int nodes = 0, const float* newPose = nullptr, dms_stream s = nullptr) {
    check(dms_fusion_process_frame_end(h, rawGraph, nodes, newPose, s), "processFrameEnd");
  }
  GlobalModel getGlobalModel() { return GlobalModel(dms_fusion_model(h)); }
  dms_fusion* h = nullptr;
};

}  // namespace dms

#if defined(DMS_WITH_EIGEN) || defined(EIGEN_CORE_H) || defined(EIGEN_CORE_MODULE_H)
#include <Eigen/Core>
namespace dms {
// The reference's own signature (RGBDOdometry.h:55-60)
inline void getIncrementalTransformation(RGBDOdometry& o, Eigen::Vector3f& trans, Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& rot,
                                         const bool& rgbOnly, const float& icpWeight, const bool& pyramid, const bool& fastOdom,
                                         const bool& so3, const bool interMap = false) {
  o.getIncrementalTransformation(trans.data(), rot.data(), rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap);
}
inline void initICPModel(RGBDOdometry& o, DeviceTexture* v, DeviceTexture* n, float depthCutoff, const Eigen::Matrix4f& modelPose) {
  Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p = modelPose;
  o.initICPModel(v, n, depthCutoff, p.data());
}
}  // namespace dms
#endif
