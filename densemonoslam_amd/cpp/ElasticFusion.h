// Source-compatible outer API of the reference's dense back end over the C ABI of libdmslam_hip.so:
//
//   class ElasticFusion  (Core/src/ElasticFusion.h:63-389)  constructor with the reference's 22 parameters (:69-82),
//                        frontend(name) (:315), processFrame with the reference's TEN parameters (:92-100), applyGlobalLoop
//                        (:110), getGlobalModel (:122), the GUI-driven setters (:168-226: setRgbOnly, setIcpWeight, setPyramid,
//                        setFastOdom, setSo3, setFrameToFrameRGB, setConfidenceThreshold, setDepthCutoff) and the end-of-run
//                        exports savePly / saveTrajectories / saveStats / saveTimes (:280-290)
//   class Context        (Core/src/Context.h:25-378)         one camera ("a SLAM frontend"): owns the dms_fusion of that camera;
//                        id(), rgbOnly(), currPose(), tick(), lost(), numFused()
//   Resolution / Intrinsics (Core/src/Utils/Resolution.h, Intrinsics.h)  the singletons the front end fills before it
//                        constructs ElasticFusion (GUI/src/MainController.cpp:39-60)
//
// so that GUI/src/MainController.cpp:203-229 (construction, frontend(), ctx.rgbOnly()) and :373-377 (the processFrame call with
// the ORB-SLAM3 pose prior and loop-closure poses) compile against this header unchanged.  The 4x4 matrix type is a template
// parameter (anything with float operator()(row, col): Eigen::Matrix4f in the reference's build); `ElasticFusion` itself is
// the Eigen instantiation when <Eigen/Core> has been included before this header, so the header also compiles without Eigen
// (tests/test_cpp_mirror.py uses a 20-line stand-in).
//
// What stays with the caller, as in SURVEY.md §8 / DESIGN.md §8: the deformation-graph optimisation
// (Deformation::addConstraint / constrain, CPU + CHOLMOD).  It plugs in through `constrain`: called with the constraint rows the
// device produced ({source xyz, target xyz, time} = the arguments of Deformation::addConstraint), it returns the node table
// `rawGraph` (16 floats per node, Deformation.cpp:192-201) — empty = no deformation — exactly where the reference calls
// rf.globalDeformation().constrain / rf.localDeformation().constrain (ElasticFusion.cpp:337, :481).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dmslam.hpp"

// ---- the reference's sensor singletons (same accessors) -------------------------------------------------------------
class Resolution {
 public:
  static const Resolution& getInstance(int width = 0, int height = 0) {
    static const Resolution instance(width, height);
    return instance;
  }
  const int& width() const { return imgWidth; }
  const int& height() const { return imgHeight; }
  const int& cols() const { return imgWidth; }
  const int& rows() const { return imgHeight; }
  const int& numPixels() const { return imgNumPixels; }

 private:
  Resolution(int width, int height) : imgWidth(width), imgHeight(height), imgNumPixels(width * height) {
    if (width <= 0 || height <= 0) throw std::runtime_error("Resolution::getInstance(width, height) must be called first");
  }
  const int imgWidth, imgHeight, imgNumPixels;
};

class Intrinsics {
 public:
  static const Intrinsics& getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0) {
    static const Intrinsics instance(fx, fy, cx, cy);
    return instance;
  }
  const float& fx() const { return fx_; }
  const float& fy() const { return fy_; }
  const float& cx() const { return cx_; }
  const float& cy() const { return cy_; }

 private:
  Intrinsics(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy) {
    if (fx == 0 || fy == 0) throw std::runtime_error("Intrinsics::getInstance(fx, fy, cx, cy) must be called first");
  }
  const float fx_, fy_, cx_, cy_;
};

namespace dms {

// the command-line options of the reference that reach this path (Core/src/Utils/Options.h: --hybrid_tracking, --hybrid_loops,
// --i <icp weight per camera>, --d, GUI "Pyramid"); the front end sets them before it constructs ElasticFusion
struct FrontEndOptions {
  bool hybrid_tracking = false;  // refine the pose prior with the dense tracker (ElasticFusion.cpp:169)
  bool hybrid_loops = false;     // accept ORB-SLAM3 loop closures (:292)
  bool pyramid = true;
  float maxDepthProcessed = 25.f;  // ElasticFusion.cpp:56
  size_t model_capacity = 0;       // 0 = the reference's MAX_VERTICES
  float depth = 3.f;                   // Options::get().depth (--d): ReferenceFrame's fern database is built with depth * 1000 (ReferenceFrame.h:17)
  float interMapPhotoThresh = 115.f;   // Options::get().interMapPhotoThresh
  float fernThresh = 0.3095f;          // Options::get().fernThresh (ReferenceFrame.h:125)
  float covThresh = 1e-05f;            // Options::get().covThresh / icpErrThresh / icpCountThresh (Options.h:91-94): the acceptance test of
  float icpErrThresh = 2e-05f;         // ReferenceFrame::resolveRelativeTransformationFern (ReferenceFrame.h:98-110)
  int icpCountThresh = 35000;
  static FrontEndOptions& get() {
    static FrontEndOptions o;
    return o;
  }
};

// One camera: Context (Context.h:25-378).  The device side of everything the reference keeps per Context (IndexMap, two
// RGBDOdometry, FillIn, the input textures) lives in the dms_fusion it owns.
template <class Mat4>
class ContextT {
 public:
  ContextT(const int id, const int num_bins_depth, const int num_bins_img, const std::string filename = "", const bool iclnuim = false,
           const bool reloc = false)
      : m_id(id), m_bins_depth(num_bins_depth), m_bins_img(num_bins_img), m_file(filename), m_reloc(reloc) {
    (void)iclnuim;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) m_currPose(r, c) = r == c ? 1.f : 0.f;
  }
  virtual ~ContextT() {
    if (fusion) dms_fusion_destroy(fusion);
    if (rgb_dev) dms_device_free(rgb_dev);
    if (depth_dev) dms_device_free(depth_dev);
  }
  ContextT(const ContextT&) = delete;
  const int& id() const { return m_id; }
  bool& rgbOnly() { return m_rgbOnly; }
  Mat4& currPose() { return m_currPose; }
  int& tick() { return m_tick; }
  // Context::computeFeedbackBuffers (Context.h:211-223): the next new cluster starts from the last processed frame's surfels
  void computeFeedbackBuffers(const int& /*maxDepthProcessed*/) {
    if (fusion && dms_fusion_compute_feedback(fusion, nullptr) != DMS_OK) throw std::runtime_error(dms_last_error());
  }
  bool& lost() { return m_lost; }
  // Context::fillIn() (Context.h; FillIn.h:33-35: vertexTexture / normalTexture / imageTexture): views of the context's fill-in
  // images in HBM, what the inter-map block hands to resolveRelativeTransformationFern (ElasticFusion.cpp:601-603)
  struct FillInTextures {
    DeviceTexture imageTexture, vertexTexture, normalTexture;
    FillInTextures(const dms_image2d& i, const dms_image2d& v, const dms_image2d& n) : imageTexture(i, 4), vertexTexture(v, 16), normalTexture(n, 16) {}
  };
  FillInTextures& fillIn() {
    if (!fusion) throw std::runtime_error("Context::fillIn: the camera has not processed a frame yet");
    if (!m_fillIn) {
      dms_image2d i, v, n;
      if (dms_fusion_get_image(fusion, 13, &i) || dms_fusion_get_image(fusion, 14, &v) || dms_fusion_get_image(fusion, 15, &n))
        throw std::runtime_error(dms_last_error());
      m_fillIn.reset(new FillInTextures(i, v, n));
    }
    return *m_fillIn;
  }
  int& numFused() { return m_numFused; }
  const std::string& filename() const { return m_file; }
  const dms_frame_result& lastResult() const { return last; }
  float lastKFScore() const { return last.nid_score; }  // Context::lastKFScore (Context.h: nidScores().back())
  // one (tick, pose) per processed frame and its time stamp (ElasticFusion.cpp:571-574)
  std::vector<std::pair<unsigned long long int, Mat4>>& poseGraph() { return m_poseGraph; }
  std::vector<int64_t>& poseLogTimes() { return m_poseLogTimes; }

  // Context::saveTrajectory (Context.h:117-156): <dir><log name from its last '/'>.freiburg, one 3 x 4 matrix per line
  void saveTrajectory(std::string dir) {
    const size_t slash = m_file.find_last_of("/");
    std::string fname = dir + (slash == std::string::npos ? m_file : m_file.substr(slash)) + ".freiburg";
    std::vector<float> flat(m_poseGraph.size() * 16);
    for (size_t i = 0; i < m_poseGraph.size(); ++i)
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) flat[i * 16 + r * 4 + c] = m_poseGraph[i].second(r, c);
    check(dms_trajectory_save(fname.c_str(), flat.data(), m_poseGraph.size()), "saveTrajectory");
  }
  // Context::saveStats (Context.h:106-115) writes the front end's Stats object (ORB / depth-net bookkeeping, not on this
  // path); here: the counters this camera has, one "name value" line each, in <dir><log name>.stats
  void saveStats(std::string dir) {
    const size_t slash = m_file.find_last_of("/");
    std::string fname = dir + (slash == std::string::npos ? m_file : m_file.substr(slash + 1)) + ".stats";
    FILE* fp = fopen(fname.c_str(), "w");
    if (!fp) throw std::runtime_error("saveStats: cannot open " + fname);
    fprintf(fp, "frames %zu\nfused %d\ntick %d\nlost %d\nsurfels %u\n", m_poseGraph.size(), m_numFused, m_tick, m_lost ? 1 : 0, last.surfels);
    fclose(fp);
  }

  dms_fusion* fusion = nullptr;  // created by the first processFrame (it needs ElasticFusion's parameters)
  void* rgb_dev = nullptr;
  void* depth_dev = nullptr;
  dms_frame_result last;
  int bins_depth() const { return m_bins_depth; }
  int bins_img() const { return m_bins_img; }
  bool reloc() const { return m_reloc; }

 private:
  const int m_id, m_bins_depth, m_bins_img;
  const std::string m_file;
  const bool m_reloc;
  bool m_rgbOnly = false, m_lost = false;
  int m_tick = 1, m_numFused = 0;
  Mat4 m_currPose;
  std::vector<std::pair<unsigned long long int, Mat4>> m_poseGraph;
  std::vector<int64_t> m_poseLogTimes;
  std::unique_ptr<FillInTextures> m_fillIn;
};

// The deformation graphs stay with the caller (their optimisation is CPU + CHOLMOD, SURVEY 8 "out of scope"); what the reference's
// GUI reads of them - the node table of the last constrain() - is kept here (MainController.cpp:583, :741)
struct Deformation {
  std::vector<float> graph;  // rawGraph: 16 floats per node (Deformation.cpp:192-201)
  std::vector<float>& getGraph() { return graph; }
};

// One map and the cameras that fuse into it: ReferenceFrame (Core/src/ReferenceFrame.h:13-214).  "When a new context is created it
// is assumed to be in its own reference frame.  As inter-map global loop closures occur reference frames are aligned, with one
// essentially consuming the other" (ElasticFusion.h:318-323).
template <class Mat4>
class ReferenceFrameT {
 public:
  typedef ContextT<Mat4> Context;
  std::string name;
  std::map<std::string, std::shared_ptr<Context>>& contexts() { return m_contexts; }
  bool& firstRun() { return m_firstRun; }
  Deformation& globalDeformation() { return m_globalDeformation; }
  Deformation& localDeformation() { return m_localDeformation; }
  // the map: the device object of the frame's founding camera (every camera merged into the frame shares it)
  GlobalModel globalModel() {
    Context* c = founder();
    if (!c || !c->fusion) throw std::runtime_error("ReferenceFrame::globalModel: no camera of this frame has processed a frame yet");
    return GlobalModel(dms_fusion_model(c->fusion), c->fusion);
  }
  // the frame's fern key-frame database: Ferns(500, Options::get().depth * 1000, Options::get().interMapPhotoThresh) (ReferenceFrame.h:17)
  Ferns& ferns() {
    if (!m_ferns) {
      const Resolution& r = Resolution::getInstance();
      const Intrinsics& k = Intrinsics::getInstance();
      const FrontEndOptions& o = FrontEndOptions::get();
      m_ferns.reset(new Ferns(500, (int)(o.depth * 1000), o.interMapPhotoThresh, r.width(), r.height(), k.cx(), k.cy(), k.fx(), k.fy()));
    }
    return *m_ferns;
  }
  // ReferenceFrame::resolveRelativeTransformationFern (ReferenceFrame.h:34-110), argument for argument (GPUTexture -> dms::DeviceTexture:
  // the querying camera's fill-in vertex / normal / colour textures, dense RGBA32F / RGBA32F / RGBA8 on this device): Ferns::findFrame
  // with interMap = true on this frame's database, then the full-resolution refinement against an INACTIVE prediction of this frame's
  // map (dms_refframe_refine = m_index + m_rgbd) and the acceptance test on covariance / lastICPError / lastICPCount.
  // `constraints` receives rows {worldRawPoint xyz1 | worldModelPoint xyz1} (Ferns::SurfaceConstraint).
  bool resolveRelativeTransformationFern(std::vector<float>& constraints, Mat4& relativeTransform, Mat4& currPose, DeviceTexture& vertexTexture,
                                         DeviceTexture& normalTexture, DeviceTexture& imageTexture, const int& tick, const bool lost,
                                         const int depthCutoff, const float& confidenceThreshold, const int& timeIdx, const int& timeDelta,
                                         const int& maxTime) {
    Context* owner = founder();
    if (!owner || !owner->fusion) throw std::runtime_error("resolveRelativeTransformationFern: this frame has no map yet");
    float cur[16];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) cur[r * 4 + c] = currPose(r, c);
    const dms_fern_match m = ferns().findFrame(constraints, cur, &vertexTexture, &normalTexture, &imageTexture, tick, lost, true);
    if (ferns().lastClosest == -1) return false;  // :66
    if (!m_refine) {
      const Resolution& rs = Resolution::getInstance();
      const Intrinsics& k = Intrinsics::getInstance();
      check(dms_refframe_create(&m_refine, rs.width(), rs.height(), k.cx(), k.cy(), k.fx(), k.fy()), "dms_refframe_create");
    }
    const FrontEndOptions& o = FrontEndOptions::get();
    dms_intermap_result res;
    check(dms_refframe_refine(m_refine, dms_fusion_model(owner->fusion), m.estPose, cur, (const float*)vertexTexture.ptr,
                              (const float*)normalTexture.ptr, imageTexture.ptr, depthCutoff, confidenceThreshold, timeIdx, timeDelta, maxTime,
                              o.covThresh, o.icpErrThresh, (float)o.icpCountThresh, &res, nullptr),
          "resolveRelativeTransformationFern");
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) relativeTransform(r, c) = res.relativeTransform[r * 4 + c];
    lastInterMap = res;
    return res.accepted != 0;
  }
  dms_intermap_result lastInterMap{};  // side outputs of the last refinement (covariance diagonal, error, count, iterations)
  ~ReferenceFrameT() {
    if (m_refine) dms_refframe_destroy(m_refine);
  }
  ReferenceFrameT() = default;
  ReferenceFrameT(const ReferenceFrameT&) = delete;
  // ReferenceFrame::consumeReferenceFrame (ReferenceFrame.h:121-150): this frame's map consumes `other`'s moved by relativeTransform,
  // the key-frame databases merge, and other's cameras move over - currPose and pose graph re-based - to fuse into this map from now on
  void consumeReferenceFrame(ReferenceFrameT& other, Mat4 relativeTransform) {
    Context* owner = founder();
    if (!owner || !owner->fusion) throw std::runtime_error("consumeReferenceFrame: the consuming frame has no map yet");
    float T[16];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T[r * 4 + c] = relativeTransform(r, c);
    if (other.m_ferns) ferns().consume(*other.m_ferns, T, FrontEndOptions::get().fernThresh);
    Context* of = other.founder();
    std::vector<Context*> order;  // the camera that owns other's map carries it over, the rest only move
    if (of) order.push_back(of);
    for (auto& kv : other.contexts())
      if (kv.second.get() != of) order.push_back(kv.second.get());
    for (Context* c : order) {
      if (c->fusion) {
        check(dms_fusion_join_map(c->fusion, owner->fusion, T, nullptr), "consumeReferenceFrame");
        float p[16], q[16];
        for (int r = 0; r < 4; ++r)
          for (int k = 0; k < 4; ++k) p[r * 4 + k] = c->currPose()(r, k);
        dms_pose_compose(T, p, q);
        for (int r = 0; r < 4; ++r)
          for (int k = 0; k < 4; ++k) c->currPose()(r, k) = q[r * 4 + k];
        for (auto& tp : c->poseGraph()) {
          for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) p[r * 4 + k] = tp.second(r, k);
          dms_pose_compose(T, p, q);
          for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) tp.second(r, k) = q[r * 4 + k];
        }
      }
    }
    for (auto& kv : other.contexts()) m_contexts[kv.first] = kv.second;
  }
  // the camera whose device context owns the map's storage: the one the frame was created for
  Context* founder() {
    if (m_contexts.empty()) return nullptr;
    auto it = m_contexts.find(name);  // (frontend() names the frame after the camera it creates it for)
    return it != m_contexts.end() ? it->second.get() : m_contexts.begin()->second.get();
  }

 private:
  std::map<std::string, std::shared_ptr<Context>> m_contexts;
  Deformation m_globalDeformation, m_localDeformation;
  std::unique_ptr<Ferns> m_ferns;
  dms_refframe* m_refine = nullptr;  // m_index + m_rgbd (ReferenceFrame.h:203-214)
  bool m_firstRun = true;
};

template <class Mat4>
class ElasticFusionT {
 public:
  typedef ContextT<Mat4> Context;
  typedef ReferenceFrameT<Mat4> ReferenceFrame;
  enum SamplingScheme { NID_KEYFRAMING, NONE, UNIFORM };

  // the reference's constructor (ElasticFusion.h:69-82)
  ElasticFusionT(const int timeDelta = 200, const int countThresh = 35000, const float errThresh = 5e-05, const float covThresh = 1e-05,
                 const bool closeLoops = true, const bool iclnuim = false, const bool reloc = false, const float photoThresh = 115,
                 const float confidence = 10, const float depthCut = 3, const float icpThresh = 10, const bool fastOdom = false,
                 const float fernThresh = 0.3095, const bool so3 = true, const bool frameToFrameRGB = false, const std::string fileName = "",
                 const SamplingScheme sampling_scheme = NID_KEYFRAMING, const float nid_threshold = 0.80f, const float nidDepthLambda = 0.7f,
                 const int num_bins_depth = 500, const int num_bins_img = 64, const int m_nid_pyramid_level = 0)
      : timeDelta(timeDelta), closeLoops(closeLoops), iclnuim(iclnuim), reloc(reloc), confidence(confidence), depthCut(depthCut),
        icpThresh(icpThresh), fastOdom(fastOdom), so3(so3), frameToFrameRGB(frameToFrameRGB), scheme(sampling_scheme),
        nid_threshold(nid_threshold), nidDepthLambda_(nidDepthLambda), bins_depth(num_bins_depth), bins_img(num_bins_img),
        nid_level(m_nid_pyramid_level), bins_img_now(num_bins_img), bins_depth_now(num_bins_depth), saveFilename(fileName) {
    (void)countThresh;  // icpCountThresh / icpErrThresh / covThresh are the thresholds of the local-loop acceptance test, fixed at the
    (void)errThresh;    // values ElasticFusion.cpp:428-442 hard-codes; photoThresh / fernThresh belong to the fern database
    (void)covThresh;    // (dms::Ferns), whose call sites this fork compiles out (:279, :589, :597)
    (void)photoThresh;
    (void)fernThresh;
  }
  virtual ~ElasticFusionT() {}

  // Deformation::constrain of the caller: (constraint rows: n x 7 floats, tick, isGlobal) -> rawGraph (16 floats per node)
  std::function<std::vector<float>(const std::vector<float>&, int, bool)> constrain;

  // "a context represents a SLAM frontend" (ElasticFusion.h:309-315)
  std::shared_ptr<Context> frontend(std::string name) {  // ElasticFusion.cpp:1069-1085
    for (const auto& rf : m_referenceFrames) {
      auto ctx = rf->contexts().find(name);
      if (ctx != rf->contexts().end()) return ctx->second;
    }
    std::shared_ptr<ReferenceFrame> rf(new ReferenceFrame());
    rf->name = name;
    auto c = std::make_shared<Context>(nextId++, bins_depth, bins_img, name, iclnuim, reloc);
    m_referenceFrames.push_back(rf);
    m_contextToReferenceFrameMap[c->id()] = rf;
    rf->contexts()[name] = c;
    m_contexts[name] = c;
    return c;
  }
  std::map<std::string, std::shared_ptr<Context>>& contextMap() { return m_contexts; }
  // ElasticFusion::contexts / referenceFrames / whichReferenceFrame (ElasticFusion.h:324-332, ElasticFusion.cpp:1087-1105)
  std::vector<std::shared_ptr<Context>> contexts() {
    std::vector<std::shared_ptr<Context>> ctxs;
    for (auto& rf : m_referenceFrames)
      for (auto& kv : rf->contexts()) ctxs.push_back(kv.second);
    return ctxs;
  }
  std::vector<std::shared_ptr<ReferenceFrame>>& referenceFrames() { return m_referenceFrames; }
  ReferenceFrame& whichReferenceFrame(Context& ctx) { return *(m_contextToReferenceFrameMap[ctx.id()]); }
  // what the (compiled-out) inter-map block does on a verified match (ElasticFusion.cpp:610-627): `consuming` takes over the
  // frame of `context`, which disappears from referenceFrames(); every camera of it is re-mapped
  void mergeReferenceFrames(ReferenceFrame& consuming, Context& context, Mat4 relativeTransform) {
    std::shared_ptr<ReferenceFrame> gone = m_contextToReferenceFrameMap[context.id()];
    std::shared_ptr<ReferenceFrame> keep;
    for (auto& rf : m_referenceFrames)
      if (rf.get() == &consuming) keep = rf;
    if (!keep || keep == gone) throw std::runtime_error("mergeReferenceFrames: not two different reference frames of this ElasticFusion");
    consuming.consumeReferenceFrame(*gone, relativeTransform);
    for (size_t i = 0; i < m_referenceFrames.size(); i++)
      if (m_referenceFrames[i] == gone) {
        m_referenceFrames.erase(m_referenceFrames.begin() + i);
        break;
      }
    for (auto& kv : consuming.contexts()) m_contextToReferenceFrameMap[kv.second->id()] = keep;
  }
  Ferns& getFerns(Context& ctx) { return whichReferenceFrame(ctx).ferns(); }                                // ElasticFusion.cpp:997
  Deformation& getLocalDeformation(Context& ctx) { return whichReferenceFrame(ctx).localDeformation(); }  // :1001
  // the session clock the GUI loop reads and fast-forwards (ElasticFusion.cpp:1050-1054): a plain member there too - the per-camera
  // clock is Context::tick()
  const int& getTick() { return tick; }
  void setTick(const int& val) { tick = val; }
  // ElasticFusion::predict(context, rf[, confidence]) (ElasticFusion.h:107-108): the model view at the camera's pose + fill-in
  void predict(Context& context, ReferenceFrame& rf) { predict(context, rf, -1.f); }
  void predict(Context& context, ReferenceFrame& rf, float confidenceThreshold) {
    (void)rf;  // (the camera's device context already renders the map of the frame it belongs to)
    if (context.fusion) check(dms_fusion_predict(context.fusion, confidenceThreshold, nullptr), "predict");
  }
  // NID key-framing accessors of the GUI (ElasticFusion.h:336-372, MainController.cpp:470, :777-781)
  float& nidThreshold() { return nid_threshold; }
  float& nidDepthLambda() { return nidDepthLambda_; }
  int& nidPyramidLevel() { return nid_level; }
  void setNumBinsImg(int numBinsImg) { bins_img_now = numBinsImg; }
  void setNumBinsDepth(int numBinsDepth) { bins_depth_now = numBinsDepth; }
  float lastKFScore(Context& context) { return context.lastKFScore(); }
  float kFThreshold() { return scheme == NID_KEYFRAMING ? nid_threshold : 0.f; }
  int& numFused(Context& context) { return context.numFused(); }
  const int& getDeforms() { return deforms; }
  const int& getFernDeforms() { return fernDeforms; }
  int surfelCount() {  // ElasticFusion.cpp:1107-1116
    int count = 0;
    for (auto& rf : m_referenceFrames) {
      Context* c = rf->founder();
      if (c && c->fusion) count += (int)rf->globalModel().lastCount();
    }
    return count;
  }

  /**
   * Process an rgb/depth map pair — ElasticFusion::processFrame (ElasticFusion.h:92-100, ElasticFusion.cpp:99-637)
   * @param rgb unsigned char row major order (RGB8, host)
   * @param depth unsigned short z-depth in millimeters, invalid depths are 0 (host)
   * @param timestamp only used for the output poses
   * @param inPose pose prior (the ORB-SLAM3 pose); with hybrid_tracking it is refined by the dense tracker, else taken as is
   * @param orbTcwOld, orbTcwNew the loop-closure candidate of the ORB-SLAM3 front end (hybrid_loops)
   * @param cluster id of the ground-truth-clusters mode: a frame that fuses under an id the map does not know starts surfel buffers
   *        of its own from the context's feedback buffers, and they stay current (GlobalModel.cpp:266-277, dmslam_fusion.h)
   * @param weightMultiplier optional full frame fusion weight
   * @param bootstrap if true, use inPose as a pose guess rather than replacement (re-assigns currPose = *inPose, :188-191)
   */
  void processFrame(const std::shared_ptr<unsigned char>& rgb, const std::shared_ptr<unsigned short>& depth, const int64_t& timestamp,
                    Context& context, const Mat4* inPose = 0, const Mat4* orbTcwOld = 0, const Mat4* orbTcwNew = 0, const int cluster = 0,
                    const float weightMultiplier = 1.f, const bool bootstrap = false) {
    (void)bootstrap;
    const int tick_before = context.tick();
    ensure(context);
    check(dms_fusion_set_cluster(context.fusion, cluster), "dms_fusion_set_cluster");
    // Context::rgbOnly() is read every frame by the reference (ElasticFusion.cpp:505): keep the device side in step
    check(dms_fusion_set_option(context.fusion, DMS_OPT_RGB_ONLY, (context.rgbOnly() || rgbOnly) ? 1.0 : 0.0), "dms_fusion_set_option");
    if (scheme == NID_KEYFRAMING) {  // fuseFrame reads these members every frame (ElasticFusion.cpp:646-675); the GUI may have moved them
      check(dms_fusion_set_option(context.fusion, DMS_OPT_NID_THRESHOLD, nid_threshold), "nid_threshold");
      check(dms_fusion_set_option(context.fusion, DMS_OPT_NID_DEPTH_LAMBDA, nidDepthLambda_), "nid_depth_lambda");
      check(dms_fusion_set_option(context.fusion, DMS_OPT_NID_BINS_IMG, bins_img_now), "setNumBinsImg");
      check(dms_fusion_set_option(context.fusion, DMS_OPT_NID_BINS_DEPTH, bins_depth_now), "setNumBinsDepth");
      check(dms_fusion_set_option(context.fusion, DMS_OPT_NID_PYRAMID_LEVEL, nid_level), "nidPyramidLevel");
    }
    const int W = Resolution::getInstance().width(), H = Resolution::getInstance().height();
    check(dms_memcpy_h2d(context.rgb_dev, rgb.get(), (size_t)W * H * 3, nullptr), "upload rgb");
    check(dms_memcpy_h2d(context.depth_dev, depth.get(), (size_t)W * H * 2, nullptr), "upload depth");
    float prior[16], po[16], pn[16];
    if (inPose) flat(*inPose, prior);
    const bool orb = FrontEndOptions::get().hybrid_loops && orbTcwOld && orbTcwNew;
    if (orb) {
      flat(*orbTcwOld, po);
      flat(*orbTcwNew, pn);
      check(dms_fusion_set_orb_loop(context.fusion, po, pn), "set_orb_loop");
    }
    if (orb || closeLoops) {
      // the frame step in two halves around the caller's deformation solver (ElasticFusion.cpp:337 / :481)
      check(dms_fusion_process_frame_begin(context.fusion, context.rgb_dev, 3, (const unsigned short*)context.depth_dev, inPose ? prior : nullptr,
                                           weightMultiplier, nullptr),
            "processFrame");
      std::vector<float> rawGraph;
      float newPose[16];
      bool havePose = false;
      const int cap = (W / 20) * (H / 20);
      if (orb) {
        std::vector<float> rows((size_t)cap * 7);
        int n = 0;
        check(dms_fusion_get_global_loop_constraints(context.fusion, rows.data(), cap, &n, nullptr), "global loop constraints");
        rows.resize((size_t)n * 7);
        if (constrain && n > 0) rawGraph = constrain(rows, context.tick(), true);
        if (!rawGraph.empty()) {
          fernDeforms += 1;  // :346
          whichReferenceFrame(context).globalDeformation().graph = rawGraph;
        }
      }
      if (rawGraph.empty() && closeLoops) {  // `rawGraph.size() == 0` (:399)
        dms_frame_result rl;
        check(dms_fusion_fetch_loop(context.fusion, &rl, nullptr), "fetch_loop");
        if (rl.loop_ok && constrain) {
          std::vector<float> rows((size_t)cap * 7);
          int n = 0;
          check(dms_fusion_get_loop_constraints(context.fusion, rows.data(), cap, &n), "loop constraints");
          rows.resize((size_t)n * 7);
          rawGraph = constrain(rows, context.tick(), false);
          if (!rawGraph.empty()) {  // context.currPose() = estPose (:489)
            deforms += 1;  // :487
            whichReferenceFrame(context).localDeformation().graph = rawGraph;
            std::memcpy(newPose, rl.loop_pose, sizeof(newPose));
            havePose = true;
          }
        }
      }
      check(dms_fusion_process_frame_end(context.fusion, rawGraph.empty() ? nullptr : rawGraph.data(), (int)(rawGraph.size() / 16),
                                         havePose ? newPose : nullptr, nullptr),
            "processFrame");
    } else {
      check(dms_fusion_process_frame(context.fusion, context.rgb_dev, 3, (const unsigned short*)context.depth_dev, inPose ? prior : nullptr,
                                     weightMultiplier, nullptr),
            "processFrame");
    }
    refresh(context);
    // :571-574 (the tick the frame was processed at; the pose after tracking / loop closure)
    context.poseGraph().push_back(std::pair<unsigned long long int, Mat4>((unsigned long long int)tick_before, context.currPose()));
    context.poseLogTimes().push_back(timestamp);
  }

  // ElasticFusion::applyGlobalLoop (ElasticFusion.h:110, ElasticFusion.cpp:1148-1240)
  void applyGlobalLoop(Context& context, Mat4& orbTcwOld, Mat4& orbTcwNew) {
    ensure(context);
    float po[16], pn[16];
    flat(orbTcwOld, po);
    flat(orbTcwNew, pn);
    check(dms_fusion_apply_global_loop_begin(context.fusion, po, pn, nullptr), "applyGlobalLoop");
    const int cap = (Resolution::getInstance().width() / 20) * (Resolution::getInstance().height() / 20);
    std::vector<float> rows((size_t)cap * 7), rawGraph;
    int n = 0;
    check(dms_fusion_get_global_loop_constraints(context.fusion, rows.data(), cap, &n, nullptr), "global loop constraints");
    rows.resize((size_t)n * 7);
    if (constrain && n > 0) rawGraph = constrain(rows, context.tick(), true);
    check(dms_fusion_apply_global_loop_end(context.fusion, rawGraph.empty() ? nullptr : rawGraph.data(), (int)(rawGraph.size() / 16),
                                           rawGraph.empty() ? 0 : 1, nullptr),
          "applyGlobalLoop");
  }

  GlobalModel getGlobalModel(Context& ctx) {
    ensure(ctx);
    return GlobalModel(dms_fusion_model(ctx.fusion));
  }
  const int& getTimeDelta() const { return timeDelta; }
  const float& getConfidenceThreshold() const { return confidence; }
  const float& getMaxDepthProcessed() const { return FrontEndOptions::get().maxDepthProcessed; }

  // the GUI-driven setters (ElasticFusion.h:168-226, ElasticFusion.cpp:1023-1043; MainController.cpp:760-775 calls them every
  // frame): they take effect with the next processFrame of every camera
  void setRgbOnly(const bool& val) { rgbOnly = val; push(DMS_OPT_RGB_ONLY, val); }
  void setIcpWeight(const float& val) { icpThresh = val; push(DMS_OPT_ICP_WEIGHT, val); }
  void setPyramid(const bool& val) { FrontEndOptions::get().pyramid = val; push(DMS_OPT_PYRAMID, val); }
  void setFastOdom(const bool& val) { fastOdom = val; push(DMS_OPT_FAST_ODOM, val); }
  void setSo3(const bool& val) { so3 = val; push(DMS_OPT_SO3, val); }
  void setFrameToFrameRGB(const bool& val) { frameToFrameRGB = val; push(DMS_OPT_FRAME_TO_FRAME_RGB, val); }
  void setConfidenceThreshold(const float& val) { confidence = val; push(DMS_OPT_CONFIDENCE, val); }
  void setDepthCutoff(const float& val) { depthCut = val; push(DMS_OPT_DEPTH_CUTOFF, val); }
  void setFernThresh(const float&) {}  // the fern relocaliser's call sites are compiled out in this fork (:279, :589)

  // End-of-run exports (MainController.cpp:806-809).
  // savePly (ElasticFusion.cpp:781-885): one file per map, <dir><fileName>.<j>.<log name>.ply, j counting from 1; the
  // reference's bytes (dms_model_save_ply; `reference_normal_offset` = its stale + 18 read, see dmslam_fusion.h)
  void savePly(std::string dir, bool reference_normal_offset = false) {
    int j = 1;
    for (auto& kv : m_contexts) {
      Context& c = *kv.second;
      if (!c.fusion) continue;
      const std::string& name = c.filename();
      const size_t slash = name.find_last_of("/");
      const std::string refName = slash == std::string::npos ? name : name.substr(slash + 1);
      const std::string filename = dir + saveFilename + "." + std::to_string(j++) + "." + refName + ".ply";
      check(dms_model_save_ply(dms_fusion_model(c.fusion), filename.c_str(), confidence, reference_normal_offset ? 1 : 0, nullptr), "savePly");
    }
  }
  void saveTrajectories(std::string dir) {  // :968-974
    for (auto& kv : m_contexts) kv.second->saveTrajectory(dir);
  }
  void saveStats(std::string dir) {  // :887-891
    for (auto& kv : m_contexts) kv.second->saveStats(dir);
  }
  // saveTimes (:893-966) dumps the reference's Stopwatch singleton (host wall-clock sections of the GUI loop, the ORB front
  // end and the depth network: none of them on this path).  Here: the device-side stage times of every camera when
  // profiling was switched on (dms_fusion_set_profiling), "camera<id><stage> total_ms launches" per line, in <dir><fileName>.timings
  void saveTimes(std::string dir) {
    const std::string fname = dir + saveFilename + ".timings";
    FILE* fp = fopen(fname.c_str(), "w");
    if (!fp) throw std::runtime_error("saveTimes: cannot open " + fname);
    static const char* stages[] = {"ingest", "preprocess", "live_pyramids", "predict", "odom_init", "track", "index_map", "fuse", "clean", "host_wait"};
    for (auto& kv : m_contexts) {
      Context& c = *kv.second;
      if (!c.fusion) continue;
      for (const char* st : stages) {
        double ms = 0.0;
        int n = 0;
        if (dms_fusion_get_kernel_time(c.fusion, st, &ms, &n) == 0) fprintf(fp, "camera%d%s %.6f %d\n", c.id(), st, ms, n);
      }
    }
    fclose(fp);
  }

 private:
  static void flat(const Mat4& m, float* o) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) o[r * 4 + c] = m(r, c);
  }
  void push(int option, double value) {
    for (auto& kv : m_contexts)
      if (kv.second->fusion) check(dms_fusion_set_option(kv.second->fusion, option, value), "dms_fusion_set_option");
  }
  void ensure(Context& c) {
    if (c.fusion) return;
    const Resolution& res = Resolution::getInstance();
    const Intrinsics& k = Intrinsics::getInstance();
    const FrontEndOptions& o = FrontEndOptions::get();
    dms_fusion_params p;
    dms_fusion_default_params(&p, res.width(), res.height(), k.fx(), k.fy(), k.cx(), k.cy());
    p.timeDelta = timeDelta;
    p.confidence = confidence;
    p.depthCut = depthCut;
    p.icpWeight = icpThresh;
    p.fastOdom = fastOdom;
    p.so3 = so3;
    p.frameToFrameRGB = frameToFrameRGB;
    p.pyramid = o.pyramid;
    p.hybrid_tracking = o.hybrid_tracking;
    p.hybrid_loops = o.hybrid_loops;
    p.rgbOnly = c.rgbOnly() || rgbOnly;
    p.timeIdx = c.id();
    p.maxDepthProcessed = o.maxDepthProcessed;
    p.model_capacity = o.model_capacity;
    p.local_loop_closure = closeLoops;
    p.reloc = c.reloc();
    p.nid_keyframing = scheme == NID_KEYFRAMING;
    p.nid_threshold = nid_threshold;
    p.nid_depth_lambda = nidDepthLambda_;
    p.nid_bins_depth = bins_depth;
    p.nid_bins_img = bins_img;
    p.nid_pyramid_level = nid_level;
    check(dms_fusion_create(&c.fusion, &p), "dms_fusion_create");
    check(dms_device_alloc(&c.rgb_dev, (size_t)res.numPixels() * 3), "dms_device_alloc");
    check(dms_device_alloc(&c.depth_dev, (size_t)res.numPixels() * 2), "dms_device_alloc");
  }
  void refresh(Context& c) {
    check(dms_fusion_fetch(c.fusion, &c.last, nullptr), "fetch");
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) c.currPose()(r, q) = c.last.pose[r * 4 + q];
    c.tick() = c.last.tick;
    c.lost() = c.last.lost != 0;
    c.numFused() += c.last.fused;
  }

  const int timeDelta;
  const bool closeLoops, iclnuim, reloc;
  float confidence, depthCut, icpThresh;  // (changed by the setters below)
  bool fastOdom, so3, frameToFrameRGB;
  bool rgbOnly = false;  // ElasticFusion::setRgbOnly (:1023): every camera tracks photometrically only and fuses nothing
  const SamplingScheme scheme;
  float nid_threshold, nidDepthLambda_;  // (the GUI writes them through nidThreshold() / nidDepthLambda() every frame)
  const int bins_depth, bins_img;        // creation-time bin counts (they size the device workspace)
  int nid_level;
  int bins_img_now, bins_depth_now;      // setNumBinsImg / setNumBinsDepth
  const std::string saveFilename;
  int tick = 1;     // ElasticFusion::tick (ElasticFusion.cpp:35): only getTick / setTick touch it there as well
  int nextId = 0;
  int deforms = 0, fernDeforms = 0;
  std::map<std::string, std::shared_ptr<Context>> m_contexts;
  std::vector<std::shared_ptr<ReferenceFrame>> m_referenceFrames;
  std::map<int, std::shared_ptr<ReferenceFrame>> m_contextToReferenceFrameMap;
};

}  // namespace dms

#if defined(EIGEN_CORE_H) || defined(EIGEN_CORE_MODULE_H) || defined(DMS_WITH_EIGEN) || defined(DMS_EIGEN_MATRIX4F_DECLARED)
// the reference's names (DMS_EIGEN_MATRIX4F_DECLARED: a build without Eigen that declares its own Eigen::Matrix4f stand-in)
typedef dms::ElasticFusionT<Eigen::Matrix4f> ElasticFusion;
typedef dms::ContextT<Eigen::Matrix4f> Context;
#endif
